#!/bin/bash
# End-of-round-3 evidence: the whole GPU suite, smoke, headline bench (with cpu baseline), kernel trace of the step, PMC
# traffic of the dominant scan launch, one-image-per-GPU lines.  Usage: bash tools/gpu_r3_final.sh [tag]
TAG=${1:-r03_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short ) > $OUT/pytest_all.log 2>&1; grep -v "^\.\.\.\|^$" $OUT/pytest_all.log | tail -12 | cut -c1-240
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log | head -2 | cut -c1-200
( time timeout 600 python bench.py ) > $OUT/bench.log 2>&1; grep "^{" $OUT/bench.log | cut -c1-1500
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/rocprof_bench.log 2>&1
cd $R
python tools/prof_summary.py $OUT/prof_bench/bench_kernel_trace.csv --last-ms 352 --top 80 > $OUT/bench_last352ms_kernel_stats.txt 2>&1
rm -f $OUT/prof_bench/bench_kernel_trace.csv
head -8 $OUT/bench_last352ms_kernel_stats.txt | cut -c1-150
bash tools/gpu_pmc.sh $TAG/pmc enc_s2_b16 traffic > $OUT/pmc.log 2>&1; tail -6 $OUT/pmc.log | cut -c1-300
( timeout 300 python bench.py --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --force-ddp ) > $OUT/bench_b1_ddp.log 2>&1; grep "^{" $OUT/bench_b1_ddp.log | cut -c1-200
( timeout 300 python bench.py --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --force-ddp --graph ) > $OUT/bench_b1_ddp_graph.log 2>&1; grep "^{" $OUT/bench_b1_ddp_graph.log | cut -c1-200
( timeout 300 python bench.py --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --graph ) > $OUT/bench_b1_graph.log 2>&1; grep "^{" $OUT/bench_b1_graph.log | cut -c1-200
