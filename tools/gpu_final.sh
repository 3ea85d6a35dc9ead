#!/bin/bash
# End-of-round evidence: headline bench, rocprofv3 kernel stats of the step and of the operator table, PMC traffic of the
# dominant shape.  (Tests: tools/gpu_round.sh.)  Usage: bash tools/gpu_final.sh [tag]
TAG=${1:-r02_final2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log | head -1
timeout 400 python tools/scan_bench.py --iters 10 --fine --out $OUT/scan_bench.jsonl > $OUT/scan_bench.log 2>&1
( time timeout 600 python bench.py --kernel-report $OUT/kernels.json ) > $OUT/bench.log 2>&1; grep "^{" $OUT/bench.log | cut -c1-400
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/rocprof_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_scan -o scan -- python $R/tools/scan_bench.py --shapes enc_s2_b16,enc_s0_b16,enc_s1_b16,enc_s3_b16,dec_s0_b8,conmb_s0_b8,enc_s0 --iters 5 --fine > $OUT/rocprof_scan.log 2>&1
cd $R
python tools/prof_summary.py $OUT/prof_bench/bench_kernel_trace.csv --last-ms 452 --top 60 > $OUT/bench_last452ms_kernel_stats.txt 2>&1
python tools/prof_summary.py $OUT/prof_scan/scan_kernel_stats.csv --top 14 > $OUT/scan_bench_kernel_stats.txt 2>&1
rm -f $OUT/prof_bench/bench_kernel_trace.csv $OUT/prof_scan/scan_kernel_trace.csv
bash tools/gpu_pmc.sh $TAG/pmc enc_s2_b16 all > $OUT/pmc.log 2>&1; tail -8 $OUT/pmc.log | cut -c1-300
head -12 $OUT/bench_last452ms_kernel_stats.txt | cut -c1-150
