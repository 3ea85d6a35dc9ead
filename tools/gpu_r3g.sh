#!/bin/bash
# round 3, GPU call G: whole GPU suite (no -x), kernel trace of the headline step
TAG=${1:-r03g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests -m gpu -q --durations=15 ) > $OUT/pytest_gpu.log 2>&1; tail -40 $OUT/pytest_gpu.log | cut -c1-220
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/rocprof_bench.log 2>&1
cd $R
python tools/prof_summary.py $OUT/prof_bench/bench_kernel_trace.csv --last-ms 420 --top 80 > $OUT/bench_last420ms_kernel_stats.txt 2>&1
rm -f $OUT/prof_bench/bench_kernel_trace.csv
head -60 $OUT/bench_last420ms_kernel_stats.txt | cut -c1-170
