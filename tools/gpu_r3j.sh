#!/bin/bash
# round 3, GPU call J: graphed data-parallel test alone, dwconv test, step bench + kernel trace, then the whole GPU suite
TAG=${1:-r03j}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 400 python -m pytest tests/test_model_gpu.py -q --tb=short -x -k "graphed_data_parallel or dwconv or real_model_under_ddp" ) > $OUT/pytest_ddp.log 2>&1; grep -v "^$" $OUT/pytest_ddp.log | tail -12 | cut -c1-240
( time timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $OUT/bench.log 2>&1; grep "^{" $OUT/bench.log | cut -c1-330
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/rocprof_bench.log 2>&1
cd $R
python tools/prof_summary.py $OUT/prof_bench/bench_kernel_trace.csv --last-ms 405 --top 80 > $OUT/bench_last405ms_kernel_stats.txt 2>&1
rm -f $OUT/prof_bench/bench_kernel_trace.csv
head -40 $OUT/bench_last405ms_kernel_stats.txt | cut -c1-150
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short ) > $OUT/pytest_all.log 2>&1; grep -v "^\.\.\.\|^$" $OUT/pytest_all.log | tail -15 | cut -c1-240
