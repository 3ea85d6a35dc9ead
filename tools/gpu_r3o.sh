#!/bin/bash
# round 3, GPU call O: XCD-local tiles, LayerNorm per-sample factor: tests, aux bench, step bench + kernel trace
TAG=${1:-r03o}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_pointwise_gpu.py tests/test_model_gpu.py -q --tb=short -x -k "not 720x1280 and not sigma_small_480x640_gradients and not graphed_data_parallel" ) > $OUT/pytest_model.log 2>&1; grep -v "^$" $OUT/pytest_model.log | tail -8 | cut -c1-220
timeout 200 python tools/aux_bench.py --iters 10 --out $OUT/aux_bench.jsonl 2>/dev/null | grep -i "cross\|dwconv" | cut -c1-200
( time timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $OUT/bench.log 2>&1; grep "^{" $OUT/bench.log | cut -c1-330
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/rocprof_bench.log 2>&1
cd $R
python tools/prof_summary.py $OUT/prof_bench/bench_kernel_trace.csv --last-ms 368 --top 70 > $OUT/bench_last368ms_kernel_stats.txt 2>&1
rm -f $OUT/prof_bench/bench_kernel_trace.csv
head -60 $OUT/bench_last368ms_kernel_stats.txt | cut -c1-150
