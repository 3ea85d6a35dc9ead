#!/bin/bash
# round 3, GPU call N: bmm projections, dz in place, LN block partials, whole-plane dwconv: tests, aux bench, step bench (eager + graph), tail
TAG=${1:-r03n}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_pointwise_gpu.py tests/test_model_gpu.py -q --tb=short -x -k "not 720x1280 and not sigma_small_480x640_gradients and not graphed_data_parallel" ) > $OUT/pytest_model.log 2>&1; grep -v "^$" $OUT/pytest_model.log | tail -15 | cut -c1-220
timeout 200 python tools/aux_bench.py --iters 10 --out $OUT/aux_bench.jsonl 2>/dev/null | grep -i "layernorm\|dwconv" | cut -c1-200
( time timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $OUT/bench.log 2>&1; grep "^{" $OUT/bench.log | cut -c1-330
( time timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --graph ) > $OUT/bench_graph.log 2>&1; grep "^{" $OUT/bench_graph.log | cut -c1-330
timeout 300 python tools/copy_parents.py > $OUT/copy_parents.txt 2>&1; head -45 $OUT/copy_parents.txt | grep -v Warning | cut -c1-60,98-200
