#!/bin/bash
# round 3, GPU call S: row-strip depthwise conv, LayerNorm grids of one resident round: tests, aux bench, step bench
TAG=${1:-r03s}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 400 python -m pytest tests/test_pointwise_gpu.py tests/test_model_gpu.py -q --tb=short -x -k "dwconv or layernorm or fused_ss2d or fixtures or vss_block or conmb or cromb" ) > $OUT/pytest.log 2>&1; grep -v "^$" $OUT/pytest.log | tail -6 | cut -c1-220
timeout 200 python tools/aux_bench.py --iters 10 --out $OUT/aux_bench.jsonl 2>/dev/null | grep -i "dwconv\|layernorm" | cut -c1-200
( time timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $OUT/bench.log 2>&1; grep "^{" $OUT/bench.log | cut -c1-330
