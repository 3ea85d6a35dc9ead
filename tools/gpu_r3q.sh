#!/bin/bash
# round 3, GPU call Q: batched loads in merge / split: tests, aux bench, step bench
TAG=${1:-r03q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_model_gpu.py -q --tb=short -x -k "cross_merge or fused_ss2d or fixtures" ) > $OUT/pytest.log 2>&1; grep -v "^$" $OUT/pytest.log | tail -4 | cut -c1-220
timeout 200 python tools/aux_bench.py --iters 10 --out $OUT/aux_bench.jsonl 2>/dev/null | grep -i "cross\|layernorm" | cut -c1-200
( time timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $OUT/bench.log 2>&1; grep "^{" $OUT/bench.log | cut -c1-330
