#!/bin/bash
# round 3, GPU call P: swizzled GEMM operand layout + LayerNorm sample index: GEMM tests / bench, LN tests, step bench
TAG=${1:-r03p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gemm_gpu.py tests/test_pointwise_gpu.py -q --tb=short -x ) > $OUT/pytest_gemm.log 2>&1; grep -v "^$" $OUT/pytest_gemm.log | tail -6 | cut -c1-220
timeout 300 python tools/gemm_bench.py --iters 10 --only nt_split3,nn_split3,tn_split3 --out $OUT/gemm_bench.jsonl > $OUT/gemm_bench.log 2>&1
python - <<PY
import json
for l in open("$OUT/gemm_bench.jsonl"):
    r=json.loads(l)
    if 'shape' in r: print(f"{r['shape']:18s} nt {r.get('nt_split3_us',0):6.1f} nn {r.get('nn_split3_us',0):6.1f} tn {r.get('tn_split3_us',0):6.1f}")
PY
( time timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $OUT/bench.log 2>&1; grep "^{" $OUT/bench.log | cut -c1-330
