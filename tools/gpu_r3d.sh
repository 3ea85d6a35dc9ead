#!/bin/bash
# round 3, GPU call D: GEMM v3 (hand-counted asynchronous operand loads): parity + speed at ring depth 1/2/3
TAG=${1:-r03d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gemm_gpu.py -x -q ) > $OUT/pytest_gemm.log 2>&1; tail -4 $OUT/pytest_gemm.log
ONLY=nt_split3,nn_split3,tn_split3,nt_split6,nt_fp32,nn_fp32,tn_fp32
timeout 300 python tools/gemm_bench.py --iters 10 --only $ONLY --out $OUT/gemm_bench_d2.jsonl > $OUT/gemm_bench_d2.log 2>&1
SIGMA_HIP_LIB=$R/sigma_amd/lib/libsigma_hip_gd1.so timeout 300 python tools/gemm_bench.py --iters 10 --only nt_split3,nn_split3,tn_split3 --out $OUT/gemm_bench_d1.jsonl > $OUT/gemm_bench_d1.log 2>&1
SIGMA_HIP_LIB=$R/sigma_amd/lib/libsigma_hip_gd3.so timeout 300 python tools/gemm_bench.py --iters 10 --only nt_split3,nn_split3,tn_split3 --out $OUT/gemm_bench_d3.jsonl > $OUT/gemm_bench_d3.log 2>&1
python - <<PY
import json
def load(f):
    d={}
    try:
        for l in open(f):
            r=json.loads(l)
            if 'shape' in r: d[r['shape']]=r
    except Exception as e: print(f, e)
    return d
d1,d2,d3=[load(f"$OUT/gemm_bench_d{i}.jsonl") for i in (1,2,3)]
for k in d2:
    b=d2[k]; a=d1.get(k,{}); c=d3.get(k,{})
    g=lambda r,n: r.get(n,0)
    print(f"{k:18s} nt d1/2/3 {g(a,'nt_split3_us'):6.1f} {g(b,'nt_split3_us'):6.1f} {g(c,'nt_split3_us'):6.1f} s6 {g(b,'nt_split6_us'):6.1f} fp32 {g(b,'nt_fp32_us'):6.1f} | nn {g(a,'nn_split3_us'):6.1f} {g(b,'nn_split3_us'):6.1f} {g(c,'nn_split3_us'):6.1f} fp32 {g(b,'nn_fp32_us'):6.1f} | tn {g(a,'tn_split3_us'):6.1f} {g(b,'tn_split3_us'):6.1f} {g(c,'tn_split3_us'):6.1f} fp32 {g(b,'tn_fp32_us'):6.1f}")
PY
