#!/bin/bash
# round 3, GPU call E: GEMM with the unserialised epilogue; quad-row kernels with 16-byte row accesses; precision detail
TAG=${1:-r03e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gemm_gpu.py -x -q ) > $OUT/pytest_gemm.log 2>&1; tail -6 $OUT/pytest_gemm.log
timeout 300 python tools/gemm_bench.py --iters 10 --only nt_split3,nn_split3,tn_split3,nt_split6,nn_split6,tn_split6,nt_fp32,nn_fp32,tn_fp32 --out $OUT/gemm_bench.jsonl > $OUT/gemm_bench.log 2>&1
python - <<PY
import json
for l in open("$OUT/gemm_bench.jsonl"):
    r=json.loads(l)
    if 'shape' not in r: continue
    g=lambda n: r.get(n,0)
    print(f"{r['shape']:18s} nt {g('nt_split3_us'):6.1f} s6 {g('nt_split6_us'):6.1f} fp32 {g('nt_fp32_us'):6.1f} | nn {g('nn_split3_us'):6.1f} s6 {g('nn_split6_us'):6.1f} fp32 {g('nn_fp32_us'):6.1f} | tn {g('tn_split3_us'):6.1f} s6 {g('tn_split6_us'):6.1f} fp32 {g('tn_fp32_us'):6.1f}")
PY
( time timeout 600 python -m pytest tests/test_scan_gpu.py -x -q -k "quad or full_size or golden or stage" ) > $OUT/pytest_scan_quad.log 2>&1; tail -4 $OUT/pytest_scan_quad.log
timeout 200 python tools/scan_bench.py --shapes enc_s2_b16,enc_s0_b16,enc_s1_b16,enc_s3_b16,dec_s0_b8 --iters 10 --fine --out $OUT/scan_bench.jsonl 2>/dev/null | python -c "import sys,json; [print({k:(round(v,1) if isinstance(v,float) else v) for k,v in json.loads(l).items() if k in ('shape','fwd_us','bwd_us','ckpt_pitch')}) for l in sys.stdin if l.startswith('{')]"
timeout 300 python tools/grad_precision.py fp32 f3_d2_w2 f2_d2_w2 > $OUT/grad_precision3.jsonl 2> $OUT/grad_precision3.err; cut -c1-700 $OUT/grad_precision3.jsonl
( time timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gemm fp32 ) > $OUT/bench_fp32.log 2>&1; grep "^{" $OUT/bench_fp32.log | cut -c1-330
( time timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gemm split3 ) > $OUT/bench_split3.log 2>&1; grep "^{" $OUT/bench_split3.log | cut -c1-330
( time SIGMA_GEMM_FWD=3 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gemm split3 ) > $OUT/bench_split3_f3.log 2>&1; grep "^{" $OUT/bench_split3_f3.log | cut -c1-330
