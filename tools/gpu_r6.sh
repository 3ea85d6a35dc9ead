#!/bin/bash
# Round-6 GPU sessions (one gpurun call each):  gpurun -- 'bash tools/gpu_r6.sh <step> [args]'
#   fwdpipe <tag>   row-lane correctness sweep on the product build, then scan_bench A/B of the library variants given
set -u
step=${1:-fwdpipe}; shift || true
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
ab() {  # out dir, shapes, pitch args, variants...
  local out=$1 shapes=$2 extra=$3; shift 3
  for v in "$@"; do
    [ "$v" = "-" ] && v=""
    echo "== libsigma_hip$v"
    SIGMA_HIP_LIB=$R/sigma_amd/lib/libsigma_hip$v.so timeout 900 python tools/scan_bench.py --fine --iters 20 --shapes $shapes $extra --out $out/scan_bench$v.jsonl 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('%-14s pitch %4d fwd %7.1f us (%.3f)  bwd %7.1f us (%.3f)' % (r['shape'], r.get('ckpt_pitch', 0), r['fwd_us'], r.get('fwd_frac_of_8TBs', 0), r.get('bwd_us', 0), r.get('bwd_frac_of_8TBs', 0)))
"
  done 2>&1 | tee -a $out/scan_bench.txt
}
case $step in
fwdpipe)
  tag=${1:-r6_fwdpipe}; shift || true; out=gpurun_out/$tag; mkdir -p $out
  ( time timeout 900 python tools/rowlane_check.py --out $out/rowlane_check.jsonl ) > $out/rowlane_check.log 2>&1
  tail -3 $out/rowlane_check.log | cut -c1-300
  python - <<PY
import json
for l in open("$out/rowlane_check.jsonl"):
    r = json.loads(l)
    if not r.get("ok"):
        print("FAIL", r.get("shape"), r.get("opts"), {k: (v["bad"], v["max_abs"]) for k, v in r.items() if isinstance(v, dict) and "bad" in v and v["bad"]}, r.get("error", "")[:200])
PY
  ab $out enc_s2_b16,enc_s2_b2,enc_s0,enc_s1_b16,enc_s0_b16,enc_s3_b16 "--pitch 16" "$@"
  ab $out dec_s0_b8,conmb_s0_b8,dec_s1_b8,cromb_s0_b8,dec_s0,cromb_s0 "--pitch 16" "$@"
  ;;
abl)   # ablation builds (timing only, self-test skipped): abl <tag> <shapes> <variants...>
  tag=$1; shapes=$2; shift 2; out=gpurun_out/$tag; mkdir -p $out
  export SIGMA_BENCH_NO_SELFTEST=1
  ab $out $shapes "--pitch 16" "$@"
  ;;
prof)  # phase profile of the row-lane kernels: prof <tag> <shape> <prof-build suffixes...>
  tag=$1; shape=$2; shift 2; out=gpurun_out/$tag; mkdir -p $out
  for v in "$@"; do
    echo "== libsigma_hip$v"
    SIGMA_HIP_LIB=$R/sigma_amd/lib/libsigma_hip$v.so timeout 300 python tools/rowlane_prof.py $shape 2>&1 | grep "^{" | tee -a $out/phases$v.jsonl | cut -c1-600
  done
  ;;
pitches)  # scan_bench of shapes at several forced pitches: pitches <tag> <shapes> <pitch...>
  tag=$1; shapes=$2; shift 2; out=gpurun_out/$tag; mkdir -p $out
  for pp in "$@"; do
    echo "== pitch $pp"
    ab $out $shapes "$([ $pp = auto ] && echo '' || echo --pitch $pp)" -
  done
  ;;
gemm)  # gemm_bench A/B: gemm <tag> <only-columns> <variants...>
  tag=$1; only=$2; shift 2; out=gpurun_out/$tag; mkdir -p $out
  for v in "$@"; do
    [ "$v" = "-" ] && v=""
    echo "== libsigma_hip$v"
    SIGMA_HIP_LIB=$R/sigma_amd/lib/libsigma_hip$v.so timeout 600 python tools/gemm_bench.py --iters 20 --only $only --out $out/gemm$v.jsonl 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('%-18s' % r['shape'], ' '.join('%s %7.1f' % (k[:-3], v) for k, v in r.items() if k.endswith('_us')))
"
  done 2>&1 | tee -a $out/gemm.txt
  ;;
ubench)
  tag=${1:-r6_ubench}; out=gpurun_out/$tag; mkdir -p $out
  timeout 600 tools/ubench/bin/stateloop_ubench | tee $out/stateloop_ubench.jsonl
  ;;
esac
