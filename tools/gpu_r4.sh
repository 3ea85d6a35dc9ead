#!/bin/bash
# Round-4 GPU sessions (one gpurun call each): bash tools/gpu_r4.sh <step>; outputs under gpurun_out/r4_<step>/
set -u
step=${1:-a}
out=gpurun_out/r4_$step
mkdir -p $out
export TMPDIR=/tmp
case $step in
a)  # first light of the row-lane kernels: correctness sweep + timing against the quad-row kernels
    timeout 900 python tools/rowlane_check.py --out $out/rowlane_check.jsonl > $out/rowlane_check.log 2>&1
    tail -3 $out/rowlane_check.log
    for pitch in 16 160; do
      timeout 600 python tools/scan_bench.py --fine --pitch $pitch --iters 10 --shapes enc_s2_b16,enc_s1_b16,enc_s0_b16,enc_s2_b2,enc_s0 --out $out/scan_bench_p$pitch.jsonl > $out/scan_bench_p$pitch.log 2>&1
      cat $out/scan_bench_p$pitch.log | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print(r['shape'], r['ckpt_pitch'], 'fwd %.0f us (%.3f)' % (r['fwd_us'], r['fwd_frac_of_8TBs']), 'bwd %.0f us (%.3f)' % (r.get('bwd_us', 0), r.get('bwd_frac_of_8TBs', 0)))
"
    done
    ;;
b)  # phase breakdown of the row-lane kernels (development build with cycle counters)
    SIGMA_HIP_LIB=$PWD/sigma_amd/lib/libsigma_hip_rlprof.so timeout 300 python tools/rowlane_prof.py enc_s2_b16 enc_s0 > $out/rowlane_prof.jsonl 2> $out/rowlane_prof.err
    cat $out/rowlane_prof.jsonl; tail -3 $out/rowlane_prof.err
    ;;
c)  # ablation builds of the row-lane kernels (wrong results, timing only)
    for v in "" _abl1 _abl2 _abl4 _abl8 _abl16 _abl32 _abl64 _abl63 _abl127; do
      echo "== lib$v"
      SIGMA_HIP_LIB=$PWD/sigma_amd/lib/libsigma_hip$v.so timeout 300 python tools/scan_bench.py --fine --pitch 16 --iters 10 --shapes enc_s2_b16 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print(r['shape'], 'fwd %.0f us' % r['fwd_us'], 'bwd %.0f us' % r.get('bwd_us', 0))
"
    done > $out/ablation.txt 2>&1
    cat $out/ablation.txt
    ;;
d)  # PMC passes of the row-lane kernels (base build and the all-ablated build)
    export SCAN_BENCH_ARGS="--pitch 16"
    bash tools/gpu_pmc.sh r4_d/base enc_s2_b16 all > $out/pmc_base.txt 2>&1
    SIGMA_HIP_LIB=$PWD/sigma_amd/lib/libsigma_hip_abl63.so bash tools/gpu_pmc.sh r4_d/abl63 enc_s2_b16 all > $out/pmc_abl63.txt 2>&1
    grep -A3 "^== " $out/pmc_base.txt | grep -v "^--" ; echo ABL63; grep -A3 "^== " $out/pmc_abl63.txt | grep -v "^--"
    ;;
esac
