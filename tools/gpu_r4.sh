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
    grep -v "^{" $out/rowlane_check.log | tail -5; python - <<PY
import json
for l in open("$out/rowlane_check.jsonl"):
    r = json.loads(l)
    bad = {k: (v["bad"], v["nan"], round(v["max_abs"], 6)) for k, v in r.items() if isinstance(v, dict) and "bad" in v and (v["bad"] or v["nan"])}
    print("ok  " if r["ok"] else "FAIL", r["shape"], r["opts"], r.get("error", ""), bad)
PY
    for pitch in 16 160; do
      timeout 600 python tools/scan_bench.py --fine --pitch $pitch --iters 10 --shapes enc_s2_b16,enc_s2_b2,enc_s0 --out $out/scan_bench_p$pitch.jsonl > $out/scan_bench_p$pitch.log 2>&1
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
e)  # row-lane (pitch 16) against the planner's own choice on every launch shape of the batch-8 and batch-1 steps
    SH=enc_s0_b16,enc_s1_b16,enc_s2_b16,enc_s3_b16,cromb_s0_b8,cromb_s1_b8,cromb_s2_b8,cromb_s3_b8,conmb_s0_b8,conmb_s1_b8,conmb_s2_b8,conmb_s3_b8,dec_s0_b8,dec_s1_b8,dec_s2_b8,enc_s0_b2,enc_s1_b2,enc_s2_b2,enc_s3_b2,cromb_s0,cromb_s1,cromb_s2,cromb_s3,conmb_s0,conmb_s1,conmb_s2,conmb_s3,dec_s0,dec_s1,dec_s2
    timeout 900 python tools/scan_bench.py --fine --pitch 16 --iters 10 --shapes $SH --out $out/p16.jsonl > /dev/null 2>&1
    SIGMA_CKPT_PITCH=auto timeout 900 python tools/scan_bench.py --fine --iters 10 --shapes $SH --out $out/auto.jsonl > /dev/null 2>&1
    python - <<PY
import json
a = {json.loads(l)["shape"]: json.loads(l) for l in open("$out/p16.jsonl")}
b = {json.loads(l)["shape"]: json.loads(l) for l in open("$out/auto.jsonl")}
for k in a:
    x, y = a[k], b[k]
    print("%-12s %-26s p16 fwd %6.0f bwd %6.0f | pitch %4d fwd %6.0f bwd %6.0f | sum %6.0f vs %6.0f  %s" % (k, x["dims"], x["fwd_us"], x["bwd_us"], y["ckpt_pitch"], y["fwd_us"], y["bwd_us"], x["fwd_us"] + x["bwd_us"], y["fwd_us"] + y["bwd_us"], "ROWLANE" if x["fwd_us"] + x["bwd_us"] < y["fwd_us"] + y["bwd_us"] else ""))
PY
    ;;
f)  # row-lane kernels under pytest + the model with the new pitch policy + bench lines (batch 8, one image per GPU)
    ( time timeout 900 python -m pytest tests/test_scan_gpu.py -q --tb=short -x -k "row_lane or real_stage or reversed_groups or checkpoint_tensor" ) > $out/pytest_scan.log 2>&1; tail -4 $out/pytest_scan.log | cut -c1-200
    ( time timeout 900 python -m pytest tests/test_model_gpu.py -q --tb=short -x ) > $out/pytest_model.log 2>&1; tail -4 $out/pytest_model.log | cut -c1-200
    ( timeout 600 python bench.py --no-cpu-baseline ) > $out/bench_b8.log 2>&1; grep "^{" $out/bench_b8.log | cut -c1-600
    ( SIGMA_CKPT_PITCH=norowlane timeout 600 python bench.py --no-cpu-baseline ) > $out/bench_b8_norowlane.log 2>&1; grep "^{" $out/bench_b8_norowlane.log | cut -c1-300
    ( timeout 600 python bench.py --per-gpu-batch 1 --graph --no-cpu-baseline ) > $out/bench_b1_graph.log 2>&1; grep "^{" $out/bench_b1_graph.log | cut -c1-300
    ( SIGMA_CKPT_PITCH=norowlane timeout 600 python bench.py --per-gpu-batch 1 --graph --no-cpu-baseline ) > $out/bench_b1_graph_norowlane.log 2>&1; grep "^{" $out/bench_b1_graph_norowlane.log | cut -c1-300
    ;;
g)  # parity additions of this round (evaluator multi-scale, oracle-driven 480x640 gradients), HBM traffic of the row-lane
    # kernels at the dominant launch shape (FETCH_SIZE / WRITE_SIZE passes only), kernel trace of the benchmark step
    ( time timeout 1200 python -m pytest tests/test_model_gpu.py -q --tb=short -x -k "evaluator or resize or gradients_vs_cpu_oracle" ) > $out/pytest_parity.log 2>&1; tail -4 $out/pytest_parity.log | cut -c1-300
    SCAN_BENCH_ARGS="--pitch 16" bash tools/gpu_pmc.sh r4_g/pmc enc_s2_b16 traffic > $out/pmc.txt 2>&1; grep -A3 "^== " $out/pmc.txt | grep -v "^--" | cut -c1-400
    R=$PWD; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -o b8 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline ) > $out/trace.log 2>&1; grep "^{" $out/trace.log | cut -c1-400
    tr=$(find $out/trace -name "*kernel_trace.csv" | head -1); python tools/prof_summary.py $tr --top 60 --last-ms 340 > $out/trace_summary.txt 2>&1; head -70 $out/trace_summary.txt | cut -c1-200
    ;;
h)  # stacked x_proj / dt_proj GEMMs, residual in the out_proj GEMM: parity, then the step
    ( time timeout 600 python -m pytest tests/test_gemm_gpu.py -q --tb=short -x ) > $out/pytest_gemm.log 2>&1; tail -4 $out/pytest_gemm.log | cut -c1-300
    ( time timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_pointwise_gpu.py -q --tb=short -x ) > $out/pytest_model.log 2>&1; tail -4 $out/pytest_model.log | cut -c1-300
    ( timeout 600 python bench.py --no-cpu-baseline ) > $out/bench_b8.log 2>&1; grep "^{" $out/bench_b8.log | cut -c1-700
    ( SIGMA_GEMM_XPROJ=fp32 timeout 600 python bench.py --no-cpu-baseline ) > $out/bench_b8_xproj_fp32.log 2>&1; grep "^{" $out/bench_b8_xproj_fp32.log | cut -c1-200
    ;;
i)  # LayerNorm pass-through, explicit gradient-buffer hand-off, GEMM self test / validation, smoke with the row-lane kernels
    ( time timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_pointwise_gpu.py tests/test_model_gpu.py -q --tb=short -x -k "not 480x640 and not 720x1280" ) > $out/pytest.log 2>&1; tail -4 $out/pytest.log | cut -c1-300
    ( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $out/smoke.log 2>&1; tail -3 $out/smoke.log | cut -c1-300
    ( timeout 600 python bench.py --no-cpu-baseline ) > $out/bench_b8.log 2>&1; grep "^{" $out/bench_b8.log | cut -c1-330
    ;;
j)  # decoder scale-residual backward kernel
    ( time timeout 900 python -m pytest tests/test_pointwise_gpu.py tests/test_model_gpu.py -q --tb=short -x -k "not 480x640 and not 720x1280" ) > $out/pytest.log 2>&1; grep -v "^  File" $out/pytest.log | tail -4 | cut -c1-300
    ( timeout 600 python bench.py --no-cpu-baseline ) > $out/bench_b8.log 2>&1; grep "^{" $out/bench_b8.log | cut -c1-330
    ;;
final)  # end-of-round evidence on the final code
    ( time AMD_LOG_LEVEL=1 timeout 1500 python -X faulthandler -m pytest tests -m gpu -q --tb=short ) > $out/pytest_gpu.log 2>&1; grep -v "^  File" $out/pytest_gpu.log | tail -4 | cut -c1-300
    ( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $out/smoke.log 2>&1; tail -5 $out/smoke.log | head -2 | cut -c1-300
    ( time timeout 900 python bench.py --steps 20 --warmup 5 --kernel-report $out/kernels.json ) > $out/bench.log 2>&1; grep "^{" $out/bench.log | cut -c1-500
    R=$PWD; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -o b8 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline ) > $out/trace.log 2>&1; grep "^{" $out/trace.log | cut -c1-200
    tr=$(find $out/trace -name "*kernel_trace.csv" | head -1); python tools/prof_summary.py $tr --top 70 --last-ms 340 > $out/bench_kernel_stats.txt 2>&1; head -12 $out/bench_kernel_stats.txt | cut -c1-180
    st=$(find $out/trace -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && python tools/prof_summary.py $st --top 40 > $out/bench_rocprof_stats_whole_run.txt 2>&1
    rm -f $tr
    SCAN_BENCH_ARGS="--pitch 16" bash tools/gpu_pmc.sh r4_final/pmc enc_s2_b16 traffic > $out/pmc.txt 2>&1; grep -A3 "^== " $out/pmc.txt | grep -v "^--" | cut -c1-300
    ( timeout 600 python bench.py --per-gpu-batch 1 --graph --no-cpu-baseline ) > $out/bench_b1_graph.log 2>&1; grep "^{" $out/bench_b1_graph.log | cut -c1-200
    ( timeout 600 python bench.py --backbone sigma_base --height 720 --width 1280 --classes 5 --per-gpu-batch 1 --no-cpu-baseline ) > $out/bench_config5.log 2>&1; grep "^{" $out/bench_config5.log | cut -c1-200
    ;;
final2)  # end-of-round evidence after the last GEMM change: full suite, smoke, bench (driver-style), kernel trace
    ( time AMD_LOG_LEVEL=1 timeout 1500 python -X faulthandler -m pytest tests -m gpu -q --tb=short ) > $out/pytest_gpu.log 2>&1; grep -v "^  File" $out/pytest_gpu.log | tail -4 | cut -c1-300
    ( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $out/smoke.log 2>&1; tail -5 $out/smoke.log | head -2 | cut -c1-300
    ( time timeout 900 python bench.py --steps 20 --warmup 5 --kernel-report $out/kernels.json ) > $out/bench.log 2>&1; grep "^{" $out/bench.log | cut -c1-500
    R=$PWD; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -o b8 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline ) > $out/trace.log 2>&1; grep "^{" $out/trace.log | cut -c1-200
    tr=$(find $out/trace -name "*kernel_trace.csv" | head -1); python tools/prof_summary.py $tr --top 70 --last-ms 332 > $out/bench_kernel_stats.txt 2>&1; head -12 $out/bench_kernel_stats.txt | cut -c1-180
    st=$(find $out/trace -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && python tools/prof_summary.py $st --top 40 > $out/bench_rocprof_stats_whole_run.txt 2>&1
    rm -f $tr
    ( timeout 600 python bench.py --per-gpu-batch 1 --graph --no-cpu-baseline ) > $out/bench_b1_graph.log 2>&1; grep "^{" $out/bench_b1_graph.log | cut -c1-200
    ;;
esac
