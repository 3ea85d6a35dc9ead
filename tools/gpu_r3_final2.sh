#!/bin/bash
# Round 3, last evidence run on the final code: targeted tests, default bench line, kernel trace, operator table,
# forward-only and sigma_base 720x1280 lines.  Usage: bash tools/gpu_r3_final2.sh [tag]
TAG=${1:-r03_final2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 400 python -m pytest tests/test_pointwise_gpu.py tests/test_model_gpu.py -q --tb=short -k "dwconv or layernorm or pointwise or fixtures or vss_block or fused_ss2d" ) > $OUT/pytest.log 2>&1; grep -v "^$" $OUT/pytest.log | tail -4 | cut -c1-220
( time timeout 600 python bench.py ) > $OUT/bench.log 2>&1; grep "^{" $OUT/bench.log | cut -c1-400
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/rocprof_bench.log 2>&1
cd $R
python tools/prof_summary.py $OUT/prof_bench/bench_kernel_trace.csv --last-ms 350 --top 80 > $OUT/bench_last350ms_kernel_stats.txt 2>&1
rm -f $OUT/prof_bench/bench_kernel_trace.csv
head -6 $OUT/bench_last350ms_kernel_stats.txt | cut -c1-150
timeout 400 python tools/scan_bench.py --iters 10 --fine --shapes enc_s2_b16,enc_s0_b16,enc_s1_b16,enc_s3_b16,dec_s1_b8,dec_s0_b8,conmb_s0_b8,enc_s0,base_s2_b2 --out $OUT/scan_bench.jsonl > $OUT/scan_bench.log 2>&1
python - <<PY
import json
for l in open("$OUT/scan_bench.jsonl"):
    r=json.loads(l); print(r["shape"], r["dims"], r.get("ckpt_pitch"), round(r["fwd_us"],1), round(r.get("fwd_frac_of_8TBs",0),3), round(r.get("bwd_us",0),1), round(r.get("bwd_frac_of_8TBs",0),3))
PY
timeout 200 python tools/eval_bench.py --backbone sigma_tiny --batch 2 --classes 9 > $OUT/eval_tiny_b2.log 2>&1; tail -1 $OUT/eval_tiny_b2.log | cut -c1-250
timeout 200 python tools/eval_bench.py --backbone sigma_small --batch 8 --classes 40 > $OUT/eval_small_b8.log 2>&1; tail -1 $OUT/eval_small_b8.log | cut -c1-250
( timeout 300 python bench.py --backbone sigma_base --height 720 --width 1280 --classes 5 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline ) > $OUT/bench_config5.log 2>&1; grep "^{" $OUT/bench_config5.log | cut -c1-220
