#!/bin/bash
# round 3, GPU call I: model / LN / GEMM changes: quick tests, aux + gemm benches, step bench, kernel trace
TAG=${1:-r03i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_model_gpu.py tests/test_gemm_gpu.py -q --tb=short -x --deselect tests/test_model_gpu.py::test_sigma_base_720x1280_logits_vs_cpu_oracle --deselect tests/test_model_gpu.py::test_sigma_small_480x640_gradients_vs_cpu_oracle_path ) > $OUT/pytest_model.log 2>&1; grep -v "^\.\.\.\|^$" $OUT/pytest_model.log | tail -25 | cut -c1-240
timeout 200 python tools/aux_bench.py --iters 10 --out $OUT/aux_bench.jsonl 2>/dev/null | grep -i "layernorm" | cut -c1-200
timeout 300 python tools/gemm_bench.py --iters 10 --only nt_split3,nn_split3,tn_split3 --out $OUT/gemm_bench.jsonl > $OUT/gemm_bench.log 2>&1
python - <<PY
import json
for l in open("$OUT/gemm_bench.jsonl"):
    r=json.loads(l)
    if 'shape' in r: print(f"{r['shape']:18s} nt {r.get('nt_split3_us',0):6.1f} nn {r.get('nn_split3_us',0):6.1f} tn {r.get('tn_split3_us',0):6.1f}")
PY
( time timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $OUT/bench.log 2>&1; grep "^{" $OUT/bench.log | cut -c1-330
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/rocprof_bench.log 2>&1
cd $R
python tools/prof_summary.py $OUT/prof_bench/bench_kernel_trace.csv --last-ms 410 --top 80 > $OUT/bench_last410ms_kernel_stats.txt 2>&1
rm -f $OUT/prof_bench/bench_kernel_trace.csv
head -45 $OUT/bench_last410ms_kernel_stats.txt | cut -c1-150
