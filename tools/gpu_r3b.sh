#!/bin/bash
# round 3, GPU call B: persistent GEMM (depth 1/2/3, 2 or 3 pieces), gradient precision per GEMM kind, kernel trace of the step
TAG=${1:-r03b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gemm_gpu.py -x -q ) > $OUT/pytest_gemm.log 2>&1; tail -4 $OUT/pytest_gemm.log
timeout 300 python tools/gemm_bench.py --iters 10 --out $OUT/gemm_bench_d2.jsonl > $OUT/gemm_bench_d2.log 2>&1; tail -2 $OUT/gemm_bench_d2.log | cut -c1-300
SIGMA_HIP_LIB=$R/sigma_amd/lib/libsigma_hip_gd1.so timeout 300 python tools/gemm_bench.py --iters 10 --out $OUT/gemm_bench_d1.jsonl > $OUT/gemm_bench_d1.log 2>&1
SIGMA_HIP_LIB=$R/sigma_amd/lib/libsigma_hip_gd3.so timeout 300 python tools/gemm_bench.py --iters 10 --out $OUT/gemm_bench_d3.jsonl > $OUT/gemm_bench_d3.log 2>&1
timeout 400 python tools/grad_precision.py > $OUT/grad_precision.jsonl 2> $OUT/grad_precision.err; cut -c1-260 $OUT/grad_precision.jsonl
( time timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gemm split3 ) > $OUT/bench_split3.log 2>&1; grep "^{" $OUT/bench_split3.log | cut -c1-400
( time SIGMA_GEMM_DGRAD=3 SIGMA_GEMM_WGRAD=3 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gemm split3 ) > $OUT/bench_split3_d3w3.log 2>&1; grep "^{" $OUT/bench_split3_d3w3.log | cut -c1-400
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --gemm split3 > $OUT/rocprof_bench.log 2>&1
cd $R
python tools/prof_summary.py $OUT/prof_bench/bench_kernel_trace.csv --last-ms 450 --top 70 > $OUT/bench_split3_last450ms_kernel_stats.txt 2>&1
rm -f $OUT/prof_bench/bench_kernel_trace.csv
head -30 $OUT/bench_split3_last450ms_kernel_stats.txt | cut -c1-180
