#!/bin/bash
# round 3, GPU call A: new GEMM kernels (parity + speed), full-size scan parity, issue-rate ubench, bwd4 phase timers,
# step benchmark with fp32 vs split3 GEMMs.
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gemm_gpu.py -x -q ) > $OUT/pytest_gemm.log 2>&1; tail -15 $OUT/pytest_gemm.log
( time timeout 300 python -m pytest tests/test_scan_gpu.py -x -q -k "full_size_step" ) > $OUT/pytest_fullsize.log 2>&1; tail -5 $OUT/pytest_fullsize.log
timeout 120 tools/ubench/bin/issue_ubench > $OUT/issue_ubench.jsonl 2> $OUT/issue_ubench.err; wc -l $OUT/issue_ubench.jsonl
VARIANTS='[["Q160 auto",160,{}],["Q160 W12",160,{"bwd_waves":12}],["Q160 W8",160,{"bwd_waves":8}],["Q160 notouch",160,{"bwd_touch":2}],["Q160 SB1",160,{"bwd_sb":1}],["Q160 nb1",160,{"bwd_nb":1}],["Q160 2wg",160,{"bwd_wgs":2}]]' \
  SIGMA_HIP_LIB=$R/sigma_amd/lib/libsigma_hip_prof.so timeout 200 python tools/bwd2_prof.py enc_s2_b16 > $OUT/bwd4_phases.jsonl 2> $OUT/bwd4_phases.err; cat $OUT/bwd4_phases.jsonl | cut -c1-600
timeout 300 python tools/gemm_bench.py --iters 10 --out $OUT/gemm_bench.jsonl > $OUT/gemm_bench.log 2>&1; cut -c1-400 $OUT/gemm_bench.log
( time timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gemm fp32 ) > $OUT/bench_fp32.log 2>&1; grep "^{" $OUT/bench_fp32.log | cut -c1-500
( time timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gemm split3 ) > $OUT/bench_split3.log 2>&1; grep "^{" $OUT/bench_split3.log | cut -c1-500
( time SIGMA_GEMM=split3 timeout 400 python -m pytest tests/test_model_gpu.py -x -q ) > $OUT/pytest_model_split3.log 2>&1; tail -5 $OUT/pytest_model_split3.log
