#!/bin/bash
# model-level round trip: model parity tests + headline bench (+ kernel report)
TAG=${1:-model}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( time timeout 900 python -m pytest tests/test_model_gpu.py tests/test_scan_gpu.py -m gpu -x -q ) > $OUT/pytest_model.log 2>&1
tail -12 $OUT/pytest_model.log
( time timeout 600 python bench.py --steps 3 --warmup 1 --kernel-report $OUT/kernels.json "$@" ) > $OUT/bench.log 2>&1
tail -4 $OUT/bench.log | cut -c1-1500
