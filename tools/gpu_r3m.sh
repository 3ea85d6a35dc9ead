#!/bin/bash
# round 3, GPU call M: layout / pointwise kernels: unit tests, model fixtures, step bench, ATen tail by node
TAG=${1:-r03m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_pointwise_gpu.py -q --tb=short ) > $OUT/pytest_pointwise.log 2>&1; grep -v "^$" $OUT/pytest_pointwise.log | tail -30 | cut -c1-220
( time timeout 600 python -m pytest tests/test_model_gpu.py -q --tb=short -x -k "fixtures or conmb or cromb or hip_graph_replay or oracle_backend or real_model" ) > $OUT/pytest_model.log 2>&1; grep -v "^$" $OUT/pytest_model.log | tail -15 | cut -c1-220
( time timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $OUT/bench.log 2>&1; grep "^{" $OUT/bench.log | cut -c1-330
timeout 300 python tools/copy_parents.py > $OUT/copy_parents.txt 2>&1; head -75 $OUT/copy_parents.txt | grep -v Warning | cut -c1-60,98-200
