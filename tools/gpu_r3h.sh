#!/bin/bash
# round 3, GPU call H: suite without the two graph tests (full tracebacks), then each graph test alone; aux kernel rooflines
TAG=${1:-r03h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short --durations=8 --deselect tests/test_model_gpu.py::test_graphed_data_parallel_step_matches_ddp_step --deselect tests/test_model_gpu.py::test_hip_graph_replay_equals_eager_step ) > $OUT/pytest_gpu.log 2>&1; grep -v "^\.\.\.\|^$" $OUT/pytest_gpu.log | tail -60 | cut -c1-240
( time timeout 300 python -m pytest tests/test_model_gpu.py -q --tb=short -k test_hip_graph_replay_equals_eager_step ) > $OUT/pytest_graph1.log 2>&1; tail -25 $OUT/pytest_graph1.log | cut -c1-240
( time timeout 300 python -m pytest tests/test_model_gpu.py -q --tb=short -k test_graphed_data_parallel_step_matches_ddp_step ) > $OUT/pytest_graph2.log 2>&1; tail -25 $OUT/pytest_graph2.log | cut -c1-240
timeout 300 python tools/aux_bench.py --iters 10 --out $OUT/aux_bench.jsonl > $OUT/aux_bench.log 2>&1; cut -c1-200 $OUT/aux_bench.log | tail -45
