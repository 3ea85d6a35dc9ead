#!/bin/bash
# round 3, GPU call F: the whole GPU suite with the new defaults, headline bench, batch-1 data-parallel graph step
TAG=${1:-r03f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > $OUT/pytest_gpu.log 2>&1; tail -30 $OUT/pytest_gpu.log | cut -c1-200
( time timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $OUT/bench.log 2>&1; grep "^{" $OUT/bench.log | cut -c1-400
( time timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 1 --force-ddp ) > $OUT/bench_b1_ddp_eager.log 2>&1; grep "^{" $OUT/bench_b1_ddp_eager.log | cut -c1-300
( time timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 1 --force-ddp --graph ) > $OUT/bench_b1_ddp_graph.log 2>&1; grep "^{" $OUT/bench_b1_ddp_graph.log | cut -c1-300; tail -3 $OUT/bench_b1_ddp_graph.log | cut -c1-300
( time timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 1 --graph ) > $OUT/bench_b1_graph.log 2>&1; grep "^{" $OUT/bench_b1_graph.log | cut -c1-300
