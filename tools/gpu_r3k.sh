#!/bin/bash
# round 3, GPU call K: graphed data-parallel worker alone (native backtrace if it aborts), then the whole GPU suite
TAG=${1:-r03k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python -m tests.graph_ddp_worker fp32 > $OUT/worker_fp32.log 2>&1; rc=$?
echo "worker fp32 rc=$rc"; grep -v "^$" $OUT/worker_fp32.log | head -12 | cut -c1-200
if [ $rc -ne 0 ]; then
  timeout 600 rocgdb -batch -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 40" -ex "info threads" --args python -m tests.graph_ddp_worker fp32 > $OUT/worker_gdb.log 2>&1
  grep -n "^#\|SIGABRT\|signal" $OUT/worker_gdb.log | head -60 | cut -c1-220
  for v in "SIGMA_GEMM=fp32" "TORCH_NCCL_ASYNC_ERROR_HANDLING=0" "NCCL_DEBUG=WARN"; do
    env $v timeout 300 python -m tests.graph_ddp_worker fp32 > $OUT/worker_$v.log 2>&1; echo "$v rc=$?"; grep "graph_ddp_worker\|NCCL WARN" $OUT/worker_$v.log | tail -3 | cut -c1-200
  done
fi
( time timeout 1800 python -m pytest tests -m gpu -q --tb=short ) > $OUT/pytest_all.log 2>&1; grep -v "^\.\.\.\|^$" $OUT/pytest_all.log | tail -25 | cut -c1-240
