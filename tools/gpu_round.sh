#!/bin/bash
# One GPU-box round trip: parity tests, operator microbench, headline bench, rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
timeout 300 python tools/scan_bench.py --iters 10 --out $OUT/scan_bench.jsonl > $OUT/scan_bench.log 2>&1
cat $OUT/scan_bench.log | cut -c1-400
( time timeout 600 python bench.py --steps 3 --warmup 1 --kernel-report $OUT/kernels.json ) > $OUT/bench.log 2>&1
tail -5 $OUT/bench.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_scan -o scan -- python $R/tools/scan_bench.py --shapes enc_s0,enc_s2,dec_s0,conmb_s0 --iters 5 > $OUT/rocprof_scan.log 2>&1
ls -R $OUT/prof_scan | head -20
