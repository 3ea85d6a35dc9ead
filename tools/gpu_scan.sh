#!/bin/bash
# scan-operator round trip: parity tests of the operator + micro-benchmark (+ optional sweep shapes)
TAG=${1:-scan}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( time timeout 900 python -m pytest tests/test_scan_gpu.py -m gpu -x -q ) > $OUT/pytest_scan.log 2>&1
tail -15 $OUT/pytest_scan.log
timeout 600 python tools/scan_bench.py --iters 10 --out $OUT/scan_bench.jsonl "$@" > $OUT/scan_bench.log 2>&1
python - <<PY
import json
for l in open("$OUT/scan_bench.log"):
    try: r = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print(r["shape"], r["items"], r["waves"], r.get("tiles"), "fwd %.0f us %.0f GB/s" % (r["fwd_us"], r["fwd_GBs"]),
          ("bwd %.0f us %.0f GB/s" % (r["bwd_us"], r["bwd_GBs"])) if "bwd_us" in r else "")
PY
