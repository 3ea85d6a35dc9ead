#!/bin/bash
# PMC passes (no tracing domains besides kernel-trace) over a short scan_bench run.
# usage: bash tools/gpu_pmc.sh <tag> <shapes>
TAG=${1:-pmc}; SHAPES=${2:-enc_s0_b8}; PASSES=${3:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG/$SHAPES
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/tools/scan_bench.py --shapes $SHAPES --iters 3 --fine $SCAN_BENCH_ARGS > $OUT/$name.log 2>&1
}
if [ "$PASSES" = all ]; then
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run p2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES
fi
run p3 GRBM_GUI_ACTIVE FETCH_SIZE
run p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls -R $OUT | head -30
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f.split("/")[-3] if "/" in f else f)
    for k, d in agg.items():
        if "scan" not in k: continue
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
