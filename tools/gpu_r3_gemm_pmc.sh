#!/bin/bash
# PMC passes of the final split-operand GEMM kernels (nt / nn / tn) on the stage-2 in_proj shape
TAG=${1:-r03_gemm_pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
pm() { local name=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_gemm/$name -o p -- python $R/tools/gemm_bench.py --shapes enc_s2_in_proj --iters 3 --only nt_split3,nn_split3,tn_split3 > $OUT/pmc_gemm_$name.log 2>&1; }
pm p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
pm p2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES
pm p3 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
cd $R
python3 - <<PY > $OUT/pmc_gemm_final.txt
import csv, glob, collections
print("# rocprofv3 --kernel-trace --pmc ... -- python tools/gemm_bench.py --shapes enc_s2_in_proj --iters 3 --only nt_split3,nn_split3,tn_split3  (final GEMM kernels of round 3; SQ cycle counters tick once per four clocks)")
for f in sorted(glob.glob("$OUT/pmc_gemm/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "gemm_split3" not in k: continue
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
cat $OUT/pmc_gemm_final.txt | cut -c1-400
rm -rf $OUT/pmc_gemm
