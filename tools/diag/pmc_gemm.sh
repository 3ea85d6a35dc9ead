mkdir -p gpurun_out/r5_pmc_gemm; export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r5_pmc_gemm/p1 -o p -- python $R/tools/gemm_bench.py --shapes enc_s2_in_proj --iters 3 --only nt_split3,nn_split3,tn_split3 > $R/gpurun_out/r5_pmc_gemm/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/r5_pmc_gemm/p2 -o p -- python $R/tools/gemm_bench.py --shapes enc_s2_in_proj --iters 3 --only nt_split3,nn_split3,tn_split3 > $R/gpurun_out/r5_pmc_gemm/p2.log 2>&1
cd $R; python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r5_pmc_gemm/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "gemm_split3" not in k: continue
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
