# Memory-side and issue-side counters of the split-operand GEMM kernels on one shape (round 6): one rocprofv3 pass per counter set.
#   bash tools/diag/pmc_gemm2.sh [shape] [columns]
shape=${1:-enc_s2_in_proj}; cols=${2:-nt_split3,nn_split3,tn_split3}
out=gpurun_out/r6_pmc_gemm; mkdir -p $out; export TMPDIR=/tmp; R=$PWD; cd /tmp
i=0
for set in \
  "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
  "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU" \
  "GRBM_GUI_ACTIVE FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
  "TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
  "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
  if [ -n "$PMC_ONLY" ] && ! echo " $PMC_ONLY " | grep -q " $((i+1)) "; then i=$((i+1)); continue; fi
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$out/p$i -o p -- python $R/tools/gemm_bench.py --shapes $shape --iters 3 --only $cols > $R/$out/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $R/$out/p$i.log)"
done
cd $R; python3 - <<PY | tee $out/summary_$shape.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$out/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "gemm_split3" not in r["Kernel_Name"]: continue
        agg[r["Kernel_Name"].split("gemm_split3_kernel")[1][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print("   %-36s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
