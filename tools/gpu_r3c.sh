#!/bin/bash
# round 3, GPU call C: ablations of scan_bwd4 and of the GEMM kernel, GEMM PMC, precision of the remaining settings
TAG=${1:-r03c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for v in "" _abl1 _abl2 _abl3 _abl4 _abl8 _abl16 _abl28 _abl32; do
  echo "== bwd4 lib$v" >> $OUT/bwd4_ablation.txt
  SIGMA_HIP_LIB=$R/sigma_amd/lib/libsigma_hip$v.so timeout 120 python tools/scan_bench.py --shapes enc_s2_b16 --iters 6 --fine 2>/dev/null | python -c "import sys,json; [print({k:round(v,1) for k,v in json.loads(l).items() if k in ('fwd_us','bwd_us')}) for l in sys.stdin if l.startswith('{')]" >> $OUT/bwd4_ablation.txt
done
cat $OUT/bwd4_ablation.txt
for v in "" _gabl1 _gabl2 _gabl4 _gabl8 _gabl16 _gabl6 _gabl7; do
  echo "== gemm lib$v" >> $OUT/gemm_ablation.txt
  SIGMA_HIP_LIB=$R/sigma_amd/lib/libsigma_hip$v.so timeout 120 python tools/gemm_bench.py --shapes enc_s2_in_proj,enc_s0_in_proj,enc_s1_out_proj --iters 10 --only nt_split3,nn_split3,tn_split3 2>/dev/null | python -c "import sys,json; [print({k:v for k,v in json.loads(l).items() if k=='shape' or k.endswith('_us')}) for l in sys.stdin if l.startswith('{\"shape')]" >> $OUT/gemm_ablation.txt
done
cat $OUT/gemm_ablation.txt
timeout 300 python tools/grad_precision.py fp32 f3_d2_w2 f3_d2_w3 f3_d3_w3 f3_d3_w3 f3_d3_w3 f3_d3_w2 > $OUT/grad_precision2.jsonl 2> $OUT/grad_precision2.err; cut -c1-300 $OUT/grad_precision2.jsonl
cd /tmp
pm() { local name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_gemm/$name -o p -- python $R/tools/gemm_bench.py --shapes enc_s2_in_proj --iters 3 --only nt_split3,tn_split3 > $OUT/pmc_gemm_$name.log 2>&1; }
pm p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
pm p2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES
pm p3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
pm p4 FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
cd $R
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc_gemm/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "gemm_split3" not in k: continue
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
rm -rf $OUT/pmc_gemm/*/*/*.db 2>/dev/null
du -sh $OUT
