#!/bin/bash
# GPU sessions, one gpurun call each:  gpurun -- 'bash tools/gpu.sh <step> [args]'   (outputs under gpurun_out/<tag>/)
#   driver [n] [tag]     the DRIVER's own test command, n times in a row (default 1), then smoke()
#   files <tag> <pytest args...>   a subset of the suite
#   bench <tag> [bench.py args]    bench.py line (+ per-launch kernel report)
#   trace <tag> [bench.py args]    rocprofv3 --kernel-trace --stats of a short bench run, summarised by tools/prof_summary.py
#   pmc <tag> <shape> [scan_bench args]   FETCH_SIZE / WRITE_SIZE passes of one scan launch shape (tools/gpu_pmc.sh)
#   scanbench <tag> <shapes> [lib suffixes...]   tools/scan_bench.py on the shapes, once per library variant ("" = the product build)
#   extras [tag]         forward-only lines, sigma_base 720x1280, batch-8 HIP graph, step without the TunableOp table
#   final [n]            end-of-round evidence on the final code: driver x n, bench, trace, pmc of the dominant launch, round-5 tree on the same box
set -u
step=${1:-driver}
shift || true
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
DRIVER_CMD="python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider"      # verbatim from GPUTEST_r04.json

driver() {  # n, out dir
  local n=$1 out=$2 rc
  mkdir -p $out
  : > $out/driver_cmd_summary.txt
  for i in $(seq 1 $n); do
    ( time $DRIVER_CMD ) > $out/pytest_$i.log 2>&1; rc=$?
    { echo "run $i of $n: \$ $DRIVER_CMD   -> rc=$rc"; grep -E " passed| failed| error|^real" $out/pytest_$i.log | tail -3 | cut -c1-200; } | tee -a $out/driver_cmd_summary.txt
  done
  ( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/driver_cmd_summary.txt
  grep "smoke ok" $out/smoke.log | cut -c1-200 | tee -a $out/driver_cmd_summary.txt
}
bench() {  # out dir, bench args
  local out=$1; shift
  mkdir -p $out
  ( time timeout 900 python bench.py --kernel-report $out/kernels.json "$@" ) > $out/bench.log 2>&1; grep "^{" $out/bench.log | cut -c1-700
}
trace() {  # out dir, bench args
  local out=$1; shift
  mkdir -p $out
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -o b8 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" ) > $out/trace.log 2>&1
  grep "^{" $out/trace.log | cut -c1-200
  local ms=$(grep "^{" $out/trace.log | python -c "import sys, json; print(int(2 * json.loads(sys.stdin.readline())['ms_per_step']) + 2)" 2>/dev/null || echo 340)
  tr=$(find $out/trace -name "*kernel_trace.csv" | head -1)
  python tools/prof_summary.py $tr --top 70 --last-ms $ms > $out/bench_kernel_stats.txt 2>&1; head -14 $out/bench_kernel_stats.txt | cut -c1-180
  st=$(find $out/trace -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && python tools/prof_summary.py $st --top 40 > $out/bench_rocprof_stats_whole_run.txt 2>&1
  rm -f $tr
}

case $step in
driver)
  driver ${1:-1} gpurun_out/${2:-driver}
  ;;
files)
  tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
  ( time timeout 1500 python3 -m pytest "$@" -q -p no:cacheprovider ) > $out/pytest.log 2>&1; echo "rc=$?"; grep -v "^  File" $out/pytest.log | tail -6 | cut -c1-300
  ;;
bench)
  tag=$1; shift; bench gpurun_out/$tag "$@"
  ;;
trace)
  tag=$1; shift; trace gpurun_out/$tag "$@"
  ;;
pmc)
  tag=$1; shape=$2; shift 2
  SCAN_BENCH_ARGS="$*" bash tools/gpu_pmc.sh $tag/pmc $shape traffic > gpurun_out/$tag/pmc.txt 2>&1; grep -A3 "^== " gpurun_out/$tag/pmc.txt | grep -v "^--" | cut -c1-300
  ;;
scanbench)
  tag=$1; shapes=$2; shift 2; out=gpurun_out/$tag; mkdir -p $out
  [ $# -eq 0 ] && set -- ""
  for v in "$@"; do
    echo "== libsigma_hip$v"
    SIGMA_HIP_LIB=$R/sigma_amd/lib/libsigma_hip$v.so timeout 600 python tools/scan_bench.py --fine --iters 20 --shapes $shapes $SCAN_BENCH_ARGS --out $out/scan_bench$v.jsonl 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('%-14s pitch %4d fwd %7.1f us (%.3f)  bwd %7.1f us (%.3f)' % (r['shape'], r.get('ckpt_pitch', 0), r['fwd_us'], r.get('fwd_frac_of_8TBs', 0), r.get('bwd_us', 0), r.get('bwd_frac_of_8TBs', 0)))
"
  done 2>&1 | tee $out/scan_bench.txt
  ;;
final)
  out=gpurun_out/final
  driver ${1:-3} $out
  bench $out --steps 20 --warmup 5
  trace $out
  SCAN_BENCH_ARGS="" bash tools/gpu_pmc.sh final/pmc enc_s2_b16 traffic > $out/pmc.txt 2>&1; grep -A3 "^== " $out/pmc.txt | grep -v "^--" | cut -c1-300
  # the driver's N-rank launch line with one rank (launch plumbing; bench.py builds a process group only for world > 1 or --force-ddp): train.py:107,168
  ( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline ) > $out/bench_torchrun_one_rank.log 2>&1; grep "^{" $out/bench_torchrun_one_rank.log | cut -c1-300
  ( timeout 600 python bench.py --per-gpu-batch 1 --graph --no-cpu-baseline ) > $out/bench_b1_graph.log 2>&1; grep "^{" $out/bench_b1_graph.log | cut -c1-200
  # the same bench line from the round-5 tree (.r05/: `git archive` of the round-5 final commit + its library, not tracked) and
  # from this tree, alternating ON THIS BOX: boxes of the pool differ by 2-3 % in sustained clock, rounds are compared here
  if [ -d .r05 ]; then
    : > $out/ab_r05_same_box.txt
    for rep in 1 2; do
      for tree in .r05 .; do
        ( cd $tree && timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-5s rep $rep  %.3f images/s  %.2f ms/step  scan bwd %.1f us  fwd %.1f us' % ('$tree', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline_fwd']['avg_launch_us']))" ) | tee -a $out/ab_r05_same_box.txt
      done
    done
  fi
  ;;
extras)
  # measurements beside the headline: forward only (BASELINE configs[1]), sigma_base 720x1280 (configs[4]), the batch-8 step as a
  # HIP graph, and the step without the TunableOp table for the vendor GEMMs
  out=gpurun_out/${1:-extras}; mkdir -p $out
  ( timeout 600 python tools/eval_bench.py; timeout 600 python tools/eval_bench.py --backbone sigma_small --batch 8 --classes 40 ) > $out/eval_fwd.log 2>&1; grep "^{" $out/eval_fwd.log | cut -c1-300
  ( timeout 600 python bench.py --backbone sigma_base --height 720 --width 1280 --classes 5 --per-gpu-batch 1 --no-cpu-baseline ) > $out/bench_config5.log 2>&1; grep "^{" $out/bench_config5.log | cut -c1-200
  ( timeout 600 python bench.py --graph --no-cpu-baseline ) > $out/bench_b8_graph.log 2>&1; grep "^{" $out/bench_b8_graph.log | cut -c1-200
  ( SIGMA_TUNED_GEMMS=0 timeout 600 python bench.py --no-cpu-baseline ) > $out/bench_b8_untuned.log 2>&1; grep "^{" $out/bench_b8_untuned.log | cut -c1-200
  ;;
*)
  echo "unknown step $step"; exit 2
  ;;
esac
