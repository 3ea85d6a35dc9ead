mkdir -p gpurun_out/r5_segs; export TMPDIR=/tmp
for opts in "" "--opt rl_segs=2" "--opt rl_segs=3" "--opt rl_segs=4" "--opt rl_waves=8" "--opt rl_waves=16" "--opt rl_waves=16 --opt rl_segs=2"; do
  echo "== pitch 16 $opts"
  timeout 300 python tools/scan_bench.py --fine --pitch 16 --iters 10 --shapes enc_s0_b16,enc_s1_b16 $opts 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('%-12s fwd %7.1f us  bwd %7.1f us' % (r['shape'], r['fwd_us'], r.get('bwd_us', 0)))
"
done 2>&1 | tee gpurun_out/r5_segs/segs.txt
echo "== automatic pitch"; timeout 300 python tools/scan_bench.py --fine --iters 10 --shapes enc_s0_b16,enc_s1_b16 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('%-12s pitch %d fwd %7.1f us  bwd %7.1f us' % (r['shape'], r['ckpt_pitch'], r['fwd_us'], r.get('bwd_us', 0)))
" | tee -a gpurun_out/r5_segs/segs.txt
