mkdir -p gpurun_out/r5_nw8; export TMPDIR=/tmp
( time timeout 900 python3 -m pytest tests/test_scan_gpu.py -q -p no:cacheprovider -x -k "row_lane or full_size or policy" ) > gpurun_out/r5_nw8/pytest.log 2>&1; grep -v "^  File" gpurun_out/r5_nw8/pytest.log | grep -v "^$" | tail -5 | cut -c1-300
for opts in "" "--opt rl_waves=8"; do
  echo "== pitch 16 $opts"
  timeout 300 python tools/scan_bench.py --fine --pitch 16 --iters 10 --shapes enc_s0_b16,enc_s1_b16,enc_s2_b16,enc_s0_b8,enc_s0_b2,enc_s1_b2,enc_s2_b2,enc_s0 $opts 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('%-12s fwd %7.1f us  bwd %7.1f us' % (r['shape'], r['fwd_us'], r.get('bwd_us', 0)))
"
done 2>&1 | tee gpurun_out/r5_nw8/nw8.txt
