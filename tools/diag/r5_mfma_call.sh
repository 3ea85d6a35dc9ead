mkdir -p gpurun_out/r5_mfma; export TMPDIR=/tmp
( time timeout 900 python3 -m pytest tests/test_scan_gpu.py -q -p no:cacheprovider -x -k "row_lane or full_size or policy" ) > gpurun_out/r5_mfma/pytest.log 2>&1; grep -v "^  File" gpurun_out/r5_mfma/pytest.log | grep -v "^$" | tail -5 | cut -c1-300
SCAN_BENCH_ARGS="--pitch 16" bash tools/gpu.sh scanbench r5_mfma enc_s2_b16,enc_s2_b2,enc_s0,enc_s0_b16,enc_s1_b16,cromb_s0_b8,dec_s0,conmb_s0 "" _nomfma
