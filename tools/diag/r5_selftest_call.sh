mkdir -p gpurun_out/r5_selftest; export TMPDIR=/tmp
( time timeout 900 python3 -m pytest tests/test_scan_gpu.py -q -p no:cacheprovider -x -k "self_test or row_lane or full_size or wave_primitives" ) > gpurun_out/r5_selftest/pytest.log 2>&1; grep -v "^  File" gpurun_out/r5_selftest/pytest.log | grep -v "^$" | tail -12 | cut -c1-400
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -4 | cut -c1-200
