mkdir -p gpurun_out/r5_prof; export TMPDIR=/tmp
( time timeout 600 python3 -m pytest tests/test_pointwise_gpu.py tests/test_model_gpu.py -q -p no:cacheprovider -k "pointwise or derived_parameter or hip_graph_replay" ) > gpurun_out/r5_prof/pytest.log 2>&1; grep -v "^  File" gpurun_out/r5_prof/pytest.log | tail -4 | cut -c1-300
SIGMA_HIP_LIB=$PWD/sigma_amd/lib/libsigma_hip_rlprof.so timeout 300 python tools/rowlane_prof.py enc_s2_b16 enc_s0_b16 > gpurun_out/r5_prof/rowlane_prof.jsonl 2> gpurun_out/r5_prof/rowlane_prof.err; cat gpurun_out/r5_prof/rowlane_prof.jsonl; tail -2 gpurun_out/r5_prof/rowlane_prof.err
SCAN_BENCH_ARGS="--pitch 16" bash tools/gpu_pmc.sh r5_prof/pmc enc_s2_b16 all > gpurun_out/r5_prof/pmc.txt 2>&1; grep -A3 "^== " gpurun_out/r5_prof/pmc.txt | grep -v "^--" | cut -c1-700
