mkdir -p gpurun_out/r5_stem; export TMPDIR=/tmp
( time timeout 900 python3 -m pytest tests/test_model_gpu.py -q -p no:cacheprovider -x -k "patch_embed or fixtures or train_mode" ) > gpurun_out/r5_stem/pytest.log 2>&1; grep -v "^  File" gpurun_out/r5_stem/pytest.log | grep -v "^$" | tail -12 | cut -c1-300
bash tools/gpu.sh bench r5_stem --no-cpu-baseline
