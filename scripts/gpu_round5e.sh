set -x
mkdir -p gpurun_out/r5e
timeout 120 scripts/ubench/valu_rate 2>&1 | tee gpurun_out/r5e/ubench_valu_rate.log
timeout 900 python -m pytest tests/test_gpu_parity.py -k "fp16" -q -s 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK" > gpurun_out/r5e/tests.log
tail -25 gpurun_out/r5e/tests.log | cut -c1-700
