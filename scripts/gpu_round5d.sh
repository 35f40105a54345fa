# round 5, call 4: the fp16 study kernels (storage, storage + sums): parity tests, per-layer times
set -x
mkdir -p gpurun_out/r5d
timeout 900 python -m pytest tests/test_gpu_parity.py -k "fp16" -q -x -s 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK" > gpurun_out/r5d/tests.log
tail -25 gpurun_out/r5d/tests.log | cut -c1-600
for lut in 1 2 3; do
  QCNN_LUT=$lut QCNN_DECODE=0 timeout 300 python scripts/layer_times.py 1000 10 1 2>&1 | grep -vE "^layerInd|^\[INFO\]|amdgpu.ids" | cut -c1-900 | tee -a gpurun_out/r5d/layer_times.log
done
