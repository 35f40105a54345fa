# round 5, call 2: VGG-16 configs[3] test (device-resident), the fp16 table-storage kernels: parity + per-layer times
set -x
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_vgg16_config3.py tests/test_gpu_parity.py -k "vgg16_batch_1000 or fp16" -q -x -s 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK" > gpurun_out/r5b/tests.log
tail -25 gpurun_out/r5b/tests.log
for lut in 1 2; do
  QCNN_LUT=$lut QCNN_DECODE=0 timeout 300 python scripts/layer_times.py 1000 10 1 2>&1 | grep -vE "^layerInd|^\[INFO\]|amdgpu.ids" | cut -c1-900 | tee -a gpurun_out/r5b/layer_times.log
done
QCNN_LUT=2 QCNN_DECODE=0 QCNN_SYM8=0 timeout 300 python scripts/layer_times.py 1000 10 1 2>&1 | grep -vE "^layerInd|^\[INFO\]|amdgpu.ids" | cut -c1-900 | tee -a gpurun_out/r5b/layer_times.log
