set -x
mkdir -p gpurun_out/r5g
timeout 600 python -m pytest tests/test_gpu_parity.py -k "split_tiles" -q -x 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK" > gpurun_out/r5g/tests.log
tail -25 gpurun_out/r5g/tests.log | cut -c1-500
for b in 125 250; do
QCNN_DEBUG_PLAN=1 timeout 300 python scripts/layer_times.py $b 20 1 2>&1 | grep -vE "^layerInd|^\[INFO\]|amdgpu.ids" | cut -c1-900 | tee -a gpurun_out/r5g/layer_times.log
done
QCNN_SYM8=2 timeout 300 python scripts/layer_times.py 125 20 1 2>&1 | grep -vE "^layerInd|^\[INFO\]|amdgpu.ids" | cut -c1-900 | tee -a gpurun_out/r5g/layer_times.log
