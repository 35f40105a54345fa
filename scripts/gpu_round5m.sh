set -x
mkdir -p gpurun_out/r5m
for lib in "" _fc12; do
  echo "variant [$lib]" | tee -a gpurun_out/r5m/fc_fused.log
  QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip$lib.so timeout 300 python scripts/layer_times.py 1000 20 1 2>&1 | grep -vE "^layerInd|^\[INFO\]|amdgpu.ids" | cut -c1-700 | tee -a gpurun_out/r5m/fc_fused.log
done
QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip_fc12.so timeout 600 python -m pytest tests/test_gpu_parity.py -k "fc_sym8" -q 2>&1 | tail -3
