set -x
mkdir -p gpurun_out/r5n
for lib in "" _anti; do
  echo "variant [$lib]" | tee -a gpurun_out/r5n/anti_phase.log
  QCNN_SYM8=2 QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip$lib.so timeout 300 python scripts/layer_times.py 1000 20 1 2>&1 | grep -vE "^layerInd|^\[INFO\]|amdgpu.ids" | cut -c1-700 | tee -a gpurun_out/r5n/anti_phase.log
done
QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip_anti.so timeout 600 python -m pytest tests/test_gpu_parity.py -k "sym8_workgroups" -q 2>&1 | tail -4
