set -x
mkdir -p gpurun_out/r5o
for lib in "" _rm; do
  for y in 1 2; do
  echo "variant [$lib] QCNN_SYM8=$y" | tee -a gpurun_out/r5o/rowmajor.log
  QCNN_SYM8=$y QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip$lib.so timeout 300 python scripts/layer_times.py 1000 20 1 2>&1 | grep -vE "^layerInd|^\[INFO\]|amdgpu.ids" | cut -c1-700 | tee -a gpurun_out/r5o/rowmajor.log
  done
done
echo "VGG16 [$lib]" | tee -a gpurun_out/r5o/rowmajor.log
QCNN_MODEL=VGG16 QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip_rm.so timeout 600 python scripts/layer_times.py 1000 2 1 2>&1 | grep -vE "^layerInd|^\[INFO\]|amdgpu.ids" | cut -c1-900 | tee -a gpurun_out/r5o/rowmajor.log
QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip_rm.so timeout 900 python -m pytest tests/test_gpu_parity.py -k "sym8 or split_tiles_of_the_eight" -q 2>&1 | tail -4
