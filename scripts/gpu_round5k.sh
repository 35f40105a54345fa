set -x
mkdir -p gpurun_out/r5k
for lib in "" _g2_4 _g4_2; do
  echo "variant [$lib]" | tee -a gpurun_out/r5k/gather_depth.log
  QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip$lib.so timeout 300 python scripts/layer_times.py 1000 10 1 2>&1 | grep -vE "^layerInd|^\[INFO\]|amdgpu.ids" | cut -c1-700 | tee -a gpurun_out/r5k/gather_depth.log
done
