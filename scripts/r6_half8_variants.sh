# round 6: timing decomposition of k_conv_half8 (variants' results are wrong; timing only)
mkdir -p gpurun_out/r6
for v in 0 1 2 4 8 3 7 15 12; do
  lib=quantized-cnn_amd/libqcnn_hip_h8v$v.so; [ $v == 0 ] && lib=quantized-cnn_amd/libqcnn_hip.so
  echo "H8_VAR=$v"
  QCNN_HIP_LIB=$PWD/$lib QCNN_HALF8=2 timeout 300 python scripts/layer_times.py 1000 10 1 2>&1 | grep -v amdgpu.ids | tail -1
done > gpurun_out/r6/half8_variants.log 2>&1
cat gpurun_out/r6/half8_variants.log
