# round 6: parity + layer times of the half-panel eight-wave kernel (QCNN_OPT_HALF8) against the round-5 kernels
# MODES: QCNN_OPT_HALF8 values to time (0 off, 1 planner, 2 forced tile form, 3 forced sliding form); NTS: tables per barrier period
set -x
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_half8.py -x -q 2>&1 | tail -15 > gpurun_out/r6/half8_tests.log
for nt in ${NTS:-2}; do
for h in ${MODES:-0 2 3 1}; do
  echo "== QCNN_OPT_HALF8=$h tables per period=$nt"
  for b in ${BATCHES:-1000 125}; do
    QCNN_HALF8_NT=$nt QCNN_HALF8=$h timeout 300 python scripts/layer_times.py $b 20 1
  done
  QCNN_HALF8_NT=$nt QCNN_MODEL=VGG16 QCNN_HALF8=$h timeout 600 python scripts/layer_times.py 1000 2 1
done; done 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/half8_layer_times.log
cat gpurun_out/r6/half8_tests.log gpurun_out/r6/half8_layer_times.log
