# symmetric workgroups (QCNN_OPT_SYM) off / forced / planner at 125 ... 1000 images: per-layer times and the planner's cuts
# (layer: -4 x 1 = k_conv_sym)
mkdir -p gpurun_out
for b in 125 250 500 1000; do for y in 0 2 1; do
echo "QCNN_SYM=$y"; QCNN_SYM=$y QCNN_SPLIT=1 timeout 300 python scripts/layer_times.py $b 12 1
done; done 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK|amdgpu.ids" | tee gpurun_out/sym_sweep.log
