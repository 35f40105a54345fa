mkdir -p gpurun_out
for b in 125 250 500; do for s in 1; do
QCNN_SPLIT=1 timeout 300 python scripts/layer_times.py $b 10 $s
done; done 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK|amdgpu.ids" | tee gpurun_out/shard_sweep.log
