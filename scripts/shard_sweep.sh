# one GPU's share of a sharded batch: split on/off x streams 1/2, per-layer times
mkdir -p gpurun_out
for b in 125 250 500; do for s in 1 2; do
QCNN_SPLIT=1 timeout 300 python scripts/layer_times.py $b 10 $s
done; done 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK|amdgpu.ids" | tee gpurun_out/shard_sweep.log
for m in 64 128; do QCNN_LRNPOOL_MIN=$m QCNN_SPLIT=1 timeout 300 python scripts/layer_times.py 125 10 1; done 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK|amdgpu.ids" | tee -a gpurun_out/shard_sweep.log
