# per-layer device times of one GPU's share of a sharded batch (125/250/500) and of the full batch
mkdir -p gpurun_out
for b in 125 250 500 1000; do for s in 1 2; do
timeout 300 python scripts/layer_times.py $b 10 $s
done; done 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK" | tee gpurun_out/layer_sweep.log
