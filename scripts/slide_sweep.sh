mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x -k "sliding" 2>&1 | tail -3
for b in 1000 500 250 125; do QCNN_DEBUG_PLAN=0 QCNN_SLIDE=1 timeout 300 python scripts/layer_times.py $b 8 1; done 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK|amdgpu.ids|plan\]" | tee gpurun_out/slide_sweep.log
QCNN_SLIDE=2 timeout 300 python scripts/layer_times.py 1000 8 1 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK|amdgpu.ids|plan\]" | tee -a gpurun_out/slide_sweep.log
