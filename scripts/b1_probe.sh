for b in 1 2; do QCNN_SPLIT=1 timeout 300 python scripts/layer_times.py $b 50 1; done 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK|amdgpu.ids"
