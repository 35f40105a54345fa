set -x
mkdir -p gpurun_out/r5c
timeout 300 python scripts/diag/f16_fc_diag.py 2>&1 | grep -vE "^layerInd|^\[INFO\]|amdgpu.ids" | tee gpurun_out/r5c/f16_fc_diag.log
