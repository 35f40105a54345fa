set -x
mkdir -p gpurun_out/r5p
QCNN_MODEL=VGG16 timeout 600 python scripts/layer_times.py 1000 2 1 2>&1 | grep -vE "^layerInd|^\[INFO\]|amdgpu.ids" | cut -c1-400 | tee gpurun_out/r5p/vgg_forward_host.log
timeout 300 python scripts/layer_times.py 1000 5 1 2>&1 | grep -vE "^layerInd|^\[INFO\]|amdgpu.ids" | cut -c1-300 | tee -a gpurun_out/r5p/vgg_forward_host.log
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_group.py tests/test_host_mirror.py -m gpu -q 2>&1 | tail -4
