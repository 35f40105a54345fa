# sliding form of the eight-wave kernel: parity, then per-layer times forced / planner (AlexNet 1000 and 125 images, VGG-16)
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "sym8_workgroups" 2>&1 | tail -3
for y in 1 3; do echo "QCNN_SYM8=$y"; QCNN_SYM8=$y python scripts/layer_times.py 1000 10 1 | grep -E "batch|_conv" | cut -c1-250; done
for y in 1 3; do echo "QCNN_SYM8=$y"; QCNN_SYM8=$y python scripts/layer_times.py 125 20 1 | grep -E "batch|_conv" | cut -c1-250; done
for y in 1 3; do echo "QCNN_SYM8=$y VGG16"; QCNN_MODEL=VGG16 QCNN_SYM8=$y python scripts/layer_times.py 1000 2 1 | grep -E "batch|_conv" | cut -c1-420; done
