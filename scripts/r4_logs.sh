# supporting logs of profiles/r4_v10: shard sweep, few-image probe, eight-wave kernels off / forced / planner, the
# NCHW-direct first layer against pack + panel kernel and its timing variants (variant libraries built beforehand:
# scripts/build_variant.sh _nv<bits> qcnn_decoded.hip -DNCHW_VAR=<bits>)
O=gpurun_out/r4; mkdir -p $O
F='^layerInd|^\[INFO\]|^\[CHECK|amdgpu.ids'
bash scripts/shard_sweep.sh > /dev/null 2>&1; cp gpurun_out/shard_sweep.log $O/shard_sweep.log
for b in 1 2; do QCNN_SPLIT=1 timeout 300 python scripts/layer_times.py $b 50 1; done 2>&1 | grep -vE "$F" > $O/b1.log
for b in 125 1000; do for y in 0 2 1; do
echo "QCNN_SYM8=$y"; QCNN_SYM8=$y timeout 300 python scripts/layer_times.py $b 10 1
done; done 2>&1 | grep -vE "$F" > $O/sym8_sweep.log
for m in AlexNet; do for d in 0 1; do
echo "QCNN_DIRECT_DEC=$d"; QCNN_DIRECT_DEC=$d timeout 300 python scripts/layer_times.py 1000 10 1
done; done 2>&1 | grep -vE "$F" > $O/direct_dec.log
echo "QCNN_SYM8=0/2/1 VGG16 1000 images" > $O/vgg16_sym8.log
for y in 0 2 1; do echo "QCNN_SYM8=$y"; QCNN_MODEL=VGG16 QCNN_SYM8=$y timeout 600 python scripts/layer_times.py 1000 2 1; done 2>&1 | grep -vE "$F" >> $O/vgg16_sym8.log
ls quantized-cnn_amd/libqcnn_hip_nv1.so > /dev/null 2>&1 && VARIANTS="_nv1 _nv2 _nv4 _nv6" bash scripts/variants_nchw.sh > $O/variants_nchw.log 2>&1
tail -n +1 $O/*.log | cut -c1-400
# the sliding form of the eight-wave kernel: planner / tile form forced / sliding form forced (AlexNet, VGG-16; 1000 images)
for m in AlexNet VGG16; do for y in 1 2 3; do echo "QCNN_MODEL=$m QCNN_SYM8=$y"; QCNN_MODEL=$m QCNN_SYM8=$y timeout 600 python scripts/layer_times.py 1000 3 1; done; done 2>&1 | grep -vE "$F" > $O/sym8_slide.log
