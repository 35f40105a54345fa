# supporting logs of profiles/r5_v11 and profiles/r5_vgg16 (run on the GPU box from the repository root).  Variant libraries are built
# beforehand on the build host: scripts/build_variant.sh _anti qcnn_sym8.hip -DS8_ANTI=1; ... _rm qcnn_sym8.hip -DS8_ROWMAJOR=1; the
# look-up statement variants by regenerating quantized-cnn_amd/csrc/qcnn_sym8_gather.h (scripts/gen/gen_sym8_gather.py 2 4 / 4 2) first.
O=gpurun_out/r5; mkdir -p $O
F='^layerInd|^\[INFO\]|^\[CHECK|amdgpu.ids'
LT="python scripts/layer_times.py"
# every layer through tables: f32 tables / fp16 tables / fp16 tables + fp16 sums
for lut in 1 2 3; do QCNN_LUT=$lut QCNN_DECODE=0 timeout 300 $LT 1000 10 1; done 2>&1 | grep -vE "$F" > $O/fp16_layer_times.log
# one and two panels with the planner's predictions; one panel with the eight-wave kernels forced (their tiles split, §3.6)
for b in 125 250; do QCNN_DEBUG_PLAN=1 timeout 300 $LT $b 20 1; done 2>&1 | grep -vE "$F" > $O/shard_sym8_split.log
QCNN_SYM8=2 timeout 300 $LT 125 20 1 2>&1 | grep -vE "$F" >> $O/shard_sym8_split.log
# micro-benchmark of the accumulate instructions; fc7 in isolation, fp16-table kernel against emulation and oracle
timeout 120 scripts/ubench/valu_rate > $O/ubench_valu_rate.log 2>&1
timeout 300 python scripts/diag/f16_fc_diag.py 2>&1 | grep -vE "$F" > $O/f16_fc_diag.log
# race hunt over every kernel family (incl. the fp16 forms and the split eight-wave tiles)
timeout 900 python scripts/soak_modes.py 30 2>&1 | grep -vE "$F" > $O/soak_modes.log
# variant libraries (where present): opposite phases, row-major f32 table, look-up pipeline depth, fused FC statements
for v in _anti _rm _g2_4 _g4_2 _fc12; do
  [ -f quantized-cnn_amd/libqcnn_hip$v.so ] || continue
  for lib in "" $v; do echo "variant [$lib]"; QCNN_SYM8=2 QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip$lib.so timeout 300 $LT 1000 20 1; done 2>&1 | grep -vE "$F" > $O/variant$v.log
done
# VGG-16: resident / host-fed (first and second call)
QCNN_MODEL=VGG16 timeout 600 $LT 1000 2 1 2>&1 | grep -vE "$F" > $O/vgg_forward_host.log
tail -n +1 $O/*.log | cut -c1-300
