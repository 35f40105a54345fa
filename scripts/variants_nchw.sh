# timing decomposition of the NCHW-direct decoded first layer (variants built by scripts/build_variant.sh _nv<bits> qcnn_decoded.hip
# -DNCHW_VAR=<bits>; results wrong, timing only)
for v in ${VARIANTS:-"" _nv1 _nv2 _nv4}; do
QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip$v.so python bench.py --steps 6 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); lm=d['roofline']['layer_ms']; print('variant [$v]', d['value'], 'conv1', lm.get('00_conv'))"
done
