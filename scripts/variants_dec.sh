# timing decomposition of the decoded first layer (variants built by scripts/build_variant.sh; results wrong, timing only)
for v in ${VARIANTS:-"" _dv1 _dv2 _dv3}; do
[ "$v" == "base" ] && v=""
QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip$v.so python bench.py --steps 6 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); lm=d['roofline']['layer_ms']; print('variant [$v]', d['value'], 'conv1', lm.get('00_conv'), 'conv2', lm.get('04_conv'))"
done
