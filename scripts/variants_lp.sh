for v in "" _lp316 _lp232 _lp416; do
QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip$v.so python bench.py --steps 6 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); lm=d['roofline']['layer_ms']; print('variant [$v]', d['value'], 'lrn1+pool', lm.get('02_lorn'), 'lrn2+pool', lm.get('06_lorn'), 'parity', d['parity']['ok'])"
done
