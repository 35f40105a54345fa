for v in "" _lpf; do QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip$v.so python bench.py --steps 6 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); lm=d['roofline']['layer_ms']; print('variant [$v]', d['value'], 'lrn1', lm.get('02_lorn'), 'lrn2', lm.get('06_lorn'), d['parity']['ok'])"; done
