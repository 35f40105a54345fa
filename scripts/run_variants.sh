for v in ${VARIANTS:-nomfma nostore noadd noread nobuild}; do
  QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip_$v.so python bench.py --steps 5 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['value'], {k:v for k,v in d['roofline']['layer_ms'].items() if 'conv' in k or 'fc' in k})"
done
python bench.py --steps 5 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('base', d['value'], {k:v for k,v in d['roofline']['layer_ms'].items() if 'conv' in k or 'fc' in k})"
