for v in nostage nobuild nogather; do
  QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip_$v.so python bench.py --batch 1 --steps 30 --warmup 5 --extras 0 --cpu-sample 0 --parity-images 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], {k:v for k,v in d['roofline']['layer_ms'].items() if 'conv' in k})"
done
