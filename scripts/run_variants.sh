for v in b3g0 b3g2 b2g0 b0g0; do
  QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip_$v.so python bench.py --steps 5 --warmup 2 --cpu-sample 0 --h2d-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['roofline']['layer_ms'])"
done
