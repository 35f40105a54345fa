timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "sym8" 2>&1 | tail -8
for v in 0 1; do
  timeout 600 python bench.py --steps 5 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 4 --sym8 $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('sym8=$v', d['value'], {k:v for k,v in d['roofline']['layer_ms'].items() if 'conv' in k or 'fc' in k}, d['parity']['ok'], d['parity']['max_rel_err_prob'])"
done
