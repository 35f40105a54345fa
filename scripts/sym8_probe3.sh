for v in 0 2 1; do
  timeout 900 python bench.py --model VGG16 --batch 1000 --steps 2 --warmup 1 --extras 0 --cpu-sample 0 --parity-images 0 --sym8 $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('VGG16 sym8=$v', d['value'], {k:(round(v,2)) for k,v in d['roofline']['layer_ms'].items() if 'conv' in k})"
done
for v in 0 2 1; do
  timeout 300 python bench.py --batch 125 --steps 10 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 0 --sym8 $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('AlexNet 125 images sym8=$v', d['value'], d['ms_per_step'], {k:(round(v,3)) for k,v in d['roofline']['layer_ms'].items() if 'conv' in k})"
done
timeout 300 python bench.py --steps 5 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('AlexNet 1000 default', d['value'], {k:(round(v,3)) for k,v in d['roofline']['layer_ms'].items() if 'conv' in k}, d['parity']['ok'])"
