# headline only + per-layer times (no extras)
python bench.py --steps 8 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('value', d['value'], 'ms', d['ms_per_step'], 'parity', d['parity']['ok'], d['parity']['max_rel_err_prob']); print(d['roofline']['layer_ms'])"
