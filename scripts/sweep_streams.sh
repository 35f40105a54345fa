for b in 125 250 500 1000; do for s in 1 2 3; do
python bench.py --batch $b --streams $s --steps 8 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('batch $b streams $s', d['value'], d['ms_per_step'])"
done; done
