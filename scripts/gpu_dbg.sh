# timing experiments: per-layer ms with parts of the conv kernel disabled (results are WRONG on purpose)
mkdir -p gpurun_out
for d in ${DBGS:-0 1 2 4 8 3 6 7 15}; do
  QCNN_DBG=$d timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --h2d-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); lm=d['roofline']['layer_ms']
print('dbg=$d', d['ms_per_step'], {k:v for k,v in lm.items() if 'conv' in k or 'fcnt' in k})"
done | tee gpurun_out/dbg.log
