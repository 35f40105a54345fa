# NCHW-direct decoded first layer: parity tests, then the headline with per-layer times
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "decoded_first_layer" 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK" | tail -5
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --extras 0 2>gpurun_out/b_direct.err | tail -1 > gpurun_out/b_direct.json
python - <<PY
import json
d=json.loads(open("gpurun_out/b_direct.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(d["roofline"].get("layer_ms")); print(d["parity"]["ok"])
PY
