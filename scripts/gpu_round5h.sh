set -x
mkdir -p gpurun_out/r5h
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK" > gpurun_out/r5h/pytest_gpu.log
tail -12 gpurun_out/r5h/pytest_gpu.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5h/bench.json 2> gpurun_out/r5h/bench.err; echo "bench rc=$?" >> gpurun_out/r5h/bench.err
tail -3 gpurun_out/r5h/bench.err
python - <<'P'
import json
b=json.loads(open('gpurun_out/r5h/bench.json').readline())
for k in ("value","value_tables_only","value_fp16_lut","value_fp16_lut_fp16_sums","value_vgg16","value_shard_125","value_b1","value_via_reference_main","speedup_vs_cpu_baseline"):
    print(k, b.get(k))
print("fp16", json.dumps(b.get("fp16_lut")))
print("vgg parity ok", b.get("vgg16",{}).get("parity",{}).get("ok"))
print("cpu", {k:v for k,v in b.get("cpu_baseline",{}).items() if k not in('sample',)})
print("roof", {k:v for k,v in b["roofline"].items() if not isinstance(v,dict)})
print("roof tab", {k:v for k,v in b["roofline_tables_only"].items() if not isinstance(v,dict)})
P
