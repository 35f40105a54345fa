# round 5, call 1: the new parity tests first, then the whole GPU tier, then the bench line
set -x
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests/test_gpu_vgg16_config3.py tests/test_gpu_soak_modes.py tests/test_gpu_parity.py -k "vgg16_batch_1000 or forced_kernel or very_end" -q -x 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK" > gpurun_out/r5a/new_tests.log
tail -15 gpurun_out/r5a/new_tests.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK" > gpurun_out/r5a/pytest_gpu.log
tail -15 gpurun_out/r5a/pytest_gpu.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err; echo "bench rc=$?" >> gpurun_out/r5a/bench.err
tail -5 gpurun_out/r5a/bench.err
python - <<'P'
import json
b=json.loads(open('gpurun_out/r5a/bench.json').readline())
for k in ("value","value_tables_only","value_fp16_lut","value_vgg16","value_shard_125","predicted_strong_scaling","speedup_vs_cpu_baseline"):
    print(k, b.get(k))
print("vgg parity", b.get("vgg16",{}).get("parity"))
print("cpu", {k:v for k,v in b.get("cpu_baseline",{}).items() if k!='sample'})
print("layer_ms", b["roofline"]["layer_ms"])
print("tab layer_ms", b.get("roofline_tables_only",{}).get("layer_ms"))
P
