# round 5, final: GPU tier, the driver's bench command, the rocprofv3 passes (profiles/r5_v11, profiles/r5_vgg16)
set -x
mkdir -p gpurun_out/r5z
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK" > gpurun_out/r5z/pytest_gpu.log
tail -6 gpurun_out/r5z/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5z/smoke.log 2>&1; tail -2 gpurun_out/r5z/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5z/bench.json 2> gpurun_out/r5z/bench.err; echo "bench rc=$?" >> gpurun_out/r5z/bench.err
tail -3 gpurun_out/r5z/bench.err
bash scripts/gpu_prof.sh > gpurun_out/prof.log 2>&1
bash scripts/gpu_prof_vgg.sh > gpurun_out/prof_vgg.log 2>&1
python - <<'P'
import json
b=json.loads(open('gpurun_out/r5z/bench.json').readline())
print({k:b[k] for k in b if k.startswith('value')})
print("parity", b['parity']['ok'], "vgg parity", b['vgg16']['parity']['ok'])
P
