# round 6, final: GPU tier, smoke, the driver's bench command, the rocprofv3 passes (profiles/r6_v12, profiles/r6_vgg16)
set -x
mkdir -p gpurun_out/r6z
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK" > gpurun_out/r6z/pytest_gpu.log
tail -6 gpurun_out/r6z/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6z/smoke.log 2>&1; tail -2 gpurun_out/r6z/smoke.log
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r6z/bench.json 2> gpurun_out/r6z/bench.err; echo "bench rc=$?" >> gpurun_out/r6z/bench.err
tail -3 gpurun_out/r6z/bench.err
bash scripts/gpu_prof.sh > gpurun_out/prof.log 2>&1
bash scripts/gpu_prof_vgg.sh > gpurun_out/prof_vgg.log 2>&1
for b in 1000 500 250 125; do for s in 1 2; do python scripts/layer_times.py $b 20 $s 2>&1 | grep -v amdgpu; done; done > gpurun_out/r6z/streams_sweep.log 2>&1
python - <<'P'
import json
b=json.loads(open('gpurun_out/r6z/bench.json').readline())
print({k:b[k] for k in b if k.startswith('value') or k.startswith('alg_')})
print("parity", b['parity']['ok'], "vgg parity", b['vgg16']['parity']['ok'])
P
