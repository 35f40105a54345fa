set -x
mkdir -p gpurun_out
rocminfo | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > gpurun_out/rocminfo.txt
nproc >> gpurun_out/rocminfo.txt; lscpu | grep "Model name" >> gpurun_out/rocminfo.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_EXTRA:-} 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK" > gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tail -5 gpurun_out/smoke.log; tail -30 gpurun_out/pytest_gpu.log; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
