# eight-wave symmetric workgroups: parity tests, then the headline with QCNN_OPT_SYM8 off / forced / forced + staggered / planner
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "sym8" 2>&1 | tail -15 > gpurun_out/sym8_tests.log
timeout 900 python -m pytest tests/test_gpu_shipped_params.py -q 2>&1 | tail -15 >> gpurun_out/sym8_tests.log
for v in 0 2 6 1; do
  timeout 600 python bench.py --steps 5 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 4 --sym8 $v > gpurun_out/bench_sym8_$v.json 2> gpurun_out/bench_sym8_$v.err
done
cat gpurun_out/sym8_tests.log
python - <<'PY'
import json
for v in (0, 2, 6, 1):
    try:
        d = json.loads(open("gpurun_out/bench_sym8_%d.json" % v).readline())
        print(v, d["value"], d["roofline"]["layer_ms"], d.get("parity", {}).get("ok"))
    except Exception as e:
        print(v, "failed", e, open("gpurun_out/bench_sym8_%d.err" % v).read()[-600:])
PY
