# kernel-trace stats of a short headline run (no counters): per-kernel average durations
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/trace_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/bench.py --steps 5 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 0 ${BENCH_EXTRA:-} > $OUT/bench.json 2> $OUT/err.log
python - <<P
import csv
for r in list(csv.reader(open("$OUT/t_kernel_stats.csv")))[1:14]:
    print("%-60s calls %3s avg %9.1f us" % (r[0].replace("(anonymous namespace)::","")[:60], r[1], float(r[3])/1e3))
P
