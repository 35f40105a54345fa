# rocprofv3 passes for profiles/: kernel trace + stats, then PMC counters in their own runs (CSV output).
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 5 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 0 ${BENCH_EXTRA:-}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err
pmc() { n=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -o $n -- $BENCH > /dev/null 2> $OUT/$n.err; }
pmc pmc1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
pmc pmc2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA
pmc pmc3 FETCH_SIZE
pmc pmc4 WRITE_SIZE
pmc pmc5 GRBM_GUI_ACTIVE GRBM_COUNT
find $OUT -name "*.db" -delete
find $OUT -type f -size +8M -delete
find $OUT -type f | xargs ls -la
tail -3 $OUT/*.err
