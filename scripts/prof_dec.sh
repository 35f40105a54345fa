# PMC counters of the decoded first-layer kernel (own passes, no trace domains)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_dec
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --extras 0 --cpu-sample 0 --parity-images 0"
pmc() { n=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -o $n -- $BENCH > /dev/null 2> $OUT/$n.err; }
pmc p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA
pmc p2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD
pmc p3 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
find $OUT -name "*.db" -delete
python $R/scripts/pmc_summary.py $OUT 2>/dev/null | grep -E "^kernel|k_conv_dec|k_conv_aprx<1.3.12.8.2.false" 
tail -2 $OUT/*.err | head -20
