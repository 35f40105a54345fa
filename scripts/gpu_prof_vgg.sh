# rocprofv3 kernel trace + one PMC pass of a VGG-16 forward (1000 images, synthetic parameters) for profiles/r4_vgg16
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_vgg
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/layer_times.py 1000 2 1"
QCNN_MODEL=VGG16 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/layer_times.log 2> $OUT/trace.err
QCNN_MODEL=VGG16 timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -o pmc1 -- $CMD > /dev/null 2> $OUT/pmc1.err
find $OUT -name "*.db" -delete
find $OUT -type f -size +8M -delete
grep -vE "^layerInd|^\[INFO\]|amdgpu.ids" $OUT/layer_times.log | cut -c1-400
head -12 $OUT/trace/trace_kernel_stats.csv | cut -c1-200
