# timing decomposition of k_conv_sym8 (results of the variants are wrong; timing only): which ingredient of a stage costs what
set -x
mkdir -p gpurun_out
for v in ${VARIANTS:-1 2 4 3 6 5 7}; do
  QCNN_HIP_LIB=$PWD/quantized-cnn_amd/libqcnn_hip_s8v$v.so timeout 300 python bench.py --steps 5 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 0 --sym8 ${SYM8:-2} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('S8_VAR=$v', d['value'], {k:v for k,v in d['roofline']['layer_ms'].items() if 'conv' in k})" | tee -a gpurun_out/variants_sym8.log
done
timeout 300 python bench.py --steps 5 --warmup 2 --extras 0 --cpu-sample 0 --parity-images 0 --sym8 ${SYM8:-2} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('base', d['value'], {k:v for k,v in d['roofline']['layer_ms'].items() if 'conv' in k})" | tee -a gpurun_out/variants_sym8.log
