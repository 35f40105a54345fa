#!/bin/bash
# Compile qcnn_kernels.hip with -save-temps into /tmp/isa and report, per hot kernel:
# VGPRs, SGPRs, scratch bytes, ds_read count, s_waitcnt lgkmcnt(0) count (serialisation indicator) and
# v_readlane/v_writelane (SGPR spills).
set -e
mkdir -p /tmp/isa && cd /tmp/isa && rm -f *.s log.txt
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -save-temps \
  -Rpass-analysis=kernel-resource-usage -c /root/repo/quantized-cnn_amd/csrc/qcnn_kernels.hip -o /tmp/isa/k.o > log.txt 2>&1 || { grep error log.txt | head; exit 1; }
S=qcnn_kernels-hip-amdgcn-amd-amdhsa-gfx950.s
for k in $(grep -o "^_ZN12_GLOBAL__N_1[0-9]*k_[a-z]*_aprxI[A-Za-z0-9]*EEv[0-9A-Za-z]*" $S | sort -u | grep -v "${SKIP:-Li0ELi1EEEv}"); do
  awk "/^${k}:/,/s_endpgm/" $S > x.s
  v=$(grep -A3 "Function Name: ${k}" log.txt | grep " VGPRs:" | head -1 | sed 's/.*VGPRs: \([0-9]*\).*/\1/')
  sg=$(grep -A3 "Function Name: ${k}" log.txt | grep " SGPRs:" | head -1 | sed 's/.*SGPRs: \([0-9]*\).*/\1/')
  sc=$(grep -A12 "Function Name: ${k}" log.txt | grep "ScratchSize" | head -1 | sed 's/.*: \([0-9]*\) .*/\1/')
  echo "${k#_ZN12_GLOBAL__N_1} vgpr=$v sgpr=$sg scratch=$sc lines=$(wc -l < x.s) ds_read=$(grep -c ds_read_b64 x.s) lgkm0=$(grep -c 'lgkmcnt(0)' x.s) readlane=$(grep -c v_readlane x.s) writelane=$(grep -c v_writelane x.s)"
done
