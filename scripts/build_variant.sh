#!/bin/bash
# build_variant.sh <suffix> <source under quantized-cnn_amd/csrc> <extra hipcc flags...>: libqcnn_hip<suffix>.so = the
# current objects with ONE source recompiled under extra flags (timing experiments; select with QCNN_HIP_LIB)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
sfx=$1; src=$2; shift 2
C=$R/quantized-cnn_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $C/$src -o /tmp/variant$sfx.o
objs=""
for o in qcnn_kernels qcnn_sym8 qcnn_half8 qcnn_planner qcnn_glue qcnn_small qcnn_dense qcnn_decoded qcnn_engine qcnn_group; do
  if [ "$o.hip" == "$src" ]; then objs="$objs /tmp/variant$sfx.o"; else objs="$objs $C/$o.hip.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/quantized-cnn_amd/libqcnn_hip$sfx.so $objs -L/opt/rocm/lib -lrccl -lpthread
echo built libqcnn_hip$sfx.so
