#!/bin/bash
# A/B of the ring backward on ONE box: LIBS="libvmp_hip.so libvmp_hip_<variant>.so ..." CFGS="K,SMM ..." (default: {16,10} x {Gaussian, Student-t})
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
mkdir -p $R/gpurun_out
for rep in 1 2; do for cfg in ${CFGS:-16,0 10,0 16,1 10,1}; do k=${cfg%,*}; smm=${cfg#*,}
  for lib in ${LIBS:-libvmp_hip.so}; do
    echo -n "$lib: "; VMP_LIB_PATH=$R/vmp-for-svae_amd/lib/$lib K=$k SMM=$smm python3 $R/tools/t2_time.py 2>&1 | tail -1; done; done; done
