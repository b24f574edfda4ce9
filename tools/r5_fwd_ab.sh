#!/bin/bash
# A/B of the in-kernel-noise forward on ONE box: round-4 library (Philox4x32-10, 4 normals per block) vs this build
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
mkdir -p $R/gpurun_out
for rep in 1 2; do for cfg in "16 0" "10 0" "16 1"; do set -- $cfg
  for lib in ${LIBS:-libvmp_hip_r4.so libvmp_hip.so}; do
    echo -n "$lib K=$1 SMM=$2: "; VMP_LIB_PATH=$R/vmp-for-svae_amd/lib/$lib K=$1 SMM=$2 python $R/tools/t2_time.py 2>&1 | tail -2 | tr '\n' ' '; echo; done; done; done
