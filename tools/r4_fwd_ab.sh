#!/bin/bash
# A/B of T2 forward builds on ONE box: tools/t2_time.py per library, K in {16, 10}, two rounds
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
for rep in 1 2; do for cfg in "16 0" "10 0" "16 1"; do set -- $cfg
  for lib in ${LIBS:-libvmp_hip.so}; do
    echo -n "$lib: "; VMP_LIB_PATH=$R/vmp-for-svae_amd/lib/$lib K=$1 SMM=$2 python $R/tools/t2_time.py 2>&1 | tail -1; done; done; done
