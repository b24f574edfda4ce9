#!/bin/bash
# usage: tools/t2_lib_ab.sh <K> lib1.so lib2.so ... : T2 forward / backward timings (tools/t2_time.py) per library build, two rounds
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; K=$1; shift
for rep in 1 2; do for lib in "$@"; do echo "== $lib"; VMP_LIB_PATH=$R/vmp-for-svae_amd/lib/$lib K=$K python $R/tools/t2_time.py 2>&1 | tail -1; done; done
