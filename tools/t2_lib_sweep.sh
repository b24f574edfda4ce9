#!/bin/bash
# usage: tools/t2_lib_sweep.sh "libA.so libB.so" : T2 kernel times for alternative builds of the library
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
for lib in $1; do
  echo "== $lib"
  VMP_LIB_PATH=$R/$lib timeout 200 bash $R/tools/kstats.sh libsweep2 tools/t2_prof_target.py | grep "svae_estep"
done
