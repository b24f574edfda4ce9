#!/bin/bash
# usage: tools/t1_lib_sweep.sh "libA.so libB.so" : T1 kernel times for alternative builds of the library (paths relative to the repo)
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
for lib in $1; do
  echo "== $lib"
  VMP_LIB_PATH=$R/$lib bash $R/tools/kstats.sh libsweep tools/t1_prof_target.py | grep "pass_kernel"
done
