#!/bin/bash
# A/B two builds of the library in one session: tools/t2_ab.sh libA.so libB.so
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
for rep in 1 2; do for lib in "$@"; do
  echo "== $lib (rep $rep)"; VMP_LIB_PATH=$R/$lib bash $R/tools/kstats.sh ab tools/t2_prof_target.py | grep svae_estep
done; done
