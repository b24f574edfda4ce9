#!/bin/bash
# usage: tools/t1_n_sweep.sh "N1 N2 ..." : kernel times vs N (fixed cost vs slope)
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
for n in $1; do
  echo "== N $n"
  N=$n bash $R/tools/kstats.sh nsweep tools/t1_prof_target.py | grep "pass_kernel"
done
