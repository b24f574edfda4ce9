#!/bin/bash
# usage: tools/t1_plan_sweep.sh "256x8 512x6 ..." : T1 kernel times for each (blocks x waves) launch plan
R=$GRAFT_REPO_ROOT
for cfg in $1; do
  b=${cfg%x*}; w=${cfg#*x}
  echo "== blocks $b nw $w"
  VMP_MIX_BLOCKS=$b VMP_MIX_NW=$w bash $R/tools/kstats.sh sweep tools/t1_prof_target.py | grep "true, true\|finalize"
done
