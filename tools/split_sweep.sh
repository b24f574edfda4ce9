#!/bin/bash
# usage: tools/split_sweep.sh "<splits>" <bench args...> : T1 step time vs the SIMD-mate row split (VMP_MIX_SPLIT, % for the older wave)
R=$GRAFT_REPO_ROOT; SP=$1; shift
for sp in $SP; do
  VMP_MIX_SPLIT=$sp python $R/bench.py --no-cpu-baseline --no-extra --reps 11 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $sp  step us %.2f  kernel us %.2f  frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline']['frac']))"
done
