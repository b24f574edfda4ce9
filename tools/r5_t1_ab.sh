#!/bin/bash
# T1 step: one-launch form vs the two launches it fuses, same box, same process settings (bench.py headline protocol)
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
for rep in 1 2; do for wl in gmm smm; do for ol in 0 1; do
  echo -n "$wl one_launch=$ol: "; VMP_T1_ONE_LAUNCH=$ol python $R/bench.py --workload $wl --steps ${STEPS:-20} --warmup 5 --no-extra --no-cpu-baseline --no-traffic ${N:+--n $N} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f us/step  kernel %.2f us  frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline']['frac']))"
done; done; done
