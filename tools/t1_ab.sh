#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
for rep in 1 2; do for lib in "$@"; do
  echo "== $lib (rep $rep)"; VMP_LIB_PATH=$R/$lib bash $R/tools/kstats.sh ab1 tools/t1_prof_target.py | grep "true, true\|finalize"
  VMP_LIB_PATH=$R/$lib python $R/bench.py --no-cpu-baseline --no-extra --steps 200 --warmup 20 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench ms_per_step', round(d['ms_per_step']*1e3,1), 'us  kernel', round(d['roofline']['kernel_ms']*1e3,1))"
done; done
