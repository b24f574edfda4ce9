#!/bin/bash
# T2 bench lines of the four variants (no parity run)
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
O=$R/gpurun_out/r4_t2; mkdir -p $O; cd $R
for v in "" "--k 10" "--smm" "--smm --k 10"; do
  n=$(echo "t2$v" | tr -d ' -')
  python bench.py --workload t2 $v --steps 10 --warmup 3 > $O/bench_$n.json 2> $O/bench_$n.err; echo "bench t2 $v rc=$?"
  python - <<PY
import json
j=json.load(open('$O/bench_$n.json')); r=j['roofline']
print('$n', 'ms/step %.3f' % j['ms_per_step'], 'bwd %.3f ms frac %.3f' % (r['kernel_ms'], r['frac']), 'fwd %.3f ms frac %.3f' % (r['fwd_kernel_ms'], r['fwd_frac']))
PY
done
