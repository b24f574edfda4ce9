#!/bin/bash
# round-4 T2 check: parity of the ring kernels (8 <= K <= 16, Student-t theta) + bench lines of the T2 variants
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
O=$R/gpurun_out/r4_t2; mkdir -p $O; cd $R
python -m pytest tests/test_svae_gpu.py -x -q -k "estep_vs_oracle or training_steps or t2_full_size" > $O/pytest_svae.txt 2>&1; echo "svae rc=$?"
tail -5 $O/pytest_svae.txt | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
python -m pytest tests/test_fullsize_gpu.py -x -q -k "t3_training" > $O/pytest_full.txt 2>&1; echo "full rc=$?"
grep -E "passed|failed|Error|assert" $O/pytest_full.txt | tail -8
for v in "" "--k 10" "--smm" "--smm --k 10"; do
  n=$(echo "t2$v" | tr -d ' -')
  python bench.py --workload t2 $v --steps 10 --warmup 3 > $O/bench_$n.json 2> $O/bench_$n.err; echo "bench t2 $v rc=$?"
  python - <<PY
import json
j=json.load(open('$O/bench_$n.json')); r=j['roofline']
print('$n', 'ms/step %.3f' % j['ms_per_step'], 'bwd %.3f ms frac %.3f' % (r['kernel_ms'], r['frac']), 'fwd %.3f ms frac %.3f' % (r['fwd_kernel_ms'], r['fwd_frac']))
PY
done
