#!/bin/bash
# Round-4 evidence run on the GPU box (everything lands in gpurun_out/r04/, summaries are copied to profiles/ afterwards):
#   the driver's bench command, full GPU test suite, rocprofv3 kernel stats of the headline command and of one T3 step,
#   the T1 counter passes, the t2 (GMM / K = 10 / Student-t) and t3 (GMM / Student-t) and smm bench workloads, T2 counters.
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; cd $R
O=$R/gpurun_out/r04; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err; echo "driver cmd rc=$?"
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -a 'passed\|failed' $O/pytest_gpu.log | tail -2
cp gpurun_out/r03_parity_errors.json $O/parity_errors.json 2>/dev/null
python bench.py > $O/bench_default.json 2>> $O/bench.err
cd /tmp; export TMPDIR=/tmp
mkdir -p $O/ks_headline $O/ks_t3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_headline -o k -- python3 $R/bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench_headline_prof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_t3 -o k -- python3 $R/tools/t3_prof_target.py 1000000 > $O/t3_prof.txt 2>/dev/null
cp $(find $O/ks_headline -name 'k_kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
cp $(find $O/ks_t3 -name 'k_kernel_stats.csv' | head -1) $O/t3_kernel_stats.csv
rm -rf $O/ks_headline $O/ks_t3
cd $R
export REPS=5
timeout 400 bash tools/pmc.sh r04/pmc_t1 tools/t1_prof_target.py > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/r04/pmc_t1 pass_xdl > $O/t1_pmc_summary.txt 2>&1
rm -rf gpurun_out/r04/pmc_t1
python bench.py --workload smm --no-extra > $O/bench_smm.json 2>> $O/bench.err
python bench.py --workload t2 --steps 10 --warmup 3 > $O/bench_t2.json 2>> $O/bench.err
python bench.py --workload t2 --k 10 --steps 10 --warmup 3 > $O/bench_t2_k10.json 2>> $O/bench.err
python bench.py --workload t2 --smm --steps 10 --warmup 3 > $O/bench_t2_smm.json 2>> $O/bench.err
python bench.py --workload t3 --steps 5 --warmup 3 > $O/bench_t3.json 2>> $O/bench.err
python bench.py --workload t3 --smm --steps 5 --warmup 3 > $O/bench_t3_smm.json 2>> $O/bench.err
VARIANTS="16_0 10_0 16_1" bash tools/r4_t2_pmc.sh
for v in 16_0 10_0 16_1; do mv gpurun_out/r4_t2_pmc_$v.txt $O/t2_pmc_$v.txt; done
grep "pass_xdl\|pass_kernel\|finalize" $O/bench_kernel_stats.csv | cut -c1-200
head -c 300 $O/bench_driver_cmd.json; echo
for f in t2 t2_k10 t2_smm t3 t3_smm smm; do python3 - <<PY
import json
j=json.load(open('$O/bench_$f.json')); r=j['roofline']
print('$f', 'ms/step %.3f' % j['ms_per_step'], 'kernel %.3f ms' % r['kernel_ms'], 'frac %.3f' % r['frac'])
PY
done
