#!/bin/bash
# Round-3 evidence run on the GPU box (everything lands in gpurun_out/, summaries are copied to profiles/ afterwards):
#   full GPU test suite, the default bench line, rocprofv3 kernel stats of the headline command and of one T3 step,
#   the T1 counter passes (traffic), and the t2 / t3 / smm bench workloads.
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; cd $R
python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu.log 2>&1; grep -a 'passed\|failed' gpurun_out/r03_pytest_gpu.log | tail -2
python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; tail -c 300 gpurun_out/r03_bench.err
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r03_ks_headline $R/gpurun_out/r03_ks_t3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_ks_headline -o k -- python3 $R/bench.py --no-extra --no-cpu-baseline > $R/gpurun_out/r03_bench_headline_prof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_ks_t3 -o k -- python3 $R/tools/t3_prof_target.py 1000000 > $R/gpurun_out/r03_t3_prof.txt 2>/dev/null
cp $R/gpurun_out/r03_ks_headline/k_kernel_stats.csv $R/gpurun_out/r03_bench_kernel_stats.csv
cp $R/gpurun_out/r03_ks_t3/k_kernel_stats.csv $R/gpurun_out/r03_t3_kernel_stats.csv
rm -rf $R/gpurun_out/r03_ks_headline $R/gpurun_out/r03_ks_t3
cd $R
export REPS=5
timeout 400 bash tools/pmc.sh r03_pmc_t1 tools/t1_prof_target.py > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/r03_pmc_t1 pass_xdl > gpurun_out/r03_t1_pmc_summary.txt 2>&1
rm -rf gpurun_out/r03_pmc_t1
python bench.py --workload smm --no-extra > gpurun_out/r03_bench_smm.json 2>> gpurun_out/r03_bench.err
python bench.py --workload t2 --steps 10 --warmup 2 > gpurun_out/r03_bench_t2.json 2>> gpurun_out/r03_bench.err
python bench.py --workload t3 --steps 3 --warmup 1 > gpurun_out/r03_bench_t3.json 2>> gpurun_out/r03_bench.err
grep "pass_xdl\|pass_kernel\|finalize" gpurun_out/r03_bench_kernel_stats.csv | cut -c1-200
head -c 400 gpurun_out/r03_bench.json
