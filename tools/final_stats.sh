#!/bin/bash
# rocprofv3 kernel stats of the headline command (bench.py, T1 only) and of one T3 step; CSVs are copied to profiles/ by hand
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r02_ks_headline $R/gpurun_out/r02_ks_t3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_ks_headline -o k -- python3 $R/bench.py --no-extra --no-cpu-baseline > $R/gpurun_out/r02_bench_headline_prof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_ks_t3 -o k -- python3 $R/tools/t3_prof_target.py 1000000 > $R/gpurun_out/r02_t3_prof.txt 2>/dev/null
grep "pass_kernel\|finalize" $R/gpurun_out/r02_ks_headline/k_kernel_stats.csv | cut -c1-220
head -6 $R/gpurun_out/r02_ks_t3/k_kernel_stats.csv | cut -c1-200
tail -1 $R/gpurun_out/r02_t3_prof.txt
