#!/bin/bash
# usage: tools/kstats.sh <tag> <script> : rocprofv3 kernel stats (true GPU durations) for our kernels
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; OUT=$R/gpurun_out/ks_$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python3 $R/$2 > /dev/null 2>&1
python3 - <<PY
import csv
for row in csv.DictReader(open('$OUT/k_kernel_stats.csv')):
    n=row['Name']
    if any(t in n for t in ('pass_kernel', 'finalize', 'pivot', 'pack_kernel', 'svae_estep', 'loglike', 'subsample')):
        print('%-70s calls %4s avg %9.1f us  min %9.1f' % (n[28:98], row['Calls'], float(row['AverageNs'])/1e3, float(row['MinNs'])/1e3))
PY
