#!/bin/bash
# usage: tools/kstats_all.sh <tag> <script> [args...] : rocprofv3 kernel stats, every kernel, sorted by total time
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; TAG=$1; shift; S=$1; shift; OUT=$R/gpurun_out/ks_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python3 $R/$S "$@" > $OUT/stdout.txt 2>&1
tail -2 $OUT/stdout.txt
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/k_kernel_stats.csv')))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for row in rows[:28]:
    print('%-90s calls %5s total %9.2f ms avg %9.1f us' % (row['Name'][:90], row['Calls'], float(row['TotalDurationNs'])/1e6, float(row['AverageNs'])/1e3))
PY
