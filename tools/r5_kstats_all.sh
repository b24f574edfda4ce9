#!/bin/bash
# usage: tools/r5_kstats_all.sh <tag> <bench.py arguments...> : rocprofv3 --kernel-trace --stats of a bench.py command, every kernel,
# sorted by total time; the CSV stays in gpurun_out/ks_<tag>/
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; tag=$1; shift; OUT=$R/gpurun_out/ks_$tag; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python3 $R/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/k_kernel_stats.csv')))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for row in rows[:40]:
    print('%-90s calls %5s avg %10.1f us  total %10.1f us' % (row['Name'][:90], row['Calls'], float(row['AverageNs'])/1e3, float(row['TotalDurationNs'])/1e3))
PY
