#!/bin/bash
# usage: tools/kseq.sh <tag> <marker-kernel-substring> <script> [args...] : rocprofv3 kernel trace; prints the ordered kernel
# sequence between the last two launches of the marker kernel (= one step of an iterative script)
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; TAG=$1; shift; MARK=$1; shift; S=$1; shift; OUT=$R/gpurun_out/kseq_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o k -- python3 $R/$S "$@" > $OUT/stdout.txt 2>&1
tail -2 $OUT/stdout.txt
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/k_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if '$MARK' in r['Kernel_Name']]
lo, hi = idx[-2], idx[-1]
t0 = int(rows[lo]['Start_Timestamp'])
with open('$OUT/seq.txt', 'w') as f:
    for r in rows[lo:hi]:
        f.write('%8.1f %6.1f  %s\n' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:120]))
print(open('$OUT/seq.txt').read())
rm = None
PY
rm -f $OUT/k_kernel_trace.csv
