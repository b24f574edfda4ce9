#!/bin/bash
# usage: tools/pmc_small.sh <outdir> : two PMC passes over the minibatch-64 training step; per-kernel means of the counters
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; OUT=$R/gpurun_out/$1; T=$R/tools/t3_small_prof.py; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT -o p1 -- python3 $T 64 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT -o p2 -- python3 $T 64 > /dev/null 2>&1
python3 - <<PY
import collections, csv, glob
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        d[row['Kernel_Name'][:60]][row['Counter_Name']].append(float(row['Counter_Value']))
with open('$OUT/summary.txt', 'w') as o:
    for name in d:
        if 'anonymous' not in name: continue
        o.write(name + '\n')
        for c, v in sorted(d[name].items()):
            o.write('    %-24s %14.1f (n=%d)\n' % (c, sum(v) / len(v), len(v)))
PY
rm -f $OUT/*counter_collection.csv $OUT/*kernel_trace.csv
cat $OUT/summary.txt
