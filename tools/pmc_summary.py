"""Summarise rocprofv3 counter_collection CSVs: mean of each counter per kernel name."""
import collections, csv, glob, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name'][:70]
        d[name][row['Counter_Name']].append(float(row['Counter_Value']))
for name in d:
    if 'pass_kernel' not in name and 'finalize' not in name and (len(sys.argv) < 3 or sys.argv[2] not in name):
        continue
    print(name)
    for c, v in sorted(d[name].items()):
        print('    %-28s %16.1f  (n=%d)' % (c, sum(v) / len(v), len(v)))
