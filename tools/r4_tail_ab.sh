#!/bin/bash
# A/B of the ELBO tail (dec_reduce_tail_kernel) inside the T3 step: rocprofv3 kernel stats per library
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
cd /tmp; export TMPDIR=/tmp
for lib in ${LIBS:-libvmp_hip.so}; do
  rm -rf /tmp/ks_$lib; mkdir -p /tmp/ks_$lib
  VMP_LIB_PATH=$R/vmp-for-svae_amd/lib/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$lib -o k -- python3 $R/tools/t3_prof_target.py 1000000 > /dev/null 2>&1
  python3 - "$lib" "$(find /tmp/ks_$lib -name 'k_kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[2])))
print(sys.argv[1], ' '.join('%s %.1f us' % (r['Name'].split('::')[-1].split('(')[0][:24], float(r['AverageNs']) / 1e3)
                            for r in rows if any(k in r['Name'] for k in ('dec_reduce_tail', 'subsample', 'elbo_tail'))))
PY
done
