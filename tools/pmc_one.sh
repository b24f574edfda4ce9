#!/bin/bash
# usage: tools/pmc_one.sh <outdir-under-gpurun_out> <python-script> <filter> <counter> [counter...] : ONE counter pass, 240 s cap
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; OUT=$R/gpurun_out/$1; T=$R/$2; F=$3; shift 3; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o one -- python3 $T > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $OUT $F
