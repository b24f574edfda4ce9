#!/bin/bash
# usage: tools/pmc_cache.sh <outdir-under-gpurun_out> <python-script> : cache-path counters (separate passes, kernel-trace only)
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; OUT=$R/gpurun_out/$1; T=$R/$2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT -o c1 -- python3 $T > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum --output-format csv -d $OUT -o c2 -- python3 $T > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT -o c3 -- python3 $T > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --output-format csv -d $OUT -o c4 -- python3 $T > /dev/null 2>&1
ls $OUT
python3 $R/tools/pmc_summary.py $OUT svae
