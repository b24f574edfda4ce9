#!/bin/bash
# T2 backward before/after (round 3): rocprofv3 counters of the round-2 kernel (libvmp_hip_r2.so, built with -DVMP_T2_RING=0)
# and the LDS-ring kernel, same box, separate passes (kernel-trace only), N rows from $N (default 250000).
# Output: gpurun_out/r3_t2_pmc_{old,ring}.txt
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
export REPS=2 N=${N:-250000}
cd /tmp; export TMPDIR=/tmp
for v in old ring; do
  if [ $v = old ]; then export VMP_LIB_PATH=$R/vmp-for-svae_amd/lib/libvmp_hip_r2.so; else unset VMP_LIB_PATH; fi
  OUT=$R/gpurun_out/r3_t2_pmc_$v; rm -rf $OUT; mkdir -p $OUT
  T=$R/tools/t2_prof_target.py
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT -o p1 -- python3 $T > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $OUT -o p3 -- python3 $T > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o p4 -- python3 $T > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $OUT -o c1 -- python3 $T > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $OUT svae_estep_bwd > $R/gpurun_out/r3_t2_pmc_$v.txt 2>&1
  rm -rf $OUT
done
