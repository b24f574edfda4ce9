#!/bin/bash
# rocprofv3 counters of the T2 backward kernels (ring forms), separate passes, N rows from $N (default 250000), variants K / SMM
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
export REPS=2 N=${N:-250000}
cd /tmp; export TMPDIR=/tmp
T=$R/tools/t2_prof_target.py
for v in ${VARIANTS:-16_0 10_0 16_1}; do
  export K=${v%_*} SMM=${v#*_}
  OUT=$R/gpurun_out/r4_t2_pmc_$v; rm -rf $OUT; mkdir -p $OUT
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT -o p1 -- python3 $T > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA --output-format csv -d $OUT -o p2 -- python3 $T > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o p3 -- python3 $T > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum --output-format csv -d $OUT -o c1 -- python3 $T > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $OUT svae_estep_bwd > $R/gpurun_out/r4_t2_pmc_$v.txt 2>&1
  rm -rf $OUT
done
