#!/bin/bash
# LDS counters of the T2 backward kernel for a list of library builds (one rocprofv3 pass each): $LIBS, $K, $SMM
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
export REPS=2 N=${N:-250000} K=${K:-16} SMM=${SMM:-1}
cd /tmp; export TMPDIR=/tmp
for lib in ${LIBS:-libvmp_hip_r4a.so libvmp_hip.so}; do
  export VMP_LIB_PATH=$R/vmp-for-svae_amd/lib/$lib
  OUT=$R/gpurun_out/r4_lds_$lib; rm -rf $OUT; mkdir -p $OUT
  timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS --output-format csv -d $OUT -o p2 -- python3 $R/tools/t2_prof_target.py > /dev/null 2>&1
  echo "== $lib K=$K SMM=$SMM"; python3 $R/tools/pmc_summary.py $OUT svae_estep_bwd_ring | grep -v "^void" 
  rm -rf $OUT
done
