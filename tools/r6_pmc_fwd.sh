#!/bin/bash
# one counter pass of the T2 forward (in-kernel noise, no epilogue) per library: VALU instructions / active cycles / waits
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
cd /tmp; export TMPDIR=/tmp
for lib in ${LIBS:-libvmp_hip_r5.so libvmp_hip_notab.so libvmp_hip.so}; do
  OUT=$R/gpurun_out/r06/pmc_fwd_$lib; mkdir -p $OUT
  VMP_LIB_PATH=$R/vmp-for-svae_amd/lib/$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT -o p1 -- python3 $R/tools/t2_fwd_rng_prof.py > /dev/null 2>&1
  VMP_LIB_PATH=$R/vmp-for-svae_amd/lib/$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC --output-format csv -d $OUT -o p2 -- python3 $R/tools/t2_fwd_rng_prof.py > /dev/null 2>&1
  echo "== $lib"; python3 $R/tools/pmc_summary.py $OUT svae_estep_fwd4
  rm -rf $OUT
done
