#!/bin/bash
# usage: tools/pmc.sh <outdir-under-gpurun_out> <python-script> ; collects 4 PMC passes (separate runs), kernel-trace only
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; OUT=$R/gpurun_out/$1; T=$R/$2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT -o p1 -- python3 $T > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o p2 -- python3 $T > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $OUT -o p3 -- python3 $T > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --output-format csv -d $OUT -o p4 -- python3 $T > /dev/null 2>&1
ls $OUT
