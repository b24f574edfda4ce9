#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; mkdir -p gpurun_out/r06
for m in ${MODES:-none prerun prerun_nograph prestepper prestepper_del}; do MODE=$m timeout 300 python tools/r6_dpg_repro.py 2>&1 | grep "^rank\|Error\|error" | tail -4; done | tee gpurun_out/r06/dpg_repro.txt
