#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; mkdir -p gpurun_out/r06
for m in base nopre sep_pools sync_between eager onegraph; do MODE=$m timeout 300 python tools/r6_dpg_linalg.py 2>&1 | grep "rank\|Error\|error" | tail -12; done | tee gpurun_out/r06/dpg_linalg.txt
