#!/bin/bash
# round 6, second GPU pass: sin/cos table A/B, tile stage stamps, tests of the epilogue
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests/test_philox.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2 3; do for lib in libvmp_hip_r5.so libvmp_hip_notab.so libvmp_hip.so; do
  VMP_LIB_PATH=$R/vmp-for-svae_amd/lib/$lib K=16 python tools/r6_fwd_ab.py 2>&1 | tail -1; done; done | tee $O/fwd_ab2.txt
VMP_LIB_PATH=$R/vmp-for-svae_amd/lib/libvmp_hip_ts.so K=16 RNG=1 python tools/fwd_tile_ts.py 2>&1 | tail -3 | tee $O/fwd_tile_ts.txt
