#!/bin/bash
# round-4 side measurements on one box: new pack kernels + multirank tests, decoder hidden-width sweep (what a 48 + 2 split could
# reach at most), T1 stage stamps (ts build)
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1
O=$R/gpurun_out/r4_misc; mkdir -p $O; cd $R
python -m pytest tests/test_step_glue_gpu.py tests/test_multirank_gpu.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" $O/pytest.txt | tail -3
for U in 32 48 50 64; do python tools/dec_perf.py 262144 $U; done 2>&1 | grep rows | tee $O/dec_u_sweep.txt
VMP_LIB_PATH=$R/vmp-for-svae_amd/lib/libvmp_hip_ts.so python tools/pass_ts.py 2048 125000 1000000 > $O/pass_ts.txt 2>&1
VMP_LIB_PATH=$R/vmp-for-svae_amd/lib/libvmp_hip_ts.so python tools/fin_ts.py > $O/fin_ts.txt 2>&1
tail -4 $O/fin_ts.txt
