#!/bin/bash
# round 6, first GPU pass: the new noise stream + epilogue - tests, then forward / tail A/B against the round-5 library on this box
R=$(cd "$(dirname "$0")/.." && pwd); [ -n "$R" ] || exit 1; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_philox.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "t2_step" 2>&1 | tail -5
for rep in 1 2; do for k in 16 10; do for lib in libvmp_hip_r5.so libvmp_hip.so; do
  VMP_LIB_PATH=$R/vmp-for-svae_amd/lib/$lib K=$k python tools/r6_fwd_ab.py 2>&1 | tail -1; done; done; done | tee $O/fwd_ab.txt
python bench.py --workload t2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_t2.json 2> $O/bench_t2.err; tail -c 1500 $O/bench_t2.json
