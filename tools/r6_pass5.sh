#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_philox.py tests/test_svae_gpu.py tests/test_step_glue_gpu.py tests/test_prep_gpu.py -m gpu -q -x 2>&1 | tail -6
bash tools/kseq.sh r06mb step_scalars tools/r5_mb_graph.py 2>&1 | tail -22 | tee $O/minibatch64_kernel_seq_b.txt
python - <<PY
import torch, sys, json
sys.path.insert(0, '$R')
import bench
torch.cuda.set_device(0)
print(json.dumps(bench.bench_minibatch(64, 10, 8, 6, 10, 50, torch.device('cuda', 0), cpu=False)))
print(json.dumps({k: v for k, v in bench.bench_t2(1000000, 8, 16, 10, 10, 3, torch.device('cuda', 0), None, 1, cpu=False, tensor_mode=False).items()}))
PY
