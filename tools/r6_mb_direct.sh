#!/bin/bash
# round 6: the direct minibatch step - parity test, the minibatch bench line, the kernel sequence of one graphed replay
cd /root/repo
mkdir -p gpurun_out/r06d
timeout 900 python -m pytest tests/test_svae_gpu.py -x -q -k "direct_minibatch or graphed_step" 2>&1 | tail -15 > gpurun_out/r06d/pytest.txt
cat gpurun_out/r06d/pytest.txt
timeout 600 python - <<'PY' 2>&1 | tail -5
import json, torch, bench
r = bench.bench_minibatch(64, 10, 8, 6, 10, 50, torch.device('cuda:0'), steps=400, cpu=False)
print(json.dumps(r))
open('gpurun_out/r06d/mb64.json', 'w').write(json.dumps(r))
PY
