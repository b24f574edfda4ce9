#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests/test_mix_gpu.py -m gpu -q -x -k "accurate" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "smm-c5-accurate" 2>&1 | tail -30
timeout 600 python -m pytest tests/test_multirank_gpu.py -m gpu -q 2>&1 | tail -15
python bench.py --workload smm --no-extra > $O/bench_smm.json 2>> $O/bench.err; python - <<PY
import json
j=json.loads(open('$O/bench_smm.json').read().strip().splitlines()[-1]); print('smm', j['ms_per_step'], j['extra']['accurate_mode'])
PY
