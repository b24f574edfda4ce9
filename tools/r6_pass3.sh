#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_mix_gpu.py tests/test_svae_gpu.py tests/test_multirank_gpu.py tests/test_philox.py tests/test_prep_gpu.py tests/test_step_glue_gpu.py -m gpu -q -x 2>&1 | tail -15
timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "smm-c5-accurate or k10-300k" 2>&1 | tail -5
python tools/r6_oracle_threads.py 2>&1 | tail -12 | tee $O/oracle_threads.txt
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3; tail -3 $O/bench_default.err; head -c 600 $O/bench_default.json
python bench.py --workload smm --no-extra > $O/bench_smm.json 2>> $O/bench.err; python - <<PY
import json
j=json.loads(open('$O/bench_smm.json').read().strip().splitlines()[-1]); print('smm', j['ms_per_step'], j['extra']['accurate_mode'])
j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); e=j['extra']
print('t1', j['ms_per_step'], 'roof', j['roofline']['frac'])
print('t2', {k: e['t2_svae_vmp'][k] for k in ('ms_per_step','fwd_kernel_ms','bwd_kernel_ms','tail_ms','frac_hbm_whole_step','moved_over_algorithmic')})
print('t3', e['t3_svae_train']['ms_per_step'], e['t3_svae_train']['roofline']['kernel_ms'], e['t3_svae_train']['roofline']['frac'])
print('mb', e['t3_minibatch64'])
print('shard', json.dumps(e['shard_steps'])[:1500])
print('cpu', j['cpu_baseline']['value'], e['t2_svae_vmp']['cpu_baseline']['value'], e['t3_svae_train']['cpu_baseline']['value'])
PY
