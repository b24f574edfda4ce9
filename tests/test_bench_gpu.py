"""bench.py's MULTI-RANK code paths executed with two ranks (the driver's 2/4/8-GPU runs cannot be rehearsed on a 1-GPU box with
RCCL, which refuses two ranks on one device: VMP_BENCH_BACKEND=gloo lets two processes share GPU 0 - HIP kernels on the
device, the per-step exchange staged through the host, or through the IPC peer buffers with --exchange peer).  Checks the
contract arithmetic of both scaling modes and that the line carries the other mode as well."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run2(args):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   VMP_BENCH_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + args, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, e = p.communicate()
        outs.append((o.decode(), e.decode(errors='replace')))
    assert all(p.returncode == 0 for p in procs), '\n'.join(e[-2000:] for _, e in outs)
    lines = [l for l in outs[0][0].splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # rank 0: exactly one JSON line on stdout
    assert outs[1][0].strip() == ''                    # other ranks: nothing
    return json.loads(lines[0])


@pytest.mark.parametrize('scaling,exchange', [('weak', 'rccl'), ('strong', 'peer'), ('weak', 'auto')])
def test_t1_two_ranks_both_scaling_modes(scaling, exchange):
    N = 40000
    d = _run2(['--n', str(N), '--steps', '5', '--warmup', '2', '--reps', '3', '--no-extra', '--no-cpu-baseline',
               '--scaling', scaling] + (['--exchange', exchange] if exchange != 'auto' else []))      # auto is the default
    ch = d['config']['exchange_choice']
    assert ch['requested'] == exchange
    if exchange == 'auto':
        # decided by the ranks together: the peer form if it opened, agreed with the all-reduce form and was faster, else the all-reduce
        assert ch['exchange'] in ('peer', 'rccl') and ch['why']
        exchange = ch['exchange']
    assert d['n_gpus'] == 2 and d['scaling'] == scaling and d['steps'] == 5 and d['warmup'] == 2
    job = 2 * N if scaling == 'weak' else N
    assert d['config']['N_job'] == job and d['config']['N_per_gpu'] == (N if scaling == 'weak' else N // 2)
    assert abs(d['value'] - job / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    o = d['extra']['other_scaling']
    assert o['scaling'] == ('strong' if scaling == 'weak' else 'weak') and o['exchange'] == exchange
    assert o['rows_job'] == (N if scaling == 'weak' else 2 * N)
    assert abs(o['value'] - o['rows_job'] / (o['ms_per_step'] * 1e-3)) < 1e-6 * o['value']
    both = d['config']['datapoints_per_sec_by_scaling']                # both modes in the top-level config
    assert both[scaling] == d['value'] and both[o['scaling']] == o['value'] and 'scaling_note' in d['config']
    r = d['roofline']
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and r['algorithmic_bytes_per_launch'] == 4.0 * d['config']['N_per_gpu'] * 48


@pytest.mark.parametrize('workload', ['t2', 't3', 'smm'])
def test_other_workloads_run_with_two_ranks(workload):
    """C4 (t3) and C5 (smm) bench commands, and t2, under a 2-rank launch."""
    N = 8192 if workload != 'smm' else 40000
    d = _run2(['--workload', workload, '--n', str(N), '--steps', '3', '--warmup', '1', '--reps', '2', '--no-extra',
               '--no-cpu-baseline'])
    assert d['n_gpus'] == 2 and d['config']['N_job'] == 2 * N
    assert abs(d['value'] - 2 * N / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    assert d['roofline']['bound'] in ('hbm', 'mfma') and d['value'] > 0
