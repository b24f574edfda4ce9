"""The N>1 path on CPU: world_size-2 gloo.  The HIP kernels cannot run here, so each rank's local moments /
gradients are produced with numpy / the oracle (test infrastructure); what is under test is the product's
exchange step - packing, ONE all-reduce, unpacking, the identical K-sized update on every rank - and that it
reproduces the single-process result on the concatenated minibatch (reference: tower split data.py:174-175,
gather + M-step on the parameter device experiments.py:247-260, gradient mean helpers/tf_utils.py:52-87)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _raw_stats(x, r):
    K, D = r.shape[1], x.shape[1]
    st = np.zeros((K, 2 + D + D * D))
    st[:, 0] = r.sum(0)
    st[:, 1] = r.sum(0)
    st[:, 2:2 + D] = r.T @ x
    st[:, 2 + D:] = np.einsum('nk,nd,ne->kde', r, x, x).reshape(K, -1)
    return torch.as_tensor(st)


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from vmp_for_svae_amd import training
    from vmp_for_svae_amd.models import svae, parallel_mix
    from oracle import svae_ref
    rng = np.random.Generator(np.random.PCG64(0))
    N, K, Ld = 64, 5, 3
    x = rng.standard_normal((N, Ld)) * 2
    r = rng.random((N, K))
    r /= r.sum(1, keepdims=True)
    shard = slice(rank * N // world, (rank + 1) * N // world)
    # ---- packed exchange: moments + gradients + scalars
    stats = _raw_stats(x[shard], r[shard])
    grads = [torch.full((3, 2), float(rank + 1)), torch.full((4,), 10.0 * (rank + 1))]
    scalars = [torch.tensor(1.0 + rank), torch.tensor(2.0), torch.tensor(3.0 * rank)]
    buf = training.pack_for_allreduce(stats, grads, scalars)
    parallel_mix.allreduce_sum_(buf)
    st_all, g_all, sc = training.unpack_after_allreduce(buf, tuple(stats.shape), [tuple(g.shape) for g in grads], 3)
    ok = bool(torch.allclose(st_all, _raw_stats(x, r), rtol=1e-12, atol=1e-12))
    ok &= bool(torch.allclose(g_all[0], torch.full((3, 2), 3.0, dtype=torch.float64)))
    ok &= bool(torch.allclose(g_all[1] / world, torch.full((4,), 15.0, dtype=torch.float64)))
    ok &= bool(torch.allclose(sc, torch.tensor([3.0, 4.0, 3.0], dtype=torch.float64)))
    # ---- every rank's theta* from the summed moments == single-process M-step on the whole minibatch
    prior64, _ = svae_ref.init_mm(K, Ld, torch.as_tensor(rng.random((K, Ld))), torch.float64)
    want = svae_ref.m_step(prior64, torch.as_tensor(x), torch.as_tensor(r))
    got = svae.m_step_from_stats([p.float() for p in prior64], st_all)
    for a, b in zip(got, want):
        ok &= bool(torch.allclose(a.double(), b, rtol=2e-6, atol=1e-5))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_world2_packed_allreduce_and_mstep():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)], res


def test_pack_unpack_roundtrip():
    from vmp_for_svae_amd import training
    st = torch.arange(12, dtype=torch.float64).reshape(3, 4)
    gs = [torch.randn(2, 3), torch.randn(5)]
    buf = training.pack_for_allreduce(st, gs, [torch.tensor(1.5), torch.tensor(-2.0)])
    s2, g2, sc = training.unpack_after_allreduce(buf, (3, 4), [(2, 3), (5,)], 2)
    assert torch.equal(s2, st) and torch.allclose(g2[0].float(), gs[0]) and torch.allclose(g2[1].float(), gs[1])
    assert sc.tolist() == [1.5, -2.0]


def test_pack_exchange_buffer_host_form():
    """host tensors (this suite has no GPU): pack_exchange_buffer falls back to the torch form and reports the gradient offsets"""
    from vmp_for_svae_amd import training
    st = torch.arange(12, dtype=torch.float64).reshape(3, 4)
    gs = [torch.randn(2, 3), torch.randn(5)]
    sc = [torch.tensor(1.5), torch.tensor(-2.0)]
    buf, goffs = training.pack_exchange_buffer(st, gs, sc)
    assert torch.equal(buf, training.pack_for_allreduce(st, gs, sc)) and goffs == [12, 18]


def test_tf_adam_matches_oracle_formulation():
    from vmp_for_svae_amd.training import TFAdam, exponential_decay
    p = torch.nn.Parameter(torch.tensor([1.0, -2.0, 3.0]))
    opt = TFAdam([p], lr=0.1)
    m = v = np.zeros(3)
    w = np.array([1.0, -2.0, 3.0])
    for t in range(1, 4):
        g = np.array([0.5, -1.0, 2.0]) * t
        opt.apply_gradients([torch.as_tensor(g, dtype=torch.float32)])
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        w = w - 0.1 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-8)
        assert np.allclose(p.detach().numpy(), w, rtol=1e-5)
    assert abs(exponential_decay(0.2, 500, 1000, 0.95) - 0.2 * 0.95 ** 0.5) < 1e-15
