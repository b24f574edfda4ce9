"""Perf exploration of the T1 kernels (not part of the product): per-kernel event timings."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import _mix
L = V._lib


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in evs)
    return t[len(t) // 2] * 1e3


def main():
    N = int(os.environ.get('N', 1000000)); D = int(os.environ.get('D', 8)); K = int(os.environ.get('K', 16))
    g = torch.Generator(device='cuda').manual_seed(0)
    c = torch.randn(K, D, device='cuda', generator=g) * 5
    x = c[torch.randint(0, K, (N,), device='cuda', generator=g)] + torch.randn(N, D, device='cuda', generator=g)
    r0 = torch.softmax(3 * torch.randn(N, K, device='cuda', generator=g), 1)
    for flav, name in ((L.VMP_GMM, 'gmm'), (L.VMP_SMM, 'smm')):
        kap = torch.full((K,), 5.0, device='cuda') if flav == L.VMP_SMM else None
        loop = _mix.VMPLoop(x, r0, flav, kappa=kap)
        loop.step()
        u = loop.u
        t_fin = timeit(loop.finalize)
        t_fused = timeit(loop.estep)
        t_eonly = timeit(lambda: _mix.estep(x, loop.post['pack'], flav, r_out=loop.r, u_out=loop.u))
        t_stats = timeit(lambda: L.check(L.lib().vmp_mix_stats_ws(L.ptr(x), L.ptr(loop.r), L.ptr(u), L.ptr(loop.pivot), N, D, K, L.ptr(loop.ws), loop.nb, L.stream()), 's'))
        t_copy = timeit(lambda: loop.r.copy_(r0))
        print('%s N=%d D=%d K=%d  finalize %.1f us | fused pass %.1f us | E-only %.1f us | stats-only %.1f us | r copy (128MB moved) %.1f us'
              % (name, N, D, K, t_fin, t_fused, t_eonly, t_stats, t_copy))


if __name__ == '__main__':
    main()
