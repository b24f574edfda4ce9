"""Profiling target: a few launches of each T1 kernel at BASELINE config 3 (GMM N=1e6 D=8 K=16)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import _mix
L = V._lib
N = int(os.environ.get('N', 1000000)); D = int(os.environ.get('D', 8)); K = int(os.environ.get('K', 16))
flav = L.VMP_SMM if os.environ.get('FLAV', 'gmm') == 'smm' else L.VMP_GMM
g = torch.Generator(device='cuda').manual_seed(0)
c = torch.randn(K, D, device='cuda', generator=g) * 5
x = c[torch.randint(0, K, (N,), device='cuda', generator=g)] + torch.randn(N, D, device='cuda', generator=g)
r0 = torch.softmax(3 * torch.randn(N, K, device='cuda', generator=g), 1)
kap = torch.full((K,), 5.0, device='cuda') if flav == L.VMP_SMM else None
loop = _mix.VMPLoop(x, r0, flav, kappa=kap)
for _ in range(int(os.environ.get('REPS', 5))):
    loop.step()
    _mix.estep(x, loop.post['pack'], flav, r_out=loop.r, u_out=loop.u)
    L.check(L.lib().vmp_mix_stats_ws(L.ptr(x), L.ptr(loop.r), L.ptr(loop.u), L.ptr(loop.pivot), N, D, K, L.ptr(loop.ws), loop.nb, L.stream()), 's')
torch.cuda.synchronize()
if os.environ.get('FIN_REPEAT'):
    for _ in range(6):
        loop.finalize()
    torch.cuda.synchronize()
