"""Profiling target: T2 (SVAE E-step fwd + bwd + M-step stats) at BASELINE config 3 size."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import svae, _svae_ops, _mix
N = int(os.environ.get('N', 1000000)); Ld = int(os.environ.get('L', 8)); K = int(os.environ.get('K', 16)); S = int(os.environ.get('S', 10))
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
eta1 = torch.randn(N, Ld, device=dev, generator=g)
eta2d = -0.5 * torch.nn.functional.softplus(torch.randn(N, Ld, device=dev, generator=g))
prior, theta = svae.init_mm(K, Ld, seed=0, param_device=dev)
phi = [p.detach().requires_grad_(True) for p in svae.init_recognition_params(theta, K, seed=0, param_device=dev)]
noise = torch.randn(N, K, Ld, S, device=dev, generator=g)
Gx = torch.randn(N, K, S, Ld, device=dev, generator=g) * 0.01
Glz = torch.randn(N, K, device=dev, generator=g) * 0.1
e1 = eta1.requires_grad_(True); e2 = eta2d.requires_grad_(True)
for it in range(int(os.environ.get('REPS', 3))):
    x, lz, pt, _ = svae.e_step((e1, e2), phi, S, noise=noise, theta=theta)
    r = torch.exp(lz.detach())
    torch.autograd.backward([x, lz, pt.T_prime], [Gx, Glz, r])
    xs = svae.subsample_x(x.detach(), lz.detach(), seed=it, nb_out=1)[:, 0, :].contiguous()
    st = _mix.raw_stats(xs, r)
    del x, lz, pt
torch.cuda.synchronize()
print('ok', torch.cuda.max_memory_allocated() / 2**30, 'GiB')
