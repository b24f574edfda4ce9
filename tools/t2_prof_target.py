"""Profiling target: T2 (SVAE E-step fwd + bwd + sub-sampling + M-step moments) at BASELINE config 3 size."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import svae, _svae_ops, _mix
N = int(os.environ.get('N', 1000000)); Ld = int(os.environ.get('L', 8)); K = int(os.environ.get('K', 16)); S = int(os.environ.get('S', 10))
dev = 'cuda'
# initial parameters on the CPU (torch's rocsolver-based linalg misbehaves under rocprofv3 --pmc), then moved
prior, theta = svae.init_mm(K, Ld, seed=0, param_device='cpu')
phi = [p.detach().to(dev).contiguous() for p in svae.init_recognition_params(theta, K, seed=0, param_device='cpu')]
theta = [t.to(dev).contiguous() for t in theta]
with torch.no_grad():
    hk, P, bias = _svae_ops.PhiPrepFn.apply(*[p.detach() for p in phi])        # K-sized inputs from the prep kernels
if os.environ.get('SMM', '0') == '1':                       # Student-t theta (svae.py:265-322): [alpha, mu_k, L_k, dof]
    mu_t, L_t = svae.make_loc_scale_variables([t.cpu() for t in prior], 'cpu')
    theta = [theta[0], (mu_t.detach() + torch.randn(K, Ld)).to(dev), L_t.detach().to(dev), torch.full((K,), 5.0, device=dev)]
mk, Wk, kap, nu = svae._theta_pack(theta)
mk, Wk = mk.detach().contiguous(), Wk.detach().contiguous()
g = torch.Generator(device=dev).manual_seed(0)
eta1 = torch.randn(N, Ld, device=dev, generator=g).requires_grad_(True)
eta2d = (-0.5 * torch.log1p(torch.exp(torch.randn(N, Ld, device=dev, generator=g)))).requires_grad_(True)
noise = torch.randn(N, K, Ld, S, device=dev, generator=g)
Gx = torch.randn(N, K, S, Ld, device=dev, generator=g) * 0.01
Glz = torch.randn(N, K, device=dev, generator=g) * 0.1
for it in range(int(os.environ.get('REPS', 3))):
    x, lz, Tp = _svae_ops.SvaeEStepFn.apply(eta1, eta2d, hk, P, bias, noise, mk, Wk, kap, nu)
    r = torch.exp(lz.detach())
    torch.autograd.backward([x, lz, Tp], [Gx, Glz, r])
    xs = svae.subsample_x(x.detach(), lz.detach(), seed=it, nb_out=1)[:, 0, :].contiguous()
    st = _mix.raw_stats(xs, r)
    del x, lz, Tp
torch.cuda.synchronize()
print('ok', torch.cuda.max_memory_allocated() / 2**30, 'GiB')
