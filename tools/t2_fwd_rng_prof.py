"""Profiling target: the T2 forward with in-kernel Philox noise and the round-6 epilogue (the trainer's default path) at C3 shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmp_for_svae_amd.models import svae, _svae_ops
N = int(os.environ.get('N', 250000)); Ld, K, S = 8, int(os.environ.get('K', 16)), 10
prior, theta = svae.init_mm(K, Ld, seed=0, param_device='cpu')
phi = [p.detach().cuda().contiguous() for p in svae.init_recognition_params(theta, K, seed=0, param_device='cpu')]
theta = [t.cuda().contiguous() for t in theta]
with torch.no_grad():
    hk, P, bias = _svae_ops.PhiPrepFn.apply(*phi)
mk, Wk, kap, nu = svae._theta_pack(theta)
g = torch.Generator(device='cuda').manual_seed(0)
eta1 = torch.randn(N, Ld, device='cuda', generator=g)
eta2d = -0.5 * torch.log1p(torch.exp(torch.randn(N, Ld, device='cuda', generator=g)))
for it in range(int(os.environ.get('REPS', 3))):
    with torch.no_grad():
        x, lz, Tp = _svae_ops.SvaeEStepFn.apply(eta1, eta2d, hk, P, bias, _svae_ops.PhiloxNoise(it, S, epilogue=True), mk, Wk, kap, None)
    del x, lz, Tp
torch.cuda.synchronize()
