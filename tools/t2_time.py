"""T2 kernel timings (HIP events): fused E-step forward and backward at C3 size"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import svae, _svae_ops
N = int(os.environ.get('N', 1000000)); Ld, K, S = 8, int(os.environ.get('K', 16)), 10
dev = 'cuda'
prior, theta = svae.init_mm(K, Ld, seed=0, param_device=dev)
phi = [p.detach().requires_grad_(True) for p in svae.init_recognition_params(theta, K, seed=0, param_device=dev)]
th_params = []
if os.environ.get('SMM', '0') == '1':                       # Student-t theta (svae.py:265-322): [alpha, mu_k, L_k, dof]
    mu_t, L_t = svae.make_loc_scale_variables(prior, dev)
    with torch.no_grad():
        mu_t.add_(torch.randn(K, Ld, device=dev, generator=torch.Generator(device=dev).manual_seed(1)))
    theta = [theta[0].clone(), mu_t, L_t, torch.full((K,), 5.0, device=dev)]
    th_params = [mu_t, L_t]
g = torch.Generator(device=dev).manual_seed(0)
eta1 = torch.randn(N, Ld, device=dev, generator=g).requires_grad_(True)
eta2d = (-0.5 * torch.nn.functional.softplus(torch.randn(N, Ld, device=dev, generator=g))).requires_grad_(True)
noise = torch.randn(N, K, Ld, S, device=dev, generator=g)
Gx = torch.randn(N, K, S, Ld, device=dev, generator=g) * 0.01
Glz = torch.randn(N, K, device=dev, generator=g) * 0.1
tf, tb, tp = [], [], []
for it in range(8):
    a, b, c = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    a.record()
    x, lz, pt, _ = svae.e_step((eta1, eta2d), phi, S, noise=noise, theta=theta)
    b.record()
    r = torch.exp(lz.detach())
    b2 = torch.cuda.Event(enable_timing=True); b2.record()
    grads = torch.autograd.grad([x, lz, pt.T_prime], [eta1, eta2d] + phi + th_params, [Gx, Glz, r])
    c.record()
    torch.cuda.synchronize()
    if it >= 2:
        tf.append(a.elapsed_time(b)); tb.append(b2.elapsed_time(c))
    del x, lz, pt, grads
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    with torch.no_grad():
        x, lz, pt, _ = svae.e_step((eta1, eta2d), phi, S, seed=it, noise='philox', theta=theta)
    b.record()
    torch.cuda.synchronize()
    if it >= 2:
        tp.append(a.elapsed_time(b))
    del x, lz, pt
print('T2 forward with in-kernel Philox noise: %.3f ms' % np.median(tp))
print('T2 N=%d K=%d SMM=%s: fwd %.3f ms  bwd %.3f ms   (bwd -> %.0f GB/s = %.2f of 8 TB/s)' % (N, K, os.environ.get('SMM', '0'), np.median(tf), np.median(tb), 4.0 * N * (2.0 * K * S * Ld + 4 * Ld + 3 * K) / (np.median(tb) * 1e-3) / 1e9, 4.0 * N * (2.0 * K * S * Ld + 4 * Ld + 3 * K) / (np.median(tb) * 1e-3) / 8e12))
