"""Stage stamps (clock64) of ONE tile of the LDS-ring backward kernel - the 9th tile of block 0, wave 0, in steady state -
in a -DVMP_DEBUG_TS build:
   tools/build_variant.sh ts "-fno-slp-vectorize -DVMP_DEBUG_TS" vmp_svae.hip vmp_svae_ring.hip vmp_svae_ring_t.hip vmp_mix.hip
   VMP_LIB_PATH=.../libvmp_hip_ts.so K=16 SMM=0 python tools/ring_ts.py"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import svae
L = V._lib
N = int(os.environ.get('N', 1000000)); K = int(os.environ.get('K', 16)); Ld, S = 8, 10
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
prior, theta = svae.init_mm(K, Ld, seed=0, param_device=dev)
phi = [p.detach().requires_grad_(True) for p in svae.init_recognition_params(theta, K, seed=0, param_device=dev)]
th_params = []
if os.environ.get('SMM', '0') == '1':
    mu_t, L_t = svae.make_loc_scale_variables(prior, dev)
    theta = [theta[0].clone(), mu_t, L_t, torch.full((K,), 5.0, device=dev)]
    th_params = [mu_t, L_t]
e1 = torch.randn(N, Ld, device=dev, generator=g).requires_grad_(True)
e2 = (-0.5 - torch.rand(N, Ld, device=dev, generator=g)).requires_grad_(True)
noise = torch.randn(N, K, Ld, S, device=dev, generator=g)
Gx = torch.randn(N, K, S, Ld, device=dev, generator=g) * 0.01
Glz = torch.randn(N, K, device=dev, generator=g) * 0.1
ts = torch.zeros(128, dtype=torch.int64, device=dev)
h = ctypes.CDLL(L.LIB_PATH); h.vmp_debug_set_svae_timestamps(ctypes.c_void_p(ts.data_ptr()))
for it in range(3):
    x, lz, pt, _ = svae.e_step((e1, e2), phi, S, noise=noise, theta=theta)
    gr = torch.autograd.grad([x, lz, pt.T_prime], [e1, e2] + phi + th_params, [Gx, Glz, torch.exp(lz.detach())])
    torch.cuda.synchronize()
    t = ts.cpu().tolist()
    del x, lz, pt, gr
    if it == 0:
        continue
    d = lambda a, b: t[b] - t[a]
    print('K=%d SMM=%s tile total %d cycles' % (K, os.environ.get('SMM', '0'), d(0, 26)))
    print('   eta, P_k table, Cholesky, mean      %6d' % d(0, 1))
    print('   upstream (N,K) inputs, row sum, Gc  %6d' % d(1, 2))
    prev = 2
    for p in range(S // 2):
        b = 3 + 4 * p
        print('   pair %d: wait for the stage %6d | drain to registers %5d | re-request %5d | arithmetic %6d' % (p, t[b] - t[prev], d(b, b + 1), d(b + 1, b + 2), d(b + 2, b + 3)))
        prev = b + 3
    print('   theta-side sums (Student-t)          %6d' % (t[23] - t[prev]))
    print('   assembly (Cholesky adjoint)          %6d' % d(23, 24))
    print('   row sums + store                     %6d' % d(24, 25))
    print('   component sums -> LDS accumulators   %6d' % d(25, 26))
