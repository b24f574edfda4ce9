"""Stage time stamps of the generic E-step backward kernel (block 0, wave 0) in a -DVMP_DEBUG_TS build:
   tools/build_variant.sh ts "-DVMP_DEBUG_TS" vmp_svae.hip ; VMP_LIB_PATH=.../libvmp_hip_ts.so python tools/svae_ts.py [N K]"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import svae
L = V._lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
Ld, S = 8, 10
g = torch.Generator(device='cuda').manual_seed(0)
e1 = torch.randn(N, Ld, device='cuda', generator=g).requires_grad_(True)
e2 = (-0.5 - torch.rand(N, Ld, device='cuda', generator=g)).requires_grad_(True)
_, theta = svae.init_mm(K, Ld, seed=0)
phi = [p.detach().requires_grad_(True) for p in svae.init_recognition_params(theta, K, seed=0)]
noise = torch.randn(N, K, Ld, S, device='cuda', generator=g)
ts = torch.zeros(64, dtype=torch.int64, device='cuda')
h = ctypes.CDLL(L.LIB_PATH); h.vmp_debug_set_svae_timestamps(ctypes.c_void_p(ts.data_ptr()))
names = ['preloads issued', 'P_k table + barrier', 'parameters arrived', 'accumulators zeroed', 'eta + Cholesky + mean',
         'upstream (N,K) + row sum', 'sample loop', 'assembly', 'row sums + stores', 'block reduction', 'partials written']
for it in range(4):
    x, lz, pt, _ = svae.e_step((e1, e2), phi, S, noise=noise, theta=theta)
    gr = torch.autograd.grad([x, lz, pt.T_prime], [e1, e2] + phi, [torch.randn_like(x), torch.randn_like(lz), torch.exp(lz.detach())])
    torch.cuda.synchronize()
    t = ts.cpu().tolist()
    if it == 0:
        continue
    fn = ['P_k table + barrier', 'parameters arrived', 'eta + Cholesky + log z', 'noise / rows arrived', 'sample loop + lz, Tp stores', 'samples copied out']
    print('FORWARD total %.2f us (+ %.2f us from kernel entry to the first stage stamp)' % ((t[48 + 6] - t[48]) / 100.0, (t[48] - t[55]) / 100.0))
    for i, n in enumerate(fn):
        print('   %-28s %7d cycles  %6.2f us' % (n, t[16 + i + 1] - t[16 + i], (t[48 + i + 1] - t[48 + i]) / 100.0))
    print('BACKWARD total %d cycles = %.2f us (wall clock, 100 MHz)' % (t[11] - t[0], (t[32 + 11] - t[32]) / 100.0))
    for i, n in enumerate(names):
        print('   %-28s %7d cycles  %6.2f us' % (n, t[i + 1] - t[i], (t[32 + i + 1] - t[32 + i]) / 100.0))
