"""Stage stamps of ONE steady-state tile of the fused E-step forward kernel (9th tile of block 0, wave 0), -DVMP_DEBUG_TS build.
   VMP_LIB_PATH=.../libvmp_hip_ts.so K=16 RNG=1 python tools/fwd_tile_ts.py"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import svae
L = V._lib
N = int(os.environ.get('N', 1000000)); K = int(os.environ.get('K', 16)); Ld, S = 8, 10
rng = os.environ.get('RNG', '1') == '1'
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
prior, theta = svae.init_mm(K, Ld, seed=0, param_device=dev)
phi = [p.detach() for p in svae.init_recognition_params(theta, K, seed=0, param_device=dev)]
e1 = torch.randn(N, Ld, device=dev, generator=g)
e2 = (-0.5 - torch.rand(N, Ld, device=dev, generator=g))
noise = None if rng else torch.randn(N, K, Ld, S, device=dev, generator=g)
ts = torch.zeros(128, dtype=torch.int64, device=dev)
h = ctypes.CDLL(L.LIB_PATH); h.vmp_debug_set_svae_timestamps(ctypes.c_void_p(ts.data_ptr()))
for it in range(3):
    with torch.no_grad():
        x, lz, pt, _ = svae.e_step((e1, e2), phi, S, seed=it, noise='philox' if rng else noise, theta=theta)
    torch.cuda.synchronize()
    t = ts.cpu().tolist()
    del x, lz, pt
    if it == 0:
        continue
    d = lambda a, b: t[64 + b] - t[64 + a]
    print('K=%d %s: tile total %d cycles | factorisation + softmax %d | next rows + wait for the tile\'s noise %d | sample loop %d | samples out %d'
          % (K, 'in-kernel noise' if rng else 'noise tensor', d(0, 4), d(0, 1), d(1, 2), d(2, 3), d(3, 4)))
