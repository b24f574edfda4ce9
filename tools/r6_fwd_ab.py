"""Round 6 A/B on one box: T2 forward with in-kernel noise (old entry point, every library) and with the round-6 epilogue (libraries
that export it), the step's tail both ways, K from the environment.  usage: VMP_LIB_PATH=... K=16 python tools/r6_fwd_ab.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd import _lib as L
from vmp_for_svae_amd.models import svae, _svae_ops, _mix
N = int(os.environ.get('N', 1000000)); Ld, K, S = 8, int(os.environ.get('K', 16)), 10
dev = 'cuda'
prior, theta = svae.init_mm(K, Ld, seed=0, param_device=dev)
phi = list(svae.init_recognition_params(theta, K, seed=0, param_device=dev))
g = torch.Generator(device=dev).manual_seed(0)
eta1 = torch.randn(N, Ld, device=dev, generator=g)
eta2d = -0.5 * torch.nn.functional.softplus(torch.randn(N, Ld, device=dev, generator=g))
has_epi = hasattr(L.lib(), 'vmp_svae_estep_fwd_rng_epi') and hasattr(_svae_ops, 'mom_cvi')
def ev(): return torch.cuda.Event(enable_timing=True)
res = {}
for mode in (['plain', 'epi'] if has_epi else ['plain']):
    tf, tt = [], []
    for it in range(10):
        a, b, c = ev(), ev(), ev()
        with torch.no_grad():
            nz = _svae_ops.PhiloxNoise(it, S, epilogue=True) if mode == 'epi' else _svae_ops.PhiloxNoise(it, S)
            a.record()
            x, lz, pt, _ = svae.e_step((eta1, eta2d), phi, S, noise=nz, theta=theta)
            b.record()
            if mode == 'epi':
                r, xs = pt.r_nk, pt.x_samples
                if pt.mom is not None:
                    _svae_ops.mom_cvi(pt.mom, prior, theta, 0.0, want_star=False, want_stats=False)
                else:
                    svae.cvi_update_from_stats(prior, theta, _mix.raw_stats(xs, r, pivot=False), 0.0, want_star=False)
            else:
                r = torch.exp(lz)
                xs = svae.subsample_x(x, lz, seed=it, nb_out=1, u='philox')[:, 0, :].contiguous()
                svae.cvi_update_from_stats(prior, theta, _mix.raw_stats(xs, r, pivot=False), 0.0, want_star=False)
            c.record()
        torch.cuda.synchronize()
        if it >= 3:
            tf.append(a.elapsed_time(b)); tt.append(b.elapsed_time(c))
        del x, lz, pt, r, xs
    res[mode] = (np.median(tf), np.min(tf), np.median(tt))
print(os.path.basename(os.environ.get('VMP_LIB_PATH', 'libvmp_hip.so')), 'K=%d N=%d' % (K, N),
      ' | '.join('%s: fwd %.3f ms (min %.3f) tail %.3f ms' % ((m,) + v) for m, v in res.items()))
