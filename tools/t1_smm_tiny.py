import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import _mix
L = V._lib
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/gmm_tiny.npz'))
dev = lambda a: torch.as_tensor(np.asarray(a)).to('cuda', torch.float32)
x, r0 = dev(g['in_x']), dev(g['in_r0'])
K = r0.shape[1]
kap = torch.full((K,), float(g['in_kappa']), device='cuda')
r_prev, u_prev = r0, torch.ones_like(r0)
for it in range(3):
    loop = _mix.VMPLoop(x, r_prev, L.VMP_SMM, kappa=kap, u_init=u_prev)
    r = loop.step()
    e = np.abs(r.double().cpu().numpy() - g['smm%d_r' % it]).max()
    ref = np.abs(g['smm%d_r' % it] - g['smm%d_r__f32' % it].astype(np.float64)).max()
    th = [np.abs(t.double().cpu().numpy() - g['smm%d_%s' % (it, n)]).max() / np.abs(g['smm%d_%s' % (it, n)]).max() for t, n in zip(loop.theta(), ('alpha', 'beta', 'm', 'C', 'v'))]
    print('it', it, 'r err %.2e (reference fp32 own err %.2e)' % (e, ref), 'theta rel', ['%.1e' % t for t in th])
    r_prev, u_prev = dev(g['smm%d_r' % it]), dev(g['smm%d_u' % it])
