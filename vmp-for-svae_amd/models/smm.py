"""Variational mixture of Student-t (Archambeau & Verleysen 2007) - mirror of reference models/smm.py:25-245.
Same kernels as gmm.py with the scale variables u_nk: weights w = r*u, W_k = sum w (smm.py:30-50),
v_k without the +1 (smm.py:73-76), Cholesky log-det without guard (smm.py:99-110)."""
import torch

from .. import _lib as L
from . import _mix


def m_step(x, r_nk, u_nk, alpha_0, beta_0, m_0, C_0, v_0, name='m_step'):
    """reference smm.py:167-196.  Returns (alpha_k, beta_k, m_k, C_k, v_k, x_k, S_k)."""
    stats = _mix.raw_stats(x, r_nk, u_nk)
    kap = torch.ones(r_nk.shape[1], dtype=torch.float32, device=x.device)       # not used by the M-part
    p = _mix.finalize(stats, (alpha_0, beta_0, m_0, C_0, v_0), L.VMP_SMM, kappa=kap, want_pack=False)
    return p['alpha'], p['beta'], p['m'], p['C'], p['v'], p['xbar'], p['S']


def e_step(x, alpha_k, beta_k, m_k, P_k, v_k, kappa_k, name='e_step'):
    """reference smm.py:140-164.  Returns (r_nk, u_nk, exp(E log pi))."""
    pack, pi = _mix.pack_from_params(alpha_k, beta_k, m_k, P_k, v_k, L.VMP_SMM, kappa=kappa_k)
    r, u, _, _ = _mix.estep(x, pack, L.VMP_SMM)
    return r, u, pi


def inference(x, K, kappa_init, seed, name='inference', r_init=None):
    """reference smm.py:199-245: as gmm.inference with u_nk initialised to ones and constant kappa."""
    N, D = x.shape
    if r_init is None:
        g = torch.Generator(device='cpu').manual_seed(int(seed))
        e = -torch.log(torch.rand(N, K, generator=g).clamp_min(1e-30))
        r_init = (e / e.sum(1, keepdim=True)).to(x.device)
    kappa = torch.full((K,), float(kappa_init), dtype=torch.float32, device=x.device)
    loop = _mix.VMPLoop(x, r_init, L.VMP_SMM, kappa=kappa)
    from .gmm import _Handle

    def step():
        return loop.step(want_logr=True)

    def theta():
        return loop.theta() + (kappa,)

    return (step, _Handle(lambda: loop.logr), _Handle(theta), _Handle(loop.aux))
