"""Variational mixture of Student-t (Archambeau & Verleysen 2007) - mirror of reference models/smm.py:25-245.
Same kernels as gmm.py with the scale variables u_nk: weights w = r*u, W_k = sum w (smm.py:30-50),
v_k without the +1 (smm.py:73-76), Cholesky log-det without guard (smm.py:99-110)."""
import math

import torch

from .. import _klinalg, _lib as L
from . import _mix


# ---- the individual update functions of reference smm.py:25-137 (API parity; m_step / e_step / inference are fused) ----
def update_Nk(r_nk):
    """reference smm.py:25-27 (eq. 34)."""
    from . import gmm
    return gmm.update_Nk(r_nk)


def update_Wk(ru_nk):
    """reference smm.py:30-32 (eq. 35)."""
    from . import gmm
    return gmm.update_Nk(ru_nk)


def update_xk(x, ru_nk, W_k, eps=1e-20):
    """reference smm.py:35-40 (eq. 32)."""
    from . import gmm
    _, sx, _ = gmm._centred_moments(x, ru_nk, W_k, None, eps)
    return (sx / (W_k.double()[:, None] + eps)).to(x.dtype)


def update_Sk(x, ru_nk, W_k, x_k, eps=1e-20):
    """reference smm.py:43-50 (eq. 33)."""
    from . import gmm
    sw, sx, sxx = gmm._centred_moments(x, ru_nk, W_k, x_k, eps)
    a = x_k.double()
    S = sxx - sx[:, :, None] * a[:, None, :] - a[:, :, None] * sx[:, None, :] + sw[:, None, None] * a[:, :, None] * a[:, None, :]
    return (S / (W_k.double()[:, None, None] + eps)).to(x.dtype)


def update_alphak(alpha_0, N_k):
    return alpha_0 + N_k                       # eq. 27, smm.py:53-55


def update_betak(beta_0, W_k):
    return beta_0 + W_k                        # eq. 28, smm.py:58-60


def update_mk(beta_0, m_0, W_k, x_k, beta_k):
    return (beta_0.reshape(-1, 1) * m_0 + W_k[:, None] * x_k) / beta_k[:, None]      # eq. 29, smm.py:63-71


def update_vk(v_0, N_k):
    return v_0 + N_k                           # eq. 30 (no +1 here), smm.py:73-76


def update_Ck(C_0, x_k, W_k, m_0, beta_0, beta_k, S_k):
    dx = x_k - m_0                             # eq. 31, smm.py:78-85
    w = (beta_0.reshape(-1) * W_k / beta_k)[:, None, None]
    return C_0 + W_k[:, None, None] * S_k + w * (dx[:, :, None] * dx[:, None, :])


def expct_mahalanobis_dist(x, beta_k, m_k, P_k, v_k):
    """reference smm.py:88-96."""
    from . import gmm
    return gmm._mahalanobis(x, beta_k, m_k, P_k, v_k)


def expct_log_det_prec(v_k, P_k):
    """reference smm.py:99-110: Cholesky log-det (no guard), digamma arguments without the +1."""
    P = P_k.double()
    D = P.shape[-1]
    ld = 2.0 * torch.log(torch.diagonal(_klinalg.cholesky(P), dim1=-2, dim2=-1)).sum(-1)
    i = torch.arange(D, dtype=torch.float64, device=P.device)
    sdg = torch.special.digamma(0.5 * (v_k.double()[:, None] + i[None, :])).sum(1)
    return (sdg + D * math.log(2.0) + ld).to(P_k.dtype)


def expct_log_pi(alpha_k):
    """reference smm.py:113-116."""
    return torch.special.digamma(alpha_k) - torch.special.digamma(alpha_k.sum())


def compute_rnk(expct_log_pi, expct_log_det_prec, expct_m_dist, kappa_k, D):
    """reference smm.py:119-128 (eq. 19)."""
    log_r = torch.lgamma((D + kappa_k) / 2.) - torch.lgamma(kappa_k / 2.) - (D / 2.) * torch.log(kappa_k * math.pi)
    log_r = log_r + expct_log_pi + 0.5 * expct_log_det_prec
    log_r = log_r - (0.5 * (D + kappa_k) * expct_m_dist - torch.log(kappa_k))
    return torch.exp(log_r - torch.logsumexp(log_r, dim=1, keepdim=True))


def compute_expct_unk(expct_m_dist, kappa_k, D):
    """reference smm.py:131-137 (eqs. 24-25)."""
    return 0.5 * (D + kappa_k) / (0.5 * (expct_m_dist + kappa_k))


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
