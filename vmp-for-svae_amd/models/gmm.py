"""Variational mixture of Gaussians (Bishop, PRML 10.2) - mirror of reference models/gmm.py:25-269.

N-sized work (statistics over the data, responsibilities) runs in the HIP kernels of csrc/vmp_mix.hip;
K-sized updates are available both as the reference's individual ``update_*`` functions (torch) and fused
on the device (vmp_mix_finalize, used by m_step / inference).
"""
import torch

from .. import _lib as L
from . import _mix


# ---- K-sized posterior updates (reference gmm.py:25-81), torch --------------------------------------
def update_Nk(r_nk):
    """reference gmm.py:25-27: N_k = sum_n r_nk (N-sized reduction -> the HIP stats kernel, D=1 dummy data)."""
    ones = torch.ones(r_nk.shape[0], 1, dtype=torch.float32, device=r_nk.device)
    return _mix.raw_stats(ones, r_nk)[:, 0].to(r_nk.dtype)


def update_alphak(alpha_0, N_k):
    return alpha_0 + N_k                       # Bishop 10.58, reference gmm.py:49-51


def update_betak(beta_0, N_k):
    return beta_0 + N_k                        # Bishop 10.60, reference gmm.py:54-56


def update_mk(beta_0, m_0, N_k, x_k, beta_k):
    b0 = beta_0.reshape(-1, 1)                 # Bishop 10.61, reference gmm.py:59-68
    return (b0 * m_0 + N_k[:, None] * x_k) / beta_k[:, None]


def update_Ck(C_0, x_k, N_k, m_0, beta_0, beta_k, S_k):
    dx = x_k - m_0                             # Bishop 10.62, reference gmm.py:71-76
    w = (beta_0.reshape(-1) * N_k / beta_k)[:, None, None]
    return C_0 + N_k[:, None, None] * S_k + w * (dx[:, :, None] * dx[:, None, :])


def update_vk(v_0, N_k):
    return v_0 + N_k + 1                       # Bishop 10.63 with the reference's +1, gmm.py:79-81


def compute_expct_log_det_prec(v_k, P_k):
    """reference gmm.py:117-131, including the det <= 1e-20 -> log det := 0 guard (K-sized, torch fp64)."""
    P = P_k.double()
    D = P.shape[-1]
    sign, lad = torch.linalg.slogdet(P)
    thresh = torch.log(torch.tensor(1e-20, dtype=torch.float64, device=P.device))
    ld = torch.where((sign > 0) & (lad > thresh), lad, torch.zeros_like(lad))
    i = torch.arange(D, dtype=torch.float64, device=P.device)
    sdg = torch.special.digamma(0.5 * (v_k.double()[:, None] + 1.0 + i[None, :])).sum(1)
    return (sdg + D * torch.log(torch.tensor(2.0, dtype=torch.float64)).item() + ld).to(P_k.dtype)


def compute_log_pi(alpha_k):
    """reference gmm.py:134-138."""
    return torch.special.digamma(alpha_k) - torch.special.digamma(alpha_k.sum())


# ---- steps ---------------------------------------------------------------------------------------------
def m_step(x, r_nk, alpha_0, beta_0, m_0, C_0, v_0, name='m_step'):
    """reference gmm.py:201-227.  Returns (alpha_k, beta_k, m_k, C_k, v_k, x_k, S_k)."""
    stats = _mix.raw_stats(x, r_nk)
    p = _mix.finalize(stats, (alpha_0, beta_0, m_0, C_0, v_0), L.VMP_GMM, want_pack=False)
    return p['alpha'], p['beta'], p['m'], p['C'], p['v'], p['xbar'], p['S']


def e_step(x, alpha_k, beta_k, m_k, P_k, v_k, name='e_step'):
    """reference gmm.py:154-174.  Returns (r_nk, exp(E log pi))."""
    pack, pi = _mix.pack_from_params(alpha_k, beta_k, m_k, P_k, v_k, L.VMP_GMM)
    r, _, _, _ = _mix.estep(x, pack, L.VMP_GMM)
    return r, pi


def e_step_missing_data(x, alpha_k, beta_k, m_k, P_k, v_k, missing_data_mask, name='e_step_imp'):
    """reference gmm.py:177-198: entries flagged in the (N,D) mask are ignored in the Mahalanobis term."""
    pack, pi = _mix.pack_from_params(alpha_k, beta_k, m_k, P_k, v_k, L.VMP_GMM)
    r, _, _, _ = _mix.estep(x, pack, L.VMP_GMM, miss_mask=missing_data_mask)
    return r, pi


def compute_expct_mahalanobis_dist(x, beta_k, m_k, P_k, v_k):
    """reference gmm.py:84-94 is fused into the E-pass kernel; this stand-alone form recovers it from the
    kernel's un-normalised log-responsibilities is not offered - use e_step."""
    raise NotImplementedError('fused into vmp_mix_estep; use e_step / e_step_missing_data')


class _Handle(object):
    """Stand-in for a TF fetch: call it to get the current value."""

    def __init__(self, fn):
        self._fn = fn

    def __call__(self):
        return self._fn()


def inference(x, K, seed, name='inference', r_init=None):
    """reference gmm.py:230-269.  Returns (step, log_r_nk, theta, (x_k, S_k, pi)) where `step()` executes one
    VMP iteration (M-step, E-step, assign) and returns the new r_nk; the other three are handles that are
    CALLED to fetch the values of the last executed iteration.  `r_init` replaces the TF-RNG Dirichlet(1)
    draw of gmm.py:246-249 (default: torch Dirichlet(1) with `seed`)."""
    N, D = x.shape
    if r_init is None:
        g = torch.Generator(device='cpu').manual_seed(int(seed))
        e = -torch.log(torch.rand(N, K, generator=g).clamp_min(1e-30))
        r_init = (e / e.sum(1, keepdim=True)).to(x.device)
    loop = _mix.VMPLoop(x, r_init, L.VMP_GMM)

    def step():
        return loop.step(want_logr=True)

    return (step, _Handle(lambda: loop.logr), _Handle(loop.theta), _Handle(loop.aux))
