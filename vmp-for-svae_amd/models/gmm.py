"""Variational mixture of Gaussians (Bishop, PRML 10.2) - mirror of reference models/gmm.py:25-269.

N-sized work (statistics over the data, responsibilities) runs in the HIP kernels of csrc/vmp_mix.hip;
K-sized updates are available both as the reference's individual ``update_*`` functions (torch) and fused
on the device (vmp_mix_finalize, used by m_step / inference).
"""
import torch

from .. import _klinalg, _lib as L
from . import _mix


# ---- K-sized posterior updates (reference gmm.py:25-81), torch --------------------------------------
def update_Nk(r_nk):
    """reference gmm.py:25-27: N_k = sum_n r_nk (N-sized reduction -> the HIP stats kernel, D=1 dummy data)."""
    ones = torch.ones(r_nk.shape[0], 1, dtype=torch.float32, device=r_nk.device)
    return _mix.raw_stats(ones, r_nk)[:, 0].to(r_nk.dtype)


def _centred_moments(x, w, W_k, a_k, eps):
    """sum_n w_nk x_n / (W_k + eps) and sum_n w_nk (x_n - a_k)(x_n - a_k)^T / (W_k + eps) from the raw moments of the HIP stats
    kernel (fp64: sxx - sx a^T - a sx^T + (sum w) a a^T); returns the un-normalised sums too."""
    st = _mix.raw_stats(x, w)                                   # (K, 2+D+D*D) fp64: [sum w | . | sum w x | sum w x x^T]
    D = x.shape[1]
    sw, sx, sxx = st[:, 0], st[:, 2:2 + D], st[:, 2 + D:].reshape(-1, D, D)
    return sw, sx, sxx


def update_xk(x, r_nk, N_k):
    """reference gmm.py:30-36 (Bishop 10.52): sum_n r_nk x_n / N_k, un-normalised where N_k == 0 (NaN fallback)."""
    _, sx, _ = _centred_moments(x, r_nk, N_k, None, 0.0)
    normed = sx / N_k.double()[:, None]
    return torch.where(torch.isnan(normed), sx, normed).to(x.dtype)


def update_Sk(x, r_nk, N_k, x_k):
    """reference gmm.py:39-46 (Bishop 10.53): sum_n r_nk (x_n - x_k)(x_n - x_k)^T / N_k with the same NaN fallback."""
    sw, sx, sxx = _centred_moments(x, r_nk, N_k, x_k, 0.0)
    a = x_k.double()
    S = sxx - sx[:, :, None] * a[:, None, :] - a[:, :, None] * sx[:, None, :] + sw[:, None, None] * a[:, :, None] * a[:, None, :]
    normed = S / N_k.double()[:, None, None]
    return torch.where(torch.isnan(normed), S, normed).to(x.dtype)


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
    sign, lad = _klinalg.slogdet(P)
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


def _mahalanobis(x, beta_k, m_k, P_k, v_k, mask=None):
    x = L.dev_f32(x, 'x')
    N, D = x.shape
    K = m_k.shape[0]
    m, P, v, b = (L.dev_f32(m_k.detach().float(), 'm_k', (K, D)), L.dev_f32(P_k.detach().float(), 'P_k', (K, D, D)),
                  L.dev_f32(v_k.detach().float(), 'v_k', (K,)), L.dev_f32(beta_k.detach().float(), 'beta_k', (K,)))
    m8 = None if mask is None else mask.to(torch.uint8).contiguous()
    out = torch.empty(N, K, dtype=torch.float32, device=x.device)
    L.check(L.lib().vmp_mix_mahalanobis(L.ptr(x), L.ptr(m), L.ptr(P), L.ptr(v), L.ptr(b), L.ptr(m8), N, D, K, L.ptr(out),
                                        L.stream()), 'vmp_mix_mahalanobis')
    return out


def compute_expct_mahalanobis_dist(x, beta_k, m_k, P_k, v_k):
    """reference gmm.py:84-94 (Bishop 10.64), stand-alone: (N,K).  Inside e_step / inference the distance never leaves
    the fused pass kernel."""
    return _mahalanobis(x, beta_k, m_k, P_k, v_k)


def compute_dev_missing_data(x, beta_k, m_k, P_k, v_k, missing_data_mask):
    """reference gmm.py:97-114: as above with the missing entries of (x - m_k) zeroed."""
    return _mahalanobis(x, beta_k, m_k, P_k, v_k, missing_data_mask)


def compute_rnk(expct_log_pi, expct_log_det_cov, expct_dev):
    """reference gmm.py:141-151 (Bishop 10.49): max-shifted softmax over k of E log pi + 1/2 E log det - 1/2 E dev."""
    return torch.softmax(expct_log_pi + 0.5 * expct_log_det_cov - 0.5 * expct_dev, dim=1)


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
