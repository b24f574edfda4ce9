"""Oracle: conjugate-mixture VMP updates (reference models/gmm.py:25-269, models/smm.py:25-245).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Literal two-pass formulation, (N,K,D,D) temporaries
and all, because this file doubles as the timed "reference CPU path" of bench.py.
"""
import math

import torch

from . import dists


# =============================================================================== GMM (Bishop 10.2)
def gmm_update_Nk(r):
    """gmm.py:25-27."""
    return r.sum(0)


def gmm_update_xk(x, r, N_k):
    """gmm.py:30-36 (NaN -> un-normalised when N_k == 0)."""
    xk = torch.einsum('nk,nd->kd', r, x)
    xn = xk / N_k.unsqueeze(1)
    return torch.where(torch.isnan(xn), xk, xn)


def gmm_update_Sk(x, r, N_k, x_k):
    """gmm.py:39-46: centred second moment, materialises (N,K,D,D)."""
    d = x.unsqueeze(1) - x_k.unsqueeze(0)
    S = torch.einsum('nk,nkde->kde', r, torch.einsum('nkd,nke->nkde', d, d))
    Sn = S / N_k.view(-1, 1, 1)
    return torch.where(torch.isnan(Sn), S, Sn)


def gmm_m_step(x, r, alpha_0, beta_0, m_0, C_0, v_0):
    """gmm.py:201-227 with update_* of gmm.py:49-81 inlined (Bishop 10.58, 10.60-10.63; v_k has the
    reference's extra +1, gmm.py:81)."""
    N_k = gmm_update_Nk(r)
    x_k = gmm_update_xk(x, r, N_k)
    S_k = gmm_update_Sk(x, r, N_k, x_k)
    alpha_k = alpha_0 + N_k
    beta_k = beta_0 + N_k
    b0 = beta_0.reshape(-1, 1) if beta_0.dim() == 1 else beta_0
    m_k = (b0 * m_0 + N_k.unsqueeze(1) * x_k) / beta_k.unsqueeze(1)
    C = C_0 + N_k.view(-1, 1, 1) * S_k
    q0 = x_k - m_0
    C_k = C + torch.einsum('k,kde->kde', beta_0 * N_k / beta_k, torch.einsum('kd,ke->kde', q0, q0))
    v_k = v_0 + N_k + 1
    return alpha_k, beta_k, m_k, C_k, v_k, x_k, S_k


def gmm_expct_mahalanobis(x, beta_k, m_k, P_k, v_k, missing_mask=None):
    """gmm.py:84-94; with a mask: gmm.py:97-114 (missing entries of x - m_k are zeroed)."""
    D = x.shape[1]
    d = x.unsqueeze(1) - m_k.unsqueeze(0)
    if missing_mask is not None:
        d = d * (~missing_mask).to(x.dtype).unsqueeze(1)
    m = torch.einsum('k,nk->nk', v_k, torch.einsum('nkd,nkd->nk', d, torch.einsum('kde,nke->nkd', P_k, d)))
    return m + (D / beta_k).reshape(1, -1)


def gmm_expct_log_det_prec(v_k, P_k):
    """gmm.py:117-131: matrix_determinant with the det <= 1e-20 -> 0 guard."""
    det = torch.linalg.det(P_k)
    log_det = torch.where(det > 1e-20, torch.log(det), torch.zeros_like(det))
    D = P_k.shape[-1]
    i = torch.arange(D, dtype=P_k.dtype).unsqueeze(0)
    sdg = torch.digamma(0.5 * (v_k.unsqueeze(1) + 1. + i)).sum(1)
    return sdg + D * math.log(2.) + log_det


def gmm_log_pi(alpha_k):
    """gmm.py:134-138."""
    return torch.digamma(alpha_k) - torch.digamma(alpha_k.sum())


def gmm_compute_rnk(e_log_pi, e_log_det, e_dev):
    """gmm.py:141-151."""
    log_rho = e_log_pi + 0.5 * e_log_det - 0.5 * e_dev
    rho = torch.exp(log_rho - log_rho.max(dim=1).values.reshape(-1, 1))
    return rho / rho.sum(1, keepdim=True)


def gmm_e_step(x, alpha_k, beta_k, m_k, P_k, v_k, missing_mask=None):
    """gmm.py:154-174 (and :177-198 with a mask).  Returns (r_nk, exp(E log pi))."""
    dev = gmm_expct_mahalanobis(x, beta_k, m_k, P_k, v_k, missing_mask)
    eld = gmm_expct_log_det_prec(v_k, P_k)
    elp = gmm_log_pi(alpha_k)
    return gmm_compute_rnk(elp, eld, dev), torch.exp(elp)


def init_mm_params(K, D, alpha_scale=.1, beta_scale=1e-5, v_init=10., m_scale=1., C_scale=10., m_uniform=None,
                   dtype=torch.float32):
    """svae.py:433-458, deterministic part; the tf.random_uniform(-1,1) draw is an input
    (`m_uniform` in [0,1), shape (K,D)).  Returns natural (alpha, A, b, beta, v_hat)."""
    alpha = alpha_scale * torch.ones(K, dtype=dtype)
    beta = beta_scale * torch.ones(K, dtype=dtype)
    v = torch.full((K,), float(D + v_init), dtype=dtype)
    if m_uniform is None:
        m_uniform = torch.full((K, D), 0.5, dtype=dtype)
    m = m_scale * (m_uniform.to(dtype) * 2. - 1.)
    C = C_scale * torch.eye(D, dtype=dtype).unsqueeze(0).repeat(K, 1, 1)
    A, b, beta, v_hat = dists.niw_standard_to_natural(beta, m, C, v)
    return dists.dir_standard_to_natural(alpha), A, b, beta, v_hat


def vmp_prior(K, D, dtype=torch.float32):
    """The prior used by gmm.inference / smm.inference (gmm.py:252-256) in standard form."""
    alpha, A, b, beta, v_hat = init_mm_params(K, D, alpha_scale=0.05 / K, beta_scale=0.5, m_scale=0, C_scale=D + 0.5,
                                              v_init=D + 0.5, dtype=dtype)
    beta_0, m_0, C_0, v_0 = dists.niw_natural_to_standard(A, b, beta, v_hat)
    return dists.dir_natural_to_standard(alpha), beta_0, m_0, C_0, v_0


def gmm_inference_step(x, r, K=None):
    """One `sess.run(step)` of gmm.inference (gmm.py:258-269): M-step from r, P = inv(C), E-step.
    Returns (r_new, log r_new, theta=(alpha,beta,m,C,v), (x_k, S_k, pi))."""
    K = r.shape[1] if K is None else K
    prior = vmp_prior(K, x.shape[1], x.dtype)
    alpha_k, beta_k, m_k, C_k, v_k, x_k, S_k = gmm_m_step(x, r, *prior)
    P_k = dists.inv(C_k)
    r_new, pi = gmm_e_step(x, alpha_k, beta_k, m_k, P_k, v_k)
    return r_new, torch.log(r_new), (alpha_k, beta_k, m_k, C_k, v_k), (x_k, S_k, pi)


# =============================================================================== SMM (Archambeau 2007)
def smm_m_step(x, r, u, alpha_0, beta_0, m_0, C_0, v_0, eps=1e-20):
    """smm.py:167-196 with update_* of smm.py:25-85 inlined (eps instead of NaN fallback; v_k has NO +1)."""
    ru = r * u
    N_k = r.sum(0)
    W_k = ru.sum(0)
    x_k = torch.einsum('nk,nd->kd', ru, x) / (W_k.unsqueeze(1) + eps)
    err = x.unsqueeze(1) - x_k.unsqueeze(0)
    S_k = torch.einsum('nk,nkde->kde', ru, torch.einsum('nkd,nke->nkde', err, err)) / (W_k.view(-1, 1, 1) + eps)
    alpha_k = alpha_0 + N_k
    beta_k = beta_0 + W_k
    b0 = beta_0.reshape(-1, 1) if beta_0.dim() == 1 else beta_0
    m_k = (b0 * m_0 + W_k.unsqueeze(1) * x_k) / beta_k.unsqueeze(1)
    C = C_0 + W_k.view(-1, 1, 1) * S_k
    e0 = x_k - m_0
    C_k = C + torch.einsum('k,kde->kde', beta_0 * W_k / beta_k, torch.einsum('kd,ke->kde', e0, e0))
    v_k = v_0 + N_k
    return alpha_k, beta_k, m_k, C_k, v_k, x_k, S_k


def smm_expct_log_det_prec(v_k, P_k):
    """smm.py:99-110: Cholesky logdet, no guard; digamma argument 0.5*(v_k + i)."""
    D = P_k.shape[-1]
    i = torch.arange(D, dtype=P_k.dtype).unsqueeze(0)
    return torch.digamma(0.5 * (v_k.unsqueeze(1) + i)).sum(1) + D * math.log(2.) + dists.logdet(P_k)


def smm_compute_rnk(e_log_pi, e_log_det, m_dist, kappa, D):
    """smm.py:119-128 - note the operator precedence of line 124: -(0.5(D+k) m - log k)."""
    log_r = torch.lgamma((D + kappa) / 2.) - torch.lgamma(kappa / 2.) - (D / 2.) * torch.log(kappa * math.pi)
    log_r = log_r + e_log_pi + 0.5 * e_log_det
    log_r = log_r - (0.5 * (D + kappa) * m_dist - torch.log(kappa))
    return torch.exp(log_r - torch.logsumexp(log_r, dim=1, keepdim=True))


def smm_e_step(x, alpha_k, beta_k, m_k, P_k, v_k, kappa):
    """smm.py:140-164.  Returns (r_nk, u_nk, exp(E log pi))."""
    D = x.shape[1]
    md = gmm_expct_mahalanobis(x, beta_k, m_k, P_k, v_k)           # smm.py:88-96 is identical to gmm.py:84-94
    eld = smm_expct_log_det_prec(v_k, P_k)
    elp = gmm_log_pi(alpha_k)                                      # smm.py:113-116
    r = smm_compute_rnk(elp, eld, md, kappa, D)
    u = (0.5 * (D + kappa)) / (0.5 * (md + kappa))                 # smm.py:131-137
    return r, u, torch.exp(elp)


def smm_inference_step(x, r, u, kappa):
    """One step of smm.inference (smm.py:232-245)."""
    K = r.shape[1]
    prior = vmp_prior(K, x.shape[1], x.dtype)
    alpha_k, beta_k, m_k, C_k, v_k, x_k, S_k = smm_m_step(x, r, u, *prior)
    P_k = dists.inv(C_k)
    kap = torch.full((K,), float(kappa), dtype=x.dtype) if not torch.is_tensor(kappa) else kappa
    r_new, u_new, pi = smm_e_step(x, alpha_k, beta_k, m_k, P_k, v_k, kap)
    return r_new, u_new, (alpha_k, beta_k, m_k, C_k, v_k, kap), (x_k, S_k, pi)


# =============================================================================== N-chunked evaluation (large N)
# The literal formulation above materialises (N,K,D,D) temporaries, which is fine for the golden-vector sizes but not
# for N = 1e6.  The functions below evaluate THE SAME two-pass update (gmm.py:25-46 / smm.py:25-50: N_k and x_k first,
# then the centred S_k around that x_k; E-step row by row) over row chunks with the partial sums accumulated in the
# working dtype - no algebraic shortcut (no raw-moment trick), so in fp64 they are the same truth the goldens pin.
# tests/test_oracle_golden.py checks chunked == un-chunked.
def _chunks(N, chunk):
    return [(i, min(N, i + chunk)) for i in range(0, N, chunk)]


WORKERS = 1          # test infrastructure: > 1 evaluates the row chunks of the *_chunked functions several at a time (thread pool)


def _pmap(fn, chunks):
    """[fn(a, b) for (a, b) in chunks], in chunk order - on a thread pool when WORKERS > 1 (ATen releases the GIL; the partial
    results are combined by the caller in chunk order, so nothing depends on the pool)"""
    if WORKERS > 1 and len(chunks) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(WORKERS)) as ex:
            return list(ex.map(lambda ab: fn(*ab), chunks))
    return [fn(a, b) for a, b in chunks]


def _m_step_sums_chunked(x, w, chunk):
    """(sum_n w, x_k (NaN/eps handling left to the caller: returns the raw sum too), centred scatter sum)."""
    K, D = w.shape[1], x.shape[1]
    W_k = torch.zeros(K, dtype=x.dtype)
    sx = torch.zeros(K, D, dtype=x.dtype)
    for wk_c, sx_c in _pmap(lambda a, b: (w[a:b].sum(0), torch.einsum('nk,nd->kd', w[a:b], x[a:b])), _chunks(x.shape[0], chunk)):
        W_k += wk_c
        sx += sx_c
    return W_k, sx


def _scatter_chunked(x, w, x_k, chunk):
    K, D = w.shape[1], x.shape[1]
    S = torch.zeros(K, D, D, dtype=x.dtype)
    def one(a, b):
        d = x[a:b].unsqueeze(1) - x_k.unsqueeze(0)
        return torch.einsum('nk,nkd,nke->kde', w[a:b], d, d)
    for S_c in _pmap(one, _chunks(x.shape[0], chunk)):
        S += S_c
    return S


def gmm_inference_step_chunked(x, r, chunk=1 << 15):
    """gmm_inference_step (gmm.py:258-269) over row chunks.  Same return value."""
    K, D = r.shape[1], x.shape[1]
    prior = vmp_prior(K, D, x.dtype)
    alpha_0, beta_0, m_0, C_0, v_0 = prior
    N_k, sx = _m_step_sums_chunked(x, r, chunk)
    xn = sx / N_k.unsqueeze(1)
    x_k = torch.where(torch.isnan(xn), sx, xn)                                   # gmm.py:34-36
    S = _scatter_chunked(x, r, x_k, chunk)
    Sn = S / N_k.view(-1, 1, 1)
    S_k = torch.where(torch.isnan(Sn), S, Sn)                                    # gmm.py:44-46
    alpha_k, beta_k = alpha_0 + N_k, beta_0 + N_k
    m_k = (beta_0.reshape(-1, 1) * m_0 + N_k.unsqueeze(1) * x_k) / beta_k.unsqueeze(1)
    q0 = x_k - m_0
    C_k = C_0 + N_k.view(-1, 1, 1) * S_k + torch.einsum('k,kd,ke->kde', beta_0 * N_k / beta_k, q0, q0)
    v_k = v_0 + N_k + 1
    P_k = dists.inv(C_k)
    r_new = torch.cat(_pmap(lambda a, b: gmm_e_step(x[a:b], alpha_k, beta_k, m_k, P_k, v_k)[0], _chunks(x.shape[0], chunk)))
    return r_new, torch.log(r_new), (alpha_k, beta_k, m_k, C_k, v_k), (x_k, S_k, torch.exp(gmm_log_pi(alpha_k)))


def smm_inference_step_chunked(x, r, u, kappa, chunk=1 << 15, eps=1e-20):
    """smm_inference_step (smm.py:232-245) over row chunks.  Same return value."""
    K, D = r.shape[1], x.shape[1]
    alpha_0, beta_0, m_0, C_0, v_0 = vmp_prior(K, D, x.dtype)
    ru = r * u
    N_k = torch.zeros(K, dtype=x.dtype)
    for a, b in _chunks(x.shape[0], chunk):
        N_k += r[a:b].sum(0)
    W_k, sx = _m_step_sums_chunked(x, ru, chunk)
    x_k = sx / (W_k.unsqueeze(1) + eps)
    S_k = _scatter_chunked(x, ru, x_k, chunk) / (W_k.view(-1, 1, 1) + eps)
    alpha_k, beta_k = alpha_0 + N_k, beta_0 + W_k
    m_k = (beta_0.reshape(-1, 1) * m_0 + W_k.unsqueeze(1) * x_k) / beta_k.unsqueeze(1)
    e0 = x_k - m_0
    C_k = C_0 + W_k.view(-1, 1, 1) * S_k + torch.einsum('k,kd,ke->kde', beta_0 * W_k / beta_k, e0, e0)
    v_k = v_0 + N_k
    P_k = dists.inv(C_k)
    kap = torch.full((K,), float(kappa), dtype=x.dtype) if not torch.is_tensor(kappa) else kappa
    rs, us = [], []
    for r_c, u_c, pi in _pmap(lambda a, b: smm_e_step(x[a:b], alpha_k, beta_k, m_k, P_k, v_k, kap), _chunks(x.shape[0], chunk)):
        rs.append(r_c)
        us.append(u_c)
    return torch.cat(rs), torch.cat(us), (alpha_k, beta_k, m_k, C_k, v_k, kap), (x_k, S_k, pi)
