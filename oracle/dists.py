"""Oracle: exponential-family algebra (reference distributions/*.py, helpers/tf_utils.py:25-49).

TEST INFRASTRUCTURE (see oracle/__init__.py).  torch-CPU, dtype follows the inputs.
"""
import math

import torch

inv = torch.linalg.inv          # tf.matrix_inverse  (LU)
solve = torch.linalg.solve      # tf.matrix_solve    (partial-pivot LU)
chol = torch.linalg.cholesky    # tf.cholesky


def logdet(A):
    """helpers/tf_utils.py:25-49: 2 * sum(log(diag(chol(A))))."""
    return 2.0 * torch.log(torch.diagonal(chol(A), dim1=-2, dim2=-1)).sum(-1)


# ---------------------------------------------------------------- Gaussian (distributions/gaussian.py)
def gauss_standard_to_natural(mu, sigma):
    """gaussian.py:11-19."""
    eta2 = -0.5 * inv(sigma)
    eta1 = (-2.0 * eta2 @ mu.unsqueeze(-1)).reshape(mu.shape)
    return eta1, eta2


def gauss_natural_to_standard(eta1, eta2):
    """gaussian.py:22-27."""
    sigma = inv(-2.0 * eta2)
    mu = (sigma @ eta1.unsqueeze(2)).reshape(eta1.shape)
    return mu, sigma


def gauss_log_probability_nat(x, eta1, eta2, weights=None):
    """gaussian.py:30-71: log N(x_n | eta1_nk, eta2_nk) (+ log w_k), normalised over k by log-sum-exp."""
    N, D = x.shape
    if eta1.dim() != 3:
        raise AssertionError("eta1 must be of shape (N,K,D). Its shape is %s." % str(tuple(eta1.shape)))
    lp = torch.einsum('nd,nkd->nk', x, eta1)
    lp = lp + torch.einsum('nkd,nd->nk', torch.einsum('nd,nkde->nke', x, eta2), x)
    lp = lp - D / 2. * math.log(2. * math.pi)
    e1 = eta1.unsqueeze(3)
    lp = lp + 0.25 * torch.einsum('nkdi,nkdi->nk', solve(eta2, e1), e1)
    lp = lp + 0.5 * logdet(-2. * eta2 + 1e-20 * torch.eye(D, dtype=x.dtype))
    if weights is not None:
        lp = lp + torch.log(weights).unsqueeze(0)
    mx = lp.max(dim=1, keepdim=True).values
    norm = mx + torch.log(torch.exp(lp - mx).sum(dim=1, keepdim=True))
    return lp - norm


def gauss_log_probability_nat_per_samp(xs, eta1, eta2):
    """gaussian.py:74-105: log N(x_nks | eta1_nk, eta2_nk), shape (N,K,S); not normalised over k."""
    N, K, S, D = xs.shape
    assert tuple(eta1.shape) == (N, K, D)
    assert tuple(eta2.shape) == (N, K, D, D)
    ln = torch.einsum('nksd,nksd->nks', torch.einsum('nkij,nksj->nksi', eta2, xs), xs)
    ln = ln + torch.einsum('nki,nksi->nks', eta1, xs)
    ln = ln + 0.25 * torch.einsum('nkdi,nkd->nki', solve(eta2, eta1.unsqueeze(-1)), eta1)
    ln = ln - D / 2. * math.log(2 * math.pi)
    ln = ln + 0.5 * logdet(-2.0 * eta2 + 1e-20 * torch.eye(D, dtype=xs.dtype)).unsqueeze(2)
    return ln


# ---------------------------------------------------------------- NIW (distributions/niw.py)
def _outer(a, b):
    """niw.py:46-49."""
    return a.unsqueeze(-1) * b.unsqueeze(-2)


def niw_expected_values(beta, m, C, v):
    """niw.py:8-17: E[mu]=m, E[Sigma]=inv(v * sym(inv(C)))."""
    Ci = inv(C)
    Ci = (Ci + Ci.transpose(-1, -2)) / 2.
    return m, inv(Ci * v.unsqueeze(1).unsqueeze(2))


def niw_standard_to_natural(beta, m, C, v):
    """niw.py:20-30."""
    K, D = m.shape
    assert tuple(beta.shape) == (K,)
    b = beta.unsqueeze(-1) * m
    return C + _outer(b, m), b, beta, v + D + 2


def niw_natural_to_standard(A, b, beta, v_hat):
    """niw.py:33-43."""
    m = b / beta.unsqueeze(-1)
    K, D = m.shape
    assert tuple(beta.shape) == (K,)
    return beta, m, A - _outer(b, m), v_hat - D - 2


# ---------------------------------------------------------------- Dirichlet (distributions/dirichlet.py)
def dir_expected_log_pi(alpha):
    """dirichlet.py:8-12."""
    return torch.digamma(alpha) - torch.digamma(alpha.sum(-1, keepdim=True))


def dir_standard_to_natural(alpha):
    """dirichlet.py:15-17."""
    return alpha - 1


def dir_natural_to_standard(alpha_nat):
    """dirichlet.py:20-22."""
    return alpha_nat + 1


# ---------------------------------------------------------------- Student-t (distributions/student_t.py)
def student_t_log_probability_per_samp(y, mu, sigma, v):
    """student_t.py:7-39,59-61.  The reference tiles sigma to (N,K,S,D,D) and LU-solves / Cholesky-
    factorises every copy; the tile is kept implicit here (broadcast) - same arithmetic per element."""
    N, K, S, D = y.shape
    assert tuple(mu.shape) == (K, D) and tuple(sigma.shape) == (K, D, D) and tuple(v.shape) == (K,)
    err = (y - mu.view(1, K, 1, D)).unsqueeze(-1)                       # N,K,S,D,1
    sol = solve(sigma.view(1, K, 1, D, D).expand(N, K, S, D, D), err)
    maha = torch.einsum('nksdi,nksdi->nks', err, sol)
    vv = v.view(1, K, 1)
    lp = torch.lgamma(0.5 * (vv + D)) - torch.lgamma(0.5 * vv)
    lp = lp - 0.5 * D * torch.log(math.pi * vv)
    lp = lp - 0.5 * logdet(sigma).view(1, K, 1)
    lp = lp - 0.5 * (vv + D) * torch.log1p(maha / vv)
    return lp
