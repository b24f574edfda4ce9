"""Normal-inverse-Wishart parameter algebra - mirror of reference distributions/niw.py.

All tensors here are K-sized ((K,), (K,D), (K,D,D)); they are evaluated with plain torch on whatever
device they live on (autograd-capable).  Symmetric-positive-definite inverses go through Cholesky.
"""
import torch

from .. import _klinalg


def _spd_inverse(M):
    return _klinalg.cholesky_inverse_spd(M)


def _rank1(u, w):
    return torch.einsum('kd,ke->kde', u, w)


def _outer(a, b):
    """reference niw.py:46-49: batched outer product a[..., :, None] * b[..., None, :] (the helper its conversions call)."""
    return a.unsqueeze(-1) * b.unsqueeze(-2)


def expected_values(niw_standard_params):
    """reference niw.py:8-17.  (beta, m, C, v) -> (E[mu] = m, E[Sigma] = (v * sym(C^-1))^-1)."""
    beta, m, C, v = niw_standard_params
    Csym = 0.5 * (C + C.transpose(-1, -2))
    prec = _spd_inverse(Csym) * v[:, None, None]
    return m, _spd_inverse(prec)


def standard_to_natural(beta, m, C, v):
    """reference niw.py:20-30.  -> (A, b, beta, v_hat)."""
    K, D = m.shape
    if tuple(beta.shape) != (K,):
        raise AssertionError('beta must have shape (K,)')
    b = m * beta[:, None]
    return C + _rank1(b, m), b, beta, v + (D + 2)


def natural_to_standard(A, b, beta, v_hat):
    """reference niw.py:33-43.  -> (beta, m, C, v)."""
    K, D = b.shape
    if tuple(beta.shape) != (K,):
        raise AssertionError('beta must have shape (K,)')
    m = b / beta[:, None]
    return beta, m, A - _rank1(b, m), v_hat - (D + 2)
