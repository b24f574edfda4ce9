"""Evaluation metrics - mirror of the hot-path-adjacent part of reference losses.py:
weighted_mse (:9-38), diagonal_gaussian_logprob (:83-145), purity (:313-349).  The (N,K,S,D)-sized reductions run
in csrc/vmp_loglike.hip (vmp_eval_cell_metrics); the cluster/label contingency table of `purity` re-uses the
mixture moment kernel (sum_n r_nk * labels_nc is exactly its first-moment block)."""
import math

import torch

from . import _lib as L
from .models import _mix


def _cell_metrics(y, mean, var, logw, mask, want_mse, want_lse):
    y = L.dev_f32(y, 'y_true')
    mean = L.dev_f32(mean, 'y_pred / mean')
    N, K, S, D = mean.shape
    if tuple(y.shape) != (N, D):
        raise AssertionError('y_true must have shape (N,D)')
    var = None if var is None else L.dev_f32(var, 'var', (N, K, S, D))
    per_s = 0
    if logw is not None:
        if tuple(logw.shape) == (N, K, S):
            per_s = 1
        elif tuple(logw.shape) != (N, K):
            raise AssertionError('log_weights must have shape (N,K) or (N,K,S)')
        logw = L.dev_f32(logw, 'log_weights')
    m8 = None
    if mask is not None:
        if tuple(mask.shape) != (N, D):
            raise AssertionError('mask must have shape (N,D)')
        m8 = mask.to(torch.uint8).contiguous()
    f32 = dict(dtype=torch.float32, device=y.device)
    mse = torch.empty(N, K, **f32) if want_mse else None
    lse = torch.empty(N, K, **f32) if want_lse else None
    L.check(L.lib().vmp_eval_cell_metrics(L.ptr(y), L.ptr(mean), L.ptr(var), L.ptr(logw), per_s, L.ptr(m8), N, K, S, D,
                                          L.ptr(mse), L.ptr(lse), L.stream()), 'vmp_eval_cell_metrics')
    return mse, lse


def weighted_mse(y_true, y_pred, r_nk_pred, name='mse'):
    """reference losses.py:9-38: mean_n sum_k r_nk mean_s sum_d (y_nd - yhat_nksd)^2."""
    mse, _ = _cell_metrics(y_true, y_pred, None, None, None, True, False)
    if tuple(r_nk_pred.shape) != tuple(mse.shape):
        raise AssertionError('r_nk_pred must have shape (N,K)')
    return (mse * r_nk_pred).sum(1).mean()


def diagonal_gaussian_logprob(y_true, mean, var, log_weights, mask=None, name='gauss_logprob'):
    """reference losses.py:83-145: mean_n log sum_k exp(log_weights) 1/S sum_s N(y_n | mean_nks, diag var_nks)."""
    _, lse = _cell_metrics(y_true, mean, var, log_weights, mask, False, True)
    return torch.logsumexp(lse, dim=1).mean()


def purity(r_nk, labels, eps=1e-10, name='purity'):
    """reference losses.py:313-349.  labels: one-hot (N,C).  Returns (entropy, purity)."""
    N, K = r_nk.shape
    C = labels.shape[1]
    r = L.dev_f32(r_nk, 'r_nk')
    cols = []
    for c0 in range(0, C, L.MAX_D):                                    # the moment kernel takes up to 8 columns
        lab = L.dev_f32(labels[:, c0:c0 + L.MAX_D].float().contiguous(), 'labels')
        st = _mix.raw_stats(lab, r, pivot=torch.zeros(lab.shape[1], dtype=torch.float32, device=r.device))
        cols.append(st[:, 2:2 + lab.shape[1]])
        N_k = st[:, 0]
    N_kc = torch.cat(cols, dim=1)
    p_kc = N_kc / (N_k + eps).unsqueeze(1)
    cluster_entropy = -(p_kc * torch.log(p_kc + eps)).sum(1)
    entropy = (N_k / N * cluster_entropy).sum()
    pur = (N_k / N * p_kc.max(dim=1).values).sum()
    return entropy.float(), pur.float()
