"""Evaluation metrics - mirror of the hot-path-adjacent part of reference losses.py:
weighted_mse (:9-38), diagonal_gaussian_logprob (:83-145), purity (:313-349), and the missing-data imputation
measurements imputation_mse (:148-170), imputation_losses (:173-246), generate_missing_data_mask (:249-285),
perturb_data (:288-310).  The (N,K,S,D)-sized reductions run
in csrc/vmp_loglike.hip (vmp_eval_cell_metrics); the cluster/label contingency table of `purity` re-uses the
mixture moment kernel (sum_n r_nk * labels_nc is exactly its first-moment block)."""
import math

import numpy as np
import torch

from . import _lib as L
from .models import _mix, _svae_ops


def _cell_metrics(y, mean, var, logw, mask, want_mse, want_lse, mask_mse=False):
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
    L.check(L.lib().vmp_eval_cell_metrics(L.ptr(y), L.ptr(mean), L.ptr(var), L.ptr(logw), per_s, L.ptr(m8), 1 if mask_mse else 0, N, K, S, D,
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


def bernoulli_logprob(y_true_bin, logits, log_weights=None, missing_data_mask=None, name='bernoulli_logprob'):
    """reference losses.py:41-80.  logits (N,S,D) or, with log_weights (N,K) / (N,K,S), (N,K,S,D); y in {-1,+1}.
    Kept as written: the sample average subtracts S, not log S (losses.py:76-78)."""
    if log_weights is None:
        N, S, D = logits.shape
        lg4 = logits.unsqueeze(1)
    else:
        N, K, S, D = logits.shape
        if tuple(log_weights.shape) not in ((N, K), (N, K, S)):
            raise AssertionError('log_weights must have shape (N,K) or (N,K,S)')
        if log_weights.dim() == 2:
            log_weights = log_weights.unsqueeze(2)
        lg4 = logits
    if tuple(y_true_bin.shape) != (N, D):
        raise AssertionError('y_true_bin must have shape (N,D)')
    rows = _svae_ops.BernoulliRowsFn.apply(y_true_bin, lg4.contiguous(), missing_data_mask)      # (N,K|1,S)
    if log_weights is not None:
        logprobs = torch.logsumexp(rows + log_weights, dim=1)                                   # (N,S)
    else:
        logprobs = rows[:, 0, :]
    return (torch.logsumexp(logprobs, dim=-1) - float(S)).mean()


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


def imputation_mse(y_true, y_pred, r_nk_pred, missing_data_mask, name='imp_mse'):
    """reference losses.py:148-170: 1/N sum_nk r_nk mean_s sum_d mask_nd (y_nd - yhat_nksd)^2 (observed entries are
    zeroed in both the truth and the prediction, i.e. they do not count)."""
    mse, _ = _cell_metrics(y_true, y_pred, None, None, missing_data_mask, True, False, mask_mse=True)
    if tuple(r_nk_pred.shape) != tuple(mse.shape):
        raise AssertionError('r_nk_pred must have shape (N,K)')
    return (mse * r_nk_pred).sum() / y_true.shape[0]


def generate_missing_data_mask(y, noise_ratio=0.3, mask_type='random', seed=0, name='make_mask'):
    """reference losses.py:249-285: a random but constant boolean (N,D) mask (same numpy RandomState draw), or the
    image-shaped 'quarter' / 'lower_half' / 'left_half' masks."""
    N, D = y.shape
    mask = np.zeros(N * D, dtype=bool)
    if mask_type == 'random':
        nb = int(N * D * noise_ratio)
        missing = np.random.RandomState(seed).choice(np.arange(N * D), size=nb, replace=False)
        mask[missing] = True
    else:
        side = np.sqrt(D)
        assert side.is_integer()
        side = int(side)
        half = side // 2
        mask = mask.reshape(N, side, side)
        if mask_type == 'quarter':
            mask[:, half:side, :half] = True
        elif mask_type == 'lower_half':
            mask[:, half:side, :side] = True
        elif mask_type == 'left_half':
            mask[:, :side, :half] = True
        else:
            raise NotImplementedError("The mask type '%s' does not exist." % mask_type)
    return torch.as_tensor(mask.reshape(N, D), device=y.device)


def perturb_data(y, missing_data_mask, seed, decoder_type='standard', name='perturb_data', noise=None):
    """reference losses.py:288-310: masked entries are replaced by N(0,1) noise (`noise` (N,D) injects the draw that
    tf.random_normal makes in the reference)."""
    m = missing_data_mask.to(y.dtype)
    if noise is None:
        g = torch.Generator(device=y.device).manual_seed(int(seed))
        if decoder_type == 'standard':
            noise = torch.randn(y.shape, generator=g, device=y.device, dtype=y.dtype)
        elif decoder_type == 'bernoulli':                 # fair coin in {-1,+1} (losses.py:301-306)
            noise = (torch.rand(y.shape, generator=g, device=y.device) < 0.5).to(y.dtype) * 2.0 - 1.0
        else:
            raise NotImplementedError
    return (1.0 - m) * y + m * noise


def imputation_losses(y_true, missing_data_mask, imputation_method, nb_samples_pert=100, nb_samples_rec=100, seed=0,
                      decoder_type='standard', name='imputation_losses', noise=None):
    """reference losses.py:173-246.  imputation_method(y_perturbed) -> (mean (N,K,S,D), var (N,K,S,D), log_r_nk (N,K)).
    Returns (expected masked MSE over the perturbations, log-likelihood of the missing entries under the mixture of all
    nb_samples_pert * S imputations).  The reference concatenates every imputation along S before one
    diagonal_gaussian_logprob; the same number is accumulated here perturbation by perturbation:
    log 1/(P S) sum_{p,s} e^{..} = logsumexp_p(lse_p) - log P with lse_p the per-cell value of one perturbation.
    `noise` (P,N,D) injects the perturbation draws; by default perturbation p uses seed + p (the reference passes the
    SAME op seed to every tf.random_normal, losses.py:207)."""
    if decoder_type not in ('standard', 'bernoulli'):
        raise NotImplementedError
    bern = decoder_type == 'bernoulli'
    # for the MSE the binary data in {-1,1} is compared as {0,1} (losses.py:193-198)
    y_cmp = torch.where(y_true == -1, torch.zeros_like(y_true), torch.ones_like(y_true)) if bern else y_true
    mse = 0.0
    lse_acc = None
    rows_all, lw_all = [], []
    for p in range(nb_samples_pert):
        y_pert = perturb_data(y_true, missing_data_mask, seed + p, decoder_type=decoder_type,
                              noise=None if noise is None else noise[p])
        with torch.no_grad():
            mean, out2, log_r_nk = imputation_method(y_pert)
        if bern:
            m_p, _ = _cell_metrics(y_cmp, mean, None, None, missing_data_mask, True, False, mask_mse=True)
            rows_all.append(_svae_ops.BernoulliRowsFn.apply(y_true, out2.contiguous(), missing_data_mask))
            lw_all.append(log_r_nk.unsqueeze(2).expand(-1, -1, out2.shape[2]))
        else:
            m_p, lse_p = _cell_metrics(y_true, mean, out2, log_r_nk, missing_data_mask, True, True, mask_mse=True)
            lse_acc = lse_p if lse_acc is None else torch.logaddexp(lse_acc, lse_p)
        mse = mse + (m_p * torch.exp(log_r_nk)).sum() / y_true.shape[0]
    expected_mse = mse / nb_samples_pert
    if bern:
        # bernoulli_logprob on the imputations concatenated along S (losses.py:231-241): only (N,K,S)-sized tensors
        rows, lw = torch.cat(rows_all, dim=2), torch.cat(lw_all, dim=2)
        logprobs = torch.logsumexp(rows + lw, dim=1)
        loglike = (torch.logsumexp(logprobs, dim=-1) - float(rows.shape[2])).mean()
    else:
        loglike = torch.logsumexp(lse_acc - math.log(nb_samples_pert), dim=1).mean()
    return expected_mse, loglike
