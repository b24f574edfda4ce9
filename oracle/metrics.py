"""Oracle: evaluation metrics (reference losses.py:9-38, 83-145, 148-310, 313-349).  TEST INFRASTRUCTURE (oracle/__init__.py)."""
import math

import torch


def weighted_mse(y_true, y_pred, r_nk):
    """losses.py:9-38: mean_n sum_k r_nk mean_s sum_d (y_nd - yhat_nksd)^2."""
    N, K, S, D = y_pred.shape
    assert tuple(y_true.shape) == (N, D) and tuple(r_nk.shape) == (N, K)
    mse = ((y_true.unsqueeze(1).unsqueeze(2) - y_pred) ** 2).sum(3).mean(2)
    return (mse * r_nk).sum(1).mean()


def diagonal_gaussian_logprob(y_true, mean, var, log_weights, mask=None):
    """losses.py:83-145: mean_n log sum_k [ w_nk(s) * 1/S sum_s N(y_n | mean_nks, var_nks) ] (two max-shifted LSEs)."""
    N, K, S, D = mean.shape
    assert var.shape == mean.shape and tuple(y_true.shape) == (N, D)
    assert tuple(log_weights.shape) in ((N, K), (N, K, S))
    y = y_true.unsqueeze(1).unsqueeze(2)
    lp = (y - mean) ** 2 / var
    lp = lp + torch.log(var)
    lp = lp + math.log(2 * math.pi)
    lp = lp * -0.5
    if mask is not None:
        lp = mask.to(lp.dtype).unsqueeze(1).unsqueeze(2) * lp
    lw = log_weights.unsqueeze(2) if log_weights.dim() == 2 else log_weights
    lp = lp.sum(3) + lw
    mx = lp.max(dim=2, keepdim=True).values
    lpz = torch.log(torch.exp(lp - mx).sum(2)) + mx.reshape(N, K)
    lpz = lpz - math.log(S)
    mz = lpz.max(dim=1, keepdim=True).values
    p_y = torch.log(torch.exp(lpz - mz).sum(1)) + mz.squeeze(1)
    return p_y.mean()


def purity(r_nk, labels, eps=1e-10):
    """losses.py:313-349.  Returns (entropy, purity)."""
    N, K = r_nk.shape
    N_kc = (r_nk.unsqueeze(2) * labels.unsqueeze(1)).sum(0)
    N_k = r_nk.sum(0)
    p_kc = N_kc / (N_k + eps).unsqueeze(1)
    cluster_entropy = -(p_kc * torch.log(p_kc + eps)).sum(1)
    entropy = (N_k / N * cluster_entropy).sum()
    pur = (N_k / N * p_kc.max(dim=1).values).sum()
    return entropy, pur


def imputation_mse(y_true, y_pred, r_nk, mask):
    """losses.py:148-170."""
    N, K, S, D = y_pred.shape
    m = mask.to(y_pred.dtype)
    yt = y_true * m
    yp = m.unsqueeze(1).unsqueeze(2) * y_pred
    se = ((yt.unsqueeze(1).unsqueeze(2) - yp) ** 2).mean(2)              # (N,K,D)
    return (se * r_nk.unsqueeze(2)).sum() / N


def perturb_data(y, mask, noise):
    """losses.py:288-310 with the tf.random_normal draw injected."""
    m = mask.to(y.dtype)
    return (1.0 - m) * y + m * noise


def imputation_losses(y_true, mask, imputation_method, noise, nb_samples_rec):
    """losses.py:173-246 ('standard' decoder): literal restatement - masked predictions, per-perturbation MSE, then ONE
    diagonal_gaussian_logprob over the imputations concatenated along the sample axis."""
    P = noise.shape[0]
    mse = 0.0
    means, vars_, lws = [], [], []
    for s in range(P):
        y_pert = perturb_data(y_true, mask, noise[s])
        mean, var, log_r = imputation_method(y_pert)
        pred = mask.to(mean.dtype).unsqueeze(1).unsqueeze(2) * mean
        mse = mse + imputation_mse(y_true, pred, torch.exp(log_r), mask)
        means.append(mean)
        vars_.append(var)
        lws.append(log_r.unsqueeze(2).repeat(1, 1, nb_samples_rec))
    ll = diagonal_gaussian_logprob(y_true, torch.cat(means, 2), torch.cat(vars_, 2), torch.cat(lws, 2), mask=mask)
    return mse / P, ll


def bernoulli_logprob(y_true_bin, logits, log_weights=None, mask=None):
    """losses.py:41-80 (note: subtracts S, not log S, :76-78)."""
    S = logits.shape[-2]
    yb = y_true_bin.unsqueeze(1)
    if log_weights is not None:
        if log_weights.dim() == 2:
            log_weights = log_weights.unsqueeze(2)
        yb = yb.unsqueeze(1)
    px = -torch.log(1. + torch.exp(-logits * yb))
    if mask is not None:
        m = mask.to(px.dtype).unsqueeze(1)
        if log_weights is not None:
            m = m.unsqueeze(1)
        px = px * m
    lp = px.sum(-1)
    if log_weights is not None:
        lp = torch.logsumexp(lp + log_weights, dim=1)
    return (torch.logsumexp(lp, dim=-1) - float(S)).mean()
