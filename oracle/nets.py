"""Oracle: encoder/decoder MLP and expected log-likelihoods (reference models/vae.py:17-151,175-250).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Weights are a dict keyed by the reference's variable
names below one net scope: 'layer_0/kernel', 'layer_0/bias', 'layer_1/...', 'gaussian_output/kernel',
'gaussian_output/bias', 'shortcut/W', 'shortcut/b1', 'shortcut/b2'.
"""
import math

import numpy as np
import torch

NET_VARS = ('layer_0/kernel', 'layer_0/bias', 'layer_1/kernel', 'layer_1/bias',
            'gaussian_output/kernel', 'gaussian_output/bias', 'shortcut/W', 'shortcut/b1', 'shortcut/b2')


def softplus(x):
    """tf.nn.softplus = log(1 + exp(x)), evaluated stably."""
    return torch.logaddexp(x, torch.zeros_like(x))


def rand_partial_isometry(m, n, stddev, seed=0):
    """vae.py:58-72: first (m x n) block of Q from QR of a (d x d) Gaussian, d = max(m, n)."""
    d = max(m, n)
    rs = np.random.RandomState(seed)
    return np.linalg.qr(rs.normal(loc=0, scale=stddev, size=(d, d)))[0][:m, :n]


def mlp(inp, w, out_type):
    """vae.py:75-128 (make_nnet): ravel to 2-D, hidden tanh layers, Gaussian head split in halves
    (vae.py:28-49), linear shortcut xW+b1 and a*log1p(exp(b2)) (vae.py:97-116), un-ravel.
    out_type 'natparam' -> (eta1, -0.5*softplus(raw2)); 'standard' -> (mean, softplus(raw2))."""
    shape = inp.shape
    h = inp.reshape(-1, shape[-1])
    x2d = h
    n_hidden = sum(1 for k in w if k.endswith('/kernel') and k.startswith('layer_'))
    for i in range(n_hidden):
        h = torch.tanh(h @ w['layer_%d/kernel' % i] + w['layer_%d/bias' % i])
    if out_type == 'bernoulli':                                       # vae.py:53-55,89-90,97-104,122
        logits = h @ w['bernoulli_output/kernel'] + w['bernoulli_output/bias'] + x2d @ w['shortcut/W'] + w['shortcut/b1']
        return logits.reshape(tuple(shape[:-1]) + (logits.shape[-1],))
    u = h @ w['gaussian_output/kernel'] + w['gaussian_output/bias']
    raw1, raw2 = torch.chunk(u, 2, dim=-1)
    if out_type == 'standard':
        o1, o2, a = raw1, softplus(raw2), 1.0
    elif out_type == 'natparam':
        o1, o2, a = raw1, -0.5 * softplus(raw2), -0.5
    else:
        raise Exception("Type '%s' does not exist." % out_type)
    res1 = x2d @ w['shortcut/W'] + w['shortcut/b1']
    res2 = a * torch.log1p(torch.exp(w['shortcut/b2']))            # naive softplus, vae.py:116
    oshape = tuple(shape[:-1]) + (o1.shape[-1],)
    return (o1 + res1).reshape(oshape), (o2 + res2).reshape(oshape)


def encoder(y, w):
    """vae.py:131-135 with the SVAE layerspec [(U,tanh),(U,tanh),(L,'natparam')] (experiments.py:139)."""
    return mlp(y, w, 'natparam')


def decoder(x, w):
    """vae.py:138-151 with [(U,tanh),(U,tanh),(Dy,'standard')] (experiments.py:140)."""
    return mlp(x, w, 'standard')


def expected_diagonal_gaussian_loglike(y, means, vars_, weights=None):
    """vae.py:201-250."""
    if weights is None:
        if means.dim() != 3:
            means, vars_ = means.unsqueeze(1), vars_.unsqueeze(1)
        M, S, L = means.shape
        sm = ((y.unsqueeze(1) - means) ** 2 / vars_).sum() + torch.log(vars_).sum()
    else:
        M, K, S, L = means.shape
        assert vars_.shape == means.shape and tuple(weights.shape) == (M, K)
        yy = y.unsqueeze(1).unsqueeze(1)
        sm = torch.einsum('nksd,nk->', (yy - means) ** 2 / vars_ + torch.log(vars_ + 1e-8), weights)
    return -0.5 * (sm / S) - M * L / 2. * math.log(2. * math.pi)


def decoder_bernoulli(x, w):
    """vae.py:138-151 with a 'bernoulli' head: (probas, logits)."""
    logits = mlp(x, w, 'bernoulli')
    return torch.sigmoid(logits), logits


def kl_divergence(enc_mean, enc_var):
    """vae.py:154-172."""
    return -(1 + torch.log(enc_var) - enc_mean ** 2 - enc_var).sum(1).mean() / 2.


def expected_bernoulli_loglike(y_binary, logits, r_nk=None):
    """vae.py:175-198 (the reference's naive -log(1 + exp(-logit*y)))."""
    yb = y_binary.unsqueeze(1)
    if r_nk is not None:
        yb = yb.unsqueeze(1)
    img = (-torch.log(1. + torch.exp(-logits * yb))).sum(-1).mean(-1)
    if r_nk is not None:
        img = (r_nk * img).sum(1)
    return img.sum()


def reparam_trick_sampling(mean, var, noise):
    """vae.py:282-296 with the Normal draw injected: (M,S,L)."""
    return mean.unsqueeze(1) + torch.sqrt(var).unsqueeze(1) * noise


def vae_compute_elbo(y, enc_mu, enc_var, dec_output, decoder_type):
    """vae.py:253-279."""
    M = y.shape[0]
    if decoder_type == 'bernoulli':
        rec = expected_bernoulli_loglike(y, dec_output[1])
    else:
        rec = expected_diagonal_gaussian_loglike(y, dec_output[0], dec_output[1])
    return rec / M - kl_divergence(enc_mu, enc_var)
