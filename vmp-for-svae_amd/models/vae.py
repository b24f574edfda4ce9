"""Encoder / decoder MLPs and expected log-likelihoods - mirror of reference models/vae.py:17-151,175-250.

The dense layers are plain torch (ROCm: hipBLASLt GEMMs) as BASELINE.json's north_star prescribes; the
(N,K,S,Dy)-sized reduction of the reconstruction term runs in the HIP kernels of csrc/vmp_loglike.hip.
Variables live in a name-keyed store that mimics tf.get_variable + variable_scope reuse.
"""
import numpy as np
import torch

from . import _svae_ops

VARIABLES = {}            # 'encoder_net/layer_0/kernel' -> torch.nn.Parameter


def reset_variables():
    VARIABLES.clear()


def _get_variable(name, init_fn, trainable=True):
    v = VARIABLES.get(name)
    if v is None:
        v = torch.nn.Parameter(init_fn(), requires_grad=trainable)
        VARIABLES[name] = v
    return v


def net_variables(scope):
    """The 9 tensors of one net in the reference's variable order."""
    names = ('layer_0/kernel', 'layer_0/bias', 'layer_1/kernel', 'layer_1/bias', 'gaussian_output/kernel',
             'gaussian_output/bias', 'shortcut/W', 'shortcut/b1', 'shortcut/b2')
    return [(scope + '/' + n, VARIABLES[scope + '/' + n]) for n in names if scope + '/' + n in VARIABLES]


def rand_partial_isometry(m, n, stddev, seed=0):
    """reference vae.py:58-72 (Johnson et al. init): (m x n) block of Q from the QR of a (d x d) Gaussian."""
    d = max(m, n)
    g = np.random.RandomState(seed).normal(loc=0, scale=stddev, size=(d, d))
    return np.linalg.qr(g)[0][:m, :n]


def make_layer(inputs, units, stddev=1, activation=torch.tanh, name='layer', param_device=None, seed=0, scope=''):
    """reference vae.py:17-25 (tf.layers.dense with N(0, stddev) kernel and bias)."""
    dev = inputs.device
    full = scope + '/' + name

    def init(shape):
        g = torch.Generator(device='cpu').manual_seed(int(seed))
        return lambda: (torch.randn(shape, generator=g) * stddev).to(dev)
    w = _get_variable(full + '/kernel', init((inputs.shape[-1], units)))
    b = _get_variable(full + '/bias', init((units,)))
    out = torch.addmm(b, inputs, w)
    return activation(out) if activation is not None else out


def _softplus(x):
    return torch.nn.functional.softplus(x, beta=1.0, threshold=30.0)


def make_nnet(input, layerspecs, stddev, name, param_device=None, seed=0):
    """reference vae.py:75-128: ravel to 2-D, hidden layers, Gaussian head split in halves (vae.py:28-49),
    linear shortcut x W + b1 and a * log1p(exp(b2)) (vae.py:97-116), un-ravel."""
    shape = tuple(input.shape)
    x2 = input.reshape(-1, shape[-1])
    h = x2
    for i, (units, act) in enumerate(layerspecs[:-1]):
        h = make_layer(h, units, stddev, act, 'layer_%d' % i, seed=seed, scope=name)
    out_dim, typ = layerspecs[-1]
    if typ == 'bernoulli':
        raise NotImplementedError("bernoulli decoder: SURVEY 8f rank 4 (not on the hot path of BASELINE's configs)")
    u = make_layer(h, 2 * out_dim, stddev, None, 'gaussian_output', seed=seed, scope=name)
    raw1, raw2 = u[:, :out_dim], u[:, out_dim:]
    if typ == 'standard':
        o1, o2, a = raw1, _softplus(raw2), 1.0
    elif typ == 'natparam':
        o1, o2, a = raw1, -0.5 * _softplus(raw2), -0.5
    else:
        raise Exception("Type '%s' does not exist." % typ)
    dev = input.device
    W = _get_variable(name + '/shortcut/W', lambda: torch.as_tensor(rand_partial_isometry(shape[-1], out_dim, 1., seed),
                                                                    dtype=torch.float32).to(dev))
    b1 = _get_variable(name + '/shortcut/b1', lambda: torch.zeros(out_dim, device=dev))
    b2 = _get_variable(name + '/shortcut/b2', lambda: torch.zeros(out_dim, device=dev))
    res1 = torch.addmm(b1, x2, W)
    res2 = a * torch.log1p(torch.exp(b2))
    oshape = shape[:-1] + (out_dim,)
    return (o1 + res1).reshape(oshape), (o2 + res2).reshape(oshape)


def make_encoder(input, layerspecs=None, stddev_init=1., param_device=None, seed=0):
    """reference vae.py:131-135."""
    if layerspecs is None:
        layerspecs = [(100, torch.tanh), (100, torch.tanh), (10, 'standard')]
    return make_nnet(input, layerspecs, stddev_init, 'encoder_net', param_device, seed)


def make_decoder(input, layerspecs=None, stddev_init=1., param_device=None, seed=0):
    """reference vae.py:138-151."""
    if layerspecs is None:
        layerspecs = [(100, torch.tanh), (100, torch.tanh), (784, 'standard')]
    return make_nnet(input, layerspecs, stddev_init, 'decoder_net', param_device, seed)


def expected_diagonal_gaussian_loglike(y, means, vars, weights=None, name='diag_gauss_expct'):
    """reference vae.py:201-250.  weights (N,K) branch: HIP reduction over (s,d) + K-cheap contraction."""
    if weights is None:
        if means.dim() != 3:
            means, vars = means.unsqueeze(1), vars.unsqueeze(1)
        M, S, Ld = means.shape
        A = _svae_ops.DiagGaussLoglikeFn.apply(y, means.unsqueeze(1), (vars - 1e-8).unsqueeze(1)) \
            if False else None
        raise NotImplementedError('plain-VAE branch (weights=None): SURVEY 8f rank 4')
    M, K, S, Ld = means.shape
    if tuple(vars.shape) != tuple(means.shape) or tuple(weights.shape) != (M, K):
        raise AssertionError('shape mismatch')
    A = _svae_ops.DiagGaussLoglikeFn.apply(y, means, vars)
    sample_mean = (A * weights).sum() / S
    return -0.5 * sample_mean - M * Ld / 2. * float(np.log(2. * np.pi))
