"""Encoder / decoder MLPs, expected log-likelihoods and the plain-VAE pieces - mirror of reference models/vae.py.

The networks the SVAE driver builds (experiments.py:139-140: two tanh layers of equal width U <= 64, Gaussian head,
in/out dimension <= 8) run in the fused fp32-MFMA kernels of csrc/vmp_decoder.hip: the decoder fused with the
reconstruction term (LazyReconstruction -> one launch for the value and every gradient), the encoder as a
stand-alone MLP (GaussMLPFn).  Any other layerspecs (e.g. the 784-wide Bernoulli decoder) use torch / hipBLASLt for
the dense layers and the streaming kernels of csrc/vmp_loglike.hip for the (N,K,S,D)-sized reductions.
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


def _layer_variables(in_dim, units, stddev, name, seed, scope, dev):
    """kernel / bias of one tf.layers.dense (reference vae.py:17-25: N(0, stddev) kernel and bias)."""
    full = scope + '/' + name

    def init(shape):
        g = torch.Generator(device='cpu').manual_seed(int(seed))
        return lambda: (torch.randn(shape, generator=g) * stddev).to(dev)
    return _get_variable(full + '/kernel', init((in_dim, units))), _get_variable(full + '/bias', init((units,)))


def make_layer(inputs, units, stddev=1, activation=torch.tanh, name='layer', param_device=None, seed=0, scope=''):
    """reference vae.py:17-25 (tf.layers.dense with N(0, stddev) kernel and bias)."""
    w, b = _layer_variables(inputs.shape[-1], units, stddev, name, seed, scope, inputs.device)
    out = torch.addmm(b, inputs, w)
    return activation(out) if activation is not None else out


def _softplus(x):
    return torch.nn.functional.softplus(x, beta=1.0, threshold=30.0)


def _shortcut_variables(in_dim, out_dim, name, seed, dev):
    """reference vae.py:97-116: W = partial isometry (vae.py:58-72), b1 = b2 = 0."""
    W = _get_variable(name + '/shortcut/W', lambda: torch.as_tensor(rand_partial_isometry(in_dim, out_dim, 1., seed),
                                                                    dtype=torch.float32).to(dev))
    b1 = _get_variable(name + '/shortcut/b1', lambda: torch.zeros(out_dim, device=dev))
    b2 = _get_variable(name + '/shortcut/b2', lambda: torch.zeros(out_dim, device=dev))
    return W, b1, b2


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
        # reference vae.py:53-55,89-90,122: logits = dense(h) + x W + b1 (no second shortcut output)
        logits = make_layer(h, out_dim, stddev, None, 'bernoulli_output', seed=seed, scope=name)
        W = _get_variable(name + '/shortcut/W', lambda: torch.as_tensor(rand_partial_isometry(shape[-1], out_dim, 1., seed),
                                                                        dtype=torch.float32).to(input.device))
        b1 = _get_variable(name + '/shortcut/b1', lambda: torch.zeros(out_dim, device=input.device))
        return (logits + torch.addmm(b1, x2, W)).reshape(shape[:-1] + (out_dim,))
    u = make_layer(h, 2 * out_dim, stddev, None, 'gaussian_output', seed=seed, scope=name)
    raw1, raw2 = u[:, :out_dim], u[:, out_dim:]
    if typ == 'standard':
        o1, o2, a = raw1, _softplus(raw2), 1.0
    elif typ == 'natparam':
        o1, o2, a = raw1, -0.5 * _softplus(raw2), -0.5
    else:
        raise Exception("Type '%s' does not exist." % typ)
    W, b1, b2 = _shortcut_variables(shape[-1], out_dim, name, seed, input.device)
    res1 = torch.addmm(b1, x2, W)
    res2 = a * torch.log1p(torch.exp(b2))
    oshape = shape[:-1] + (out_dim,)
    return (o1 + res1).reshape(oshape), (o2 + res2).reshape(oshape)


def make_encoder(input, layerspecs=None, stddev_init=1., param_device=None, seed=0):
    """reference vae.py:131-135.  The encoder the SVAE driver builds (experiments.py:139: two tanh layers of width
    U <= 64, Gaussian head, in/out <= 8) runs in the fused MFMA MLP kernels (one launch each way)."""
    if layerspecs is None:
        layerspecs = [(100, torch.tanh), (100, torch.tanh), (10, 'standard')]
    if input.is_cuda and input.dim() == 2 and _fused_mlp_eligible(input.shape[-1], layerspecs):
        ps = decoder_variables(input.shape[-1], layerspecs, stddev_init, seed, input.device, name='encoder_net')
        return _svae_ops.GaussMLPFn.apply(input, 1.0 if layerspecs[-1][1] == 'standard' else -0.5, *ps)
    return make_nnet(input, layerspecs, stddev_init, 'encoder_net', param_device, seed)


def _fused_mlp_eligible(in_dim, layerspecs):
    if len(layerspecs) != 3 or layerspecs[-1][1] not in ('standard', 'natparam'):
        return False
    (u0, a0), (u1, a1), (dy, _) = layerspecs
    return (a0 is torch.tanh and a1 is torch.tanh and u0 == u1 and _svae_ops.fused_decoder_supported(in_dim, u0, dy))


def fused_decoder_eligible(in_dim, layerspecs):
    """The fused MFMA decoder (csrc/vmp_decoder.hip) covers the decoder the SVAE driver builds
    (experiments.py:140): two tanh layers of equal width U <= 64, 'standard' Gaussian head, L, Dy <= 8."""
    if len(layerspecs) != 3 or layerspecs[-1][1] != 'standard':
        return False
    (u0, a0), (u1, a1), (dy, _) = layerspecs
    return (a0 is torch.tanh and a1 is torch.tanh and u0 == u1
            and _svae_ops.fused_decoder_supported(in_dim, u0, dy))


def decoder_variables(in_dim, layerspecs, stddev_init=1., seed=0, device='cuda', name='decoder_net'):
    """The 9 decoder variables in the reference's order, created (as make_nnet would) if they do not exist yet."""
    ps = []
    d = in_dim
    for i, (units, _) in enumerate(layerspecs[:-1]):
        ps += list(_layer_variables(d, units, stddev_init, 'layer_%d' % i, seed, name, device))
        d = units
    out_dim = layerspecs[-1][0]
    ps += list(_layer_variables(d, 2 * out_dim, stddev_init, 'gaussian_output', seed, name, device))
    ps += list(_shortcut_variables(in_dim, out_dim, name, seed, device))
    return ps


class LazyReconstruction(object):
    """Deferred decoder output for the training step: stands where the reference has y_reconstruction =
    (means, vars) of shape (N,K,S,Dy) (svae.py:511-512) but keeps only the decoder INPUT.  compute_elbo turns it
    into the reconstruction term with the fused decoder+log-likelihood kernels; iterating / indexing it
    materialises (means, vars) through the fused forward kernel (no gradient)."""

    def __init__(self, x_k_samples, params):
        self.x = x_k_samples
        self.params = list(params)
        self._out = None

    def loglike_cells(self, y):
        """A (N,K) = sum_{s,d} (y - mean)^2 / var + log(var + 1e-8), differentiable w.r.t. x and the parameters."""
        return _svae_ops.DecoderLoglikeFn.apply(y, self.x, *self.params)

    def weighted_loglike(self, y, weights):
        """sum_nk weights_nk A_nk (the einsum of vae.py:240) - value and all gradients from one kernel launch."""
        return _svae_ops.DecoderWeightedLoglikeFn.apply(y, self.x, weights, *self.params)

    def materialize(self):
        if self._out is None:
            self._out = _svae_ops.decoder_outputs(self.x, self.params)
        return self._out

    def __iter__(self):
        return iter(self.materialize())

    def __getitem__(self, i):
        return self.materialize()[i]

    def __len__(self):
        return 2


def make_decoder(input, layerspecs=None, stddev_init=1., param_device=None, seed=0, lazy=False):
    """reference vae.py:138-151.  lazy=True (training step) returns a LazyReconstruction when the fused decoder
    kernels cover the layerspecs; without gradients the fused forward kernel produces (means, vars) directly."""
    if layerspecs is None:
        layerspecs = [(100, torch.tanh), (100, torch.tanh), (784, 'standard')]
    if input.is_cuda and fused_decoder_eligible(input.shape[-1], layerspecs):
        if lazy and input.dim() == 4:
            return LazyReconstruction(input, decoder_variables(input.shape[-1], layerspecs, stddev_init, seed, input.device))
        if not torch.is_grad_enabled():
            ps = decoder_variables(input.shape[-1], layerspecs, stddev_init, seed, input.device)
            return _svae_ops.decoder_outputs(input, ps)
    output = make_nnet(input, layerspecs, stddev_init, 'decoder_net', param_device, seed)
    if layerspecs[-1][1] == 'bernoulli':                      # vae.py:147-149: (probas, logits)
        output = torch.sigmoid(output), output
    return output


def make_gaussian_layer(inputs, output_dim, stddev=1, type='standard', name='gaussian_output', param_device=None, seed=0,
                        scope=''):
    """reference vae.py:28-50: dense layer of width 2*output_dim whose halves are (mean, softplus) ('standard') or
    (eta1, -1/2 softplus) ('natparam')."""
    u = make_layer(inputs, 2 * output_dim, stddev, None, name, param_device, seed, scope)
    raw1, raw2 = u[..., :output_dim], u[..., output_dim:]
    if type == 'standard':
        return raw1, _softplus(raw2)
    if type == 'natparam':
        return raw1, -0.5 * _softplus(raw2)
    raise Exception("Type '%s' does not exist." % type)


def make_bernoulli_layer(input, output_dim, stddev=1, name='bernoulli_output', param_device=None, seed=0, scope=''):
    """reference vae.py:53-55."""
    return make_layer(input, output_dim, stddev, None, name, param_device, seed, scope)


def build_kl_divergence(enc_mean, enc_var, name='kl_divergence'):
    """reference vae.py:154-172: KL(N(mean, diag var) || N(0, I)) averaged over the minibatch ((M,L)-sized: torch)."""
    return -(1 + torch.log(enc_var) - enc_mean ** 2 - enc_var).sum(dim=1).mean() / 2.


def expected_bernoulli_loglike(y_binary, logits, r_nk=None, name='bernoulli_expct_loglike'):
    """reference vae.py:175-198: sum_n [sum_k r_nk] mean_s sum_d -log(1 + exp(-logit y)); the (.., S, D)-sized part
    runs in the HIP kernel vmp_bernoulli_rows."""
    if r_nk is None:
        N, S, D = logits.shape
        if tuple(y_binary.shape) != (N, D):
            raise AssertionError('y_binary must have shape (N,D)')
        rows = _svae_ops.BernoulliRowsFn.apply(y_binary, logits.unsqueeze(1))         # (N,1,S)
        return rows.mean(-1).sum()
    N, K, S, D = logits.shape
    if tuple(y_binary.shape) != (N, D) or tuple(r_nk.shape) != (N, K):
        raise AssertionError('shape mismatch')
    rows = _svae_ops.BernoulliRowsFn.apply(y_binary, logits)                          # (N,K,S)
    return (r_nk * rows.mean(-1)).sum()


def compute_elbo(y_true, enc_mu, enc_var, dec_output, decoder_type='standard', name='elbo'):
    """reference vae.py:253-279: the plain-VAE ELBO (Kingma & Welling, eq. 8), per datapoint."""
    M, D = y_true.shape
    d_kl = build_kl_divergence(enc_mu, enc_var)
    if decoder_type == 'bernoulli':
        _, dec_logits = dec_output
        neg_rec_err = expected_bernoulli_loglike(y_true, dec_logits)
    elif decoder_type == 'standard':
        dec_mu, dec_var = dec_output
        neg_rec_err = expected_diagonal_gaussian_loglike(y_true, dec_mu, dec_var)
    else:
        raise NotImplementedError
    return neg_rec_err / M - d_kl


def reparam_trick_sampling(mean, var_diag, nb_samples, seed, noise=None):
    """reference vae.py:282-296: mean + sqrt(var) * eps, eps (M,S,L) ~ N(0,1) (`noise` injects the draw)."""
    M, Ld = mean.shape
    if noise is None:
        g = torch.Generator(device=mean.device).manual_seed(int(seed))
        noise = torch.randn(M, nb_samples, Ld, generator=g, device=mean.device, dtype=mean.dtype)
    return mean.unsqueeze(1) + torch.sqrt(var_diag).unsqueeze(1) * noise


def expected_diagonal_gaussian_loglike(y, means, vars, weights=None, name='diag_gauss_expct'):
    """reference vae.py:201-250.  weights (N,K) branch: HIP reduction over (s,d) + K-cheap contraction."""
    if weights is None:
        # plain-VAE branch (vae.py:217-230): sum (y - mean)^2 / var + sum log var (no epsilon), K = 1
        if means.dim() != 3:
            means, vars = means.unsqueeze(1), vars.unsqueeze(1)
        M, S, Ld = means.shape
        if tuple(y.shape) != (M, Ld):
            raise AssertionError('y must have shape (M,L)')
        A = _svae_ops.DiagGaussLoglikeFn.apply(y, means.unsqueeze(1).contiguous(), vars.unsqueeze(1).contiguous(), 0.0)
        return -0.5 * A.sum() / S - M * Ld / 2. * float(np.log(2. * np.pi))
    if isinstance(means, LazyReconstruction):
        M, K, S, _ = means.x.shape
        Ld = y.shape[1]
        if tuple(weights.shape) != (M, K):
            raise AssertionError('shape mismatch')
        # -1/2 * (sum_nk w A) / S with the constant folded into the weights: the fused kernel then sees the final
        # dLoss/dA and the upstream gradient of this term is exactly 1 for loss = -elbo
        return -means.weighted_loglike(y, weights * (0.5 / S)) - M * Ld / 2. * float(np.log(2. * np.pi))
    else:
        M, K, S, Ld = means.shape
        if tuple(vars.shape) != tuple(means.shape) or tuple(weights.shape) != (M, K):
            raise AssertionError('shape mismatch')
        A = _svae_ops.DiagGaussLoglikeFn.apply(y, means, vars)
    sample_mean = (A * weights).sum() / S
    return -0.5 * sample_mean - M * Ld / 2. * float(np.log(2. * np.pi))
