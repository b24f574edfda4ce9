"""torch.autograd wrappers of the T2 / reconstruction HIP kernels (C ABI: include/vmp_hip.h).
No CPU fallback: tensors must be contiguous fp32 GPU tensors."""
import threading

import torch

from .. import _klinalg, _lib as L


def _c(t, name, shape=None):
    return L.dev_f32(t, name, shape)


class GradSeed(object):
    """The upstream gradient a caller will differentiate the ELBO with, announced beforehand: `tensor` (0-dim, on the
    device) is what it passes as grad_outputs, `value` its content.  FusedElboFn folds the factor into its one backward
    launch and recognises the tensor by address - no rescaling pass, no host read-back (graph capture).  The tensor must not
    be written to after it has been announced (FusedElboFn checks its version counter and falls back to rescaling)."""

    def __init__(self, value, device):
        self.value = float(value)
        self.tensor = torch.full((), float(value), dtype=torch.float32, device=device)


class PhiloxNoise(object):
    """Stand-in for the (N,K,L,S) noise tensor: "generate it in the kernel" (csrc/vmp_svae.hip, Philox4x32-7 keyed by
    `seed`; include/vmp_hip.h vmp_svae_estep_fwd_rng).  The reference draws eps inside the step the same way
    (models/svae.py:113-114).  materialise() returns the identical stream as a tensor."""

    def __init__(self, seed, nb_samples, seed_dev=None, epilogue=False):
        """seed_dev: a one-element int64 device tensor that holds the key instead of `seed` - read by the kernel when it
        RUNS, so that a launch captured in a HIP graph draws fresh noise per replay (in-kernel shapes only).
        epilogue: ask the E-step kernel that consumes this object to also do what the step does next with a cell's values
        (vmp_svae_estep_fwd_rng_epi): the one-draw-per-row sub-sampling (svae.py:122-151, 514), r = exp(log_z) and - where the
        kernel covers it (K = 16, L = 8) - per-block partials of the M-step's raw moments; SvaeEStepFn leaves them in
        .x_samples (N,L), .r_nk (N,K), .mom ((blocks,16,48) fp64 or None).  All three None when the shape is not in-kernel."""
        self.seed, self.S = int(seed) & 0xFFFFFFFFFFFFFFFF, int(nb_samples)
        self.seed_dev = seed_dev
        self.epilogue = bool(epilogue)
        self.x_samples = self.r_nk = self.mom = None

    def materialise(self, N, K, Ld, device):
        out = torch.empty(N, K, Ld, self.S, dtype=torch.float32, device=device)
        if self.seed_dev is not None:
            L.check(L.lib().vmp_svae_philox_noise_dev(L.ptr(self.seed_dev), N, K, Ld, self.S, L.ptr(out), L.stream()),
                    'vmp_svae_philox_noise_dev')
        else:
            L.check(L.lib().vmp_svae_philox_noise(self.seed, N, K, Ld, self.S, L.ptr(out), L.stream()), 'vmp_svae_philox_noise')
        return out


class SvaeEStepFn(torch.autograd.Function):
    """(eta1, eta2d, hk, Pk, bias, noise, mk, Wk, kappa, nu) -> (x (N,K,S,L), log_z (N,K), T' (N,K)).
    Gradients flow to eta1, eta2d (N,L), to hk, Pk, bias (K-sized, summed over n) and - for the Student-t theta of the
    SMM model (nu is not None), whose mu_k, L_k are trainable (experiments.py:160-161) - to mk, Wk, kappa.  For the
    Gaussian theta the reference stops the gradient (svae.py:211-214)."""

    @staticmethod
    def forward(ctx, eta1, eta2d, hk, Pk, bias, noise, mk, Wk, kappa, nu):
        eta1 = _c(eta1, 'eta1')
        N, Ld = eta1.shape
        eta2d = _c(eta2d, 'eta2_diag', (N, Ld))
        K = hk.shape[0]
        hk, Pk, bias = _c(hk, 'eta1_phi2', (K, Ld)), _c(Pk, 'P_k', (K, Ld, Ld)), _c(bias, 'bias_k', (K,))
        rng = noise if isinstance(noise, PhiloxNoise) else None
        if rng is None:
            noise = _c(noise, 'noise')
            if noise.dim() != 4 or tuple(noise.shape[:3]) != (N, K, Ld):
                raise L.VmpError('noise must have shape (N,K,L,S), got %s' % (tuple(noise.shape),))
            S = noise.shape[3]
        else:
            S = rng.S
        mk, Wk, kappa = _c(mk, 'm_k', (K, Ld)), _c(Wk, 'W_k', (K, Ld, Ld)), _c(kappa, 'kappa_k', (K,))
        nu = None if nu is None else _c(nu, 'nu_k', (K,))
        f32 = dict(dtype=torch.float32, device=eta1.device)
        x = torch.empty(N, K, S, Ld, **f32)
        lz = torch.empty(N, K, **f32)
        Tp = torch.empty(N, K, **f32)
        if rng is not None:
            rng.x_samples = rng.r_nk = rng.mom = None
        if rng is not None and rng.epilogue and L.lib().vmp_svae_rng_in_kernel(K, Ld, S):
            xs, r = torch.empty(N, Ld, **f32), torch.empty(N, K, **f32)
            nb = L.lib().vmp_svae_fwd_mom_blocks(N, K, Ld, S)
            mom = torch.empty(nb, 16, 48, dtype=torch.float64, device=eta1.device) if nb > 0 else None
            L.check(L.lib().vmp_svae_estep_fwd_rng_epi(L.ptr(eta1), L.ptr(eta2d), L.ptr(hk), L.ptr(Pk), L.ptr(bias), rng.seed,
                                                       L.ptr(rng.seed_dev), L.ptr(mk), L.ptr(Wk), L.ptr(kappa), L.ptr(nu), N, K, Ld,
                                                       S, L.ptr(x), L.ptr(lz), L.ptr(Tp), L.ptr(xs), L.ptr(r), L.ptr(mom),
                                                       0 if mom is None else mom.numel() * 8, L.stream()),
                    'vmp_svae_estep_fwd_rng_epi')
            rng.x_samples, rng.r_nk, rng.mom = xs, r, mom
        elif rng is not None and rng.seed_dev is not None:
            L.check(L.lib().vmp_svae_estep_fwd_rng_dev(L.ptr(eta1), L.ptr(eta2d), L.ptr(hk), L.ptr(Pk), L.ptr(bias),
                                                       L.ptr(rng.seed_dev), L.ptr(mk), L.ptr(Wk), L.ptr(kappa), L.ptr(nu), N, K,
                                                       Ld, S, L.ptr(x), L.ptr(lz), L.ptr(Tp), L.stream()),
                    'vmp_svae_estep_fwd_rng_dev')
        elif rng is not None:
            ws = None
            if not L.lib().vmp_svae_rng_in_kernel(K, Ld, S):     # shape outside the in-kernel path: same stream via a scratch tensor
                ws = torch.empty(N, K, Ld, S, **f32)
            L.check(L.lib().vmp_svae_estep_fwd_rng(L.ptr(eta1), L.ptr(eta2d), L.ptr(hk), L.ptr(Pk), L.ptr(bias), rng.seed,
                                                   L.ptr(mk), L.ptr(Wk), L.ptr(kappa), L.ptr(nu), N, K, Ld, S, L.ptr(x),
                                                   L.ptr(lz), L.ptr(Tp), L.ptr(ws), L.stream()), 'vmp_svae_estep_fwd_rng')
        else:
            L.check(L.lib().vmp_svae_estep_fwd(L.ptr(eta1), L.ptr(eta2d), L.ptr(hk), L.ptr(Pk), L.ptr(bias), L.ptr(noise),
                                               L.ptr(mk), L.ptr(Wk), L.ptr(kappa), L.ptr(nu), N, K, Ld, S, L.ptr(x),
                                               L.ptr(lz), L.ptr(Tp), L.stream()), 'vmp_svae_estep_fwd')
        ctx.save_for_backward(eta1, eta2d, hk, Pk, bias, mk, Wk, x, lz, *([nu] if nu is not None else []))
        ctx.dims = (N, K, Ld, S)
        return x, lz, Tp

    @staticmethod
    def backward(ctx, g_x, g_lz, g_T):
        sv = ctx.saved_tensors
        eta1, eta2d, hk, Pk, bias, mk, Wk, x, lz = sv[:9]
        nu = sv[9] if len(sv) > 9 else None
        N, K, Ld, S = ctx.dims
        f32 = dict(dtype=torch.float32, device=eta1.device)
        g_x = torch.zeros_like(x) if g_x is None else g_x.contiguous()
        g_lz = torch.zeros_like(lz) if g_lz is None else g_lz.contiguous()
        g_T = torch.zeros_like(lz) if g_T is None else g_T.contiguous()
        g_eta1 = torch.empty(N, Ld, **f32)
        g_eta2d = torch.empty(N, Ld, **f32)
        nblk = L.lib().vmp_svae_bwd_blocks_for(N, K, Ld, S, int(nu is not None))   # minibatch sizes: one partial row per tile
        PW = L.lib().vmp_svae_bwd_partial_words(Ld)
        partials = torch.empty(nblk, K, PW, **f32)
        L.check(L.lib().vmp_svae_estep_bwd_n(L.ptr(eta1), L.ptr(eta2d), L.ptr(hk), L.ptr(Pk), L.ptr(bias), L.ptr(mk),
                                             L.ptr(Wk), L.ptr(nu), L.ptr(x), L.ptr(lz), L.ptr(g_x), L.ptr(g_lz), L.ptr(g_T),
                                             N, K, Ld, S, L.ptr(g_eta1), L.ptr(g_eta2d), L.ptr(partials),
                                             partials.numel() * 4, nblk, L.stream()), 'vmp_svae_estep_bwd_n')
        # fixed-order fp64 reduction of the per-block partials + unpacking into the K-sized gradients: one launch
        g_hk, g_P, g_bias = torch.empty(K, Ld, **f32), torch.empty(K, Ld, Ld, **f32), torch.empty(K, **f32)
        g_mk = g_W = g_kappa = None
        if nu is not None:
            g_mk, g_W, g_kappa = torch.empty(K, Ld, **f32), torch.empty(K, Ld, Ld, **f32), torch.empty(K, **f32)
        L.check(L.lib().vmp_svae_bwd_reduce(L.ptr(partials), nblk, K, Ld, L.ptr(g_hk), L.ptr(g_P), L.ptr(g_bias),
                                            L.ptr(g_mk), L.ptr(g_W), L.ptr(g_kappa), L.stream()), 'vmp_svae_bwd_reduce')
        return g_eta1, g_eta2d, g_hk, g_P, g_bias, None, g_mk, g_W, g_kappa, None


class DiagGaussLoglikeFn(torch.autograd.Function):
    """A_nk = sum_{s,d} (y - mean)^2 / var + log(var + eps), eps = 1e-8 in the weights branch (reference vae.py:240),
    0 in the plain-VAE branch (vae.py:225); gradients to mean, var."""

    @staticmethod
    def forward(ctx, y, mean, var, eps=1e-8):
        y = _c(y, 'y')
        mean = _c(mean, 'means')
        var = _c(var, 'vars', tuple(mean.shape))
        N, K, S, Dy = mean.shape
        if tuple(y.shape) != (N, Dy):
            raise L.VmpError('y must have shape (N,Dy)')
        A = torch.empty(N, K, dtype=torch.float32, device=y.device)
        L.check(L.lib().vmp_diag_gauss_loglike_fwd(L.ptr(y), L.ptr(mean), L.ptr(var), N, K, S, Dy, float(eps), L.ptr(A),
                                                   L.stream()), 'vmp_diag_gauss_loglike_fwd')
        ctx.save_for_backward(y, mean, var)
        ctx.eps = float(eps)
        return A

    @staticmethod
    def backward(ctx, gA):
        y, mean, var = ctx.saved_tensors
        N, K, S, Dy = mean.shape
        gA = gA.contiguous()
        gm, gv = torch.empty_like(mean), torch.empty_like(var)
        L.check(L.lib().vmp_diag_gauss_loglike_bwd(L.ptr(y), L.ptr(mean), L.ptr(var), L.ptr(gA), N, K, S, Dy, ctx.eps,
                                                   L.ptr(gm), L.ptr(gv), L.stream()), 'vmp_diag_gauss_loglike_bwd')
        return None, gm, gv, None


class BernoulliRowsFn(torch.autograd.Function):
    """rows_nks = sum_d m_nd (-log(1 + exp(-logit_nksd y_nd)))  (reference vae.py:190-192, losses.py:61-69);
    y in {-1,+1} (N,D), logits (N,K,S,D), mask (N,D) bool or None.  Gradient to the logits."""

    @staticmethod
    def forward(ctx, y, logits, mask=None):
        y = _c(y, 'y_binary')
        logits = _c(logits, 'logits')
        N, K, S, D = logits.shape
        if tuple(y.shape) != (N, D):
            raise AssertionError('y_binary must have shape (N,D)')
        m8 = None
        if mask is not None:
            if tuple(mask.shape) != (N, D):
                raise AssertionError('mask must have shape (N,D)')
            m8 = mask.to(torch.uint8).contiguous()
        rows = torch.empty(N, K, S, dtype=torch.float32, device=y.device)
        L.check(L.lib().vmp_bernoulli_rows_fwd(L.ptr(y), L.ptr(logits), L.ptr(m8), N, K, S, D, L.ptr(rows), L.stream()),
                'vmp_bernoulli_rows_fwd')
        ctx.save_for_backward(y, logits)
        ctx.m8 = m8
        return rows

    @staticmethod
    def backward(ctx, g_rows):
        y, logits = ctx.saved_tensors
        N, K, S, D = logits.shape
        g_rows = g_rows.contiguous().float()
        gl = torch.empty_like(logits)
        L.check(L.lib().vmp_bernoulli_rows_bwd(L.ptr(y), L.ptr(logits), L.ptr(ctx.m8), L.ptr(g_rows), N, K, S, D, L.ptr(gl),
                                               L.stream()), 'vmp_bernoulli_rows_bwd')
        return None, gl, None


def gauss_logprob_nat(x, eta1, eta2, weights=None):
    """gaussian.log_probability_nat (reference gaussian.py:30-71): (N,K) log N(x_n | eta_nk) + log w_k, normalised
    over k.  Forward only."""
    x = _c(x.detach(), 'x')
    N, D = x.shape
    if eta1.dim() != 3:
        raise AssertionError("eta1 must be of shape (N,K,D). Its shape is %s." % str(tuple(eta1.shape)))
    K = eta1.shape[1]
    eta1 = _c(eta1.detach(), 'eta1', (N, K, D))
    eta2 = _c(eta2.detach(), 'eta2', (N, K, D, D))
    lw = None if weights is None else _c(torch.log(weights.detach()).float(), 'weights', (K,))
    out = torch.empty(N, K, dtype=torch.float32, device=x.device)
    L.check(L.lib().vmp_gauss_logprob_nat(L.ptr(x), L.ptr(eta1), L.ptr(eta2), L.ptr(lw), N, K, D, L.ptr(out), L.stream()),
            'vmp_gauss_logprob_nat')
    return out


class GaussLogprobPerSampFn(torch.autograd.Function):
    """gaussian.log_probability_nat_per_samp (reference gaussian.py:74-105): (x (N,K,S,D), eta1 (N,K,D), eta2 (N,K,D,D)) ->
    (N,K,S), differentiable in all three (vmp_gauss_logprob_nat_per_samp / _bwd)."""

    @staticmethod
    def forward(ctx, x, eta1, eta2):
        x = _c(x, 'x_samps')
        if x.dim() != 4:
            raise AssertionError('x_samps must be of shape (N,K,S,D)')
        N, K, S, D = x.shape
        eta1 = _c(eta1, 'eta1', (N, K, D))
        eta2 = _c(eta2, 'eta2', (N, K, D, D))
        out = torch.empty(N, K, S, dtype=torch.float32, device=x.device)
        L.check(L.lib().vmp_gauss_logprob_nat_per_samp(L.ptr(x), L.ptr(eta1), L.ptr(eta2), N, K, S, D, L.ptr(out),
                                                       L.stream()), 'vmp_gauss_logprob_nat_per_samp')
        ctx.save_for_backward(x, eta1, eta2)
        return out

    @staticmethod
    def backward(ctx, g):
        x, eta1, eta2 = ctx.saved_tensors
        N, K, S, D = x.shape
        g = g.contiguous().float()
        gx, ge1, ge2 = torch.empty_like(x), torch.empty_like(eta1), torch.empty_like(eta2)
        L.check(L.lib().vmp_gauss_logprob_nat_per_samp_bwd(L.ptr(x), L.ptr(eta1), L.ptr(eta2), L.ptr(g), N, K, S, D, L.ptr(gx),
                                                           L.ptr(ge1), L.ptr(ge2), L.stream()), 'vmp_gauss_logprob_nat_per_samp_bwd')
        return gx, ge1, ge2


def gauss_logprob_per_samp(x_samps, eta1, eta2):
    """gaussian.log_probability_nat_per_samp (reference gaussian.py:74-105): (N,K,S); differentiable."""
    return GaussLogprobPerSampFn.apply(x_samps, eta1, eta2)


class StudentTLogprobFn(torch.autograd.Function):
    """(y (N,K,S,D), mu (K,D), W (K,D,D) lower with W^T W = Sigma^-1, cst (K), v (K)) -> cst_k - 1/2 (v_k + D)
    log1p(|W_k (y - mu_k)|^2 / v_k) (N,K,S); gradients to y, mu, W, cst (v: constant, as the reference's DoF)."""

    @staticmethod
    def forward(ctx, y, mu, W, cst, v):
        y = _c(y, 'y')
        N, K, S, D = y.shape
        mu, W, cst, v = _c(mu, 'mu', (K, D)), _c(W, 'W', (K, D, D)), _c(cst, 'cst', (K,)), _c(v, 'v', (K,))
        out = torch.empty(N, K, S, dtype=torch.float32, device=y.device)
        L.check(L.lib().vmp_student_t_logprob(L.ptr(y), L.ptr(mu), L.ptr(W), L.ptr(cst), L.ptr(v), N, K, S, D, L.ptr(out),
                                              L.stream()), 'vmp_student_t_logprob')
        ctx.save_for_backward(y, mu, W, v)
        return out

    @staticmethod
    def backward(ctx, g):
        y, mu, W, v = ctx.saved_tensors
        N, K, S, D = y.shape
        g = g.contiguous().float()
        gy = torch.empty_like(y)
        nb = L.lib().vmp_student_t_bwd_blocks(N, S)
        TRI = D * (D + 1) // 2
        part = torch.empty(nb, K, D + TRI + 1, dtype=torch.float32, device=y.device)
        L.check(L.lib().vmp_student_t_logprob_bwd(L.ptr(y), L.ptr(mu), L.ptr(W), L.ptr(v), L.ptr(g), N, K, S, D, L.ptr(gy),
                                                  L.ptr(part), L.stream()), 'vmp_student_t_logprob_bwd')
        tot = part.double().sum(0)                                   # fixed order over the blocks, fp64
        gmu = tot[:, :D].float()
        gW = torch.zeros(K, D, D, dtype=torch.float32, device=y.device)
        ii, jj = torch.tril_indices(D, D, device=y.device)
        gW[:, ii, jj] = tot[:, D:D + TRI].float()
        return gy, gmu, gW, tot[:, D + TRI].float(), None


def student_t_logprob(y, mu, sigma, v):
    """student_t.log_probability_per_samp (reference student_t.py:7-39,59-61): (N,K,S).  The K scale matrices are
    factorised once (K-sized, torch fp64, differentiable) instead of being tiled to (N,K,S,D,D); gradients reach y, mu and
    sigma through vmp_student_t_logprob_bwd."""
    import math
    N, K, S, D = y.shape
    if tuple(mu.shape) != (K, D) or tuple(sigma.shape) != (K, D, D) or tuple(v.shape) != (K,):
        raise AssertionError('shape mismatch')
    sig = sigma.double()
    Lc = _klinalg.cholesky(0.5 * (sig + sig.transpose(-1, -2)))
    eye = torch.eye(D, dtype=Lc.dtype, device=Lc.device).expand_as(Lc)
    W = torch.linalg.solve_triangular(Lc, eye, upper=False)
    vd = v.detach().double()
    cst = (torch.lgamma(0.5 * (vd + D)) - torch.lgamma(0.5 * vd) - 0.5 * D * torch.log(math.pi * vd)
           - torch.log(torch.diagonal(Lc, dim1=-2, dim2=-1)).sum(-1))
    return StudentTLogprobFn.apply(y, mu.float(), W.float().contiguous(), cst.float().contiguous(), v.detach().float())


DECODER_PARAM_NAMES = ('layer_0/kernel', 'layer_0/bias', 'layer_1/kernel', 'layer_1/bias', 'gaussian_output/kernel',
                       'gaussian_output/bias', 'shortcut/W', 'shortcut/b1', 'shortcut/b2')


def _decoder_dims(x, params):
    W0, b0, W1, b1, W2, b2, Ws, bs1, bs2 = params
    Ld, U = W0.shape
    Dy = Ws.shape[1]
    shapes = ((Ld, U), (U,), (U, U), (U,), (U, 2 * Dy), (2 * Dy,), (Ld, Dy), (Dy,), (Dy,))
    for n, p, s in zip(DECODER_PARAM_NAMES, params, shapes):
        if tuple(p.shape) != s:
            raise AssertionError('decoder parameter %s has shape %s, expected %s' % (n, tuple(p.shape), s))
    if x.shape[-1] != Ld:
        raise AssertionError('decoder input has %d features, layer_0/kernel expects %d' % (x.shape[-1], Ld))
    return Ld, U, Dy


def fused_decoder_supported(Ld, U, Dy):
    """Compiled range of the fused MFMA decoder kernels (csrc/vmp_decoder.hip)."""
    return 1 <= Ld <= 8 and 1 <= Dy <= 8 and 1 <= U <= 64


class DecoderLoglikeFn(torch.autograd.Function):
    """(y (N,Dy), x (N,K,S,L), 9 decoder parameters) -> A (N,K) = sum_{s,d} (y - mean)^2 / var + log(var + 1e-8) with
    (mean, var) = decoder(x) (reference vae.py:75-128 + :233-248), one fused HIP kernel each way: the
    (N,K,S,U) activations and (N,K,S,Dy) outputs never reach memory.  Gradients to x and the 9 parameters."""

    @staticmethod
    def forward(ctx, y, x, *params):
        x = _c(x, 'x_k_samples')
        if x.dim() != 4:
            raise L.VmpError('x must have shape (N,K,S,L)')
        N, K, S, _ = x.shape
        params = [_c(p, n) for p, n in zip(params, DECODER_PARAM_NAMES)]
        Ld, U, Dy = _decoder_dims(x, params)
        y = _c(y, 'y', (N, Dy))
        ll = torch.empty(N, K, S, dtype=torch.float32, device=x.device)
        L.check(L.lib().vmp_decoder_loglike_fwd(L.ptr(x), L.ptr(y), *[L.ptr(p) for p in params], N, K, S, Ld, Dy, U,
                                                L.ptr(ll), None, None, L.stream()), 'vmp_decoder_loglike_fwd')
        ctx.save_for_backward(y, x, *params)
        ctx.dims = (N, K, S, Ld, Dy, U)
        return ll.sum(-1)

    @staticmethod
    def backward(ctx, gA):
        sv = ctx.saved_tensors
        y, x, params = sv[0], sv[1], sv[2:]
        N, K, S, Ld, Dy, U = ctx.dims
        gA = gA.contiguous().float()
        dx = torch.empty_like(x)
        PW = L.lib().vmp_decoder_param_words(Ld, U, Dy)
        dp = torch.empty(PW, dtype=torch.float32, device=x.device)
        nbytes = L.lib().vmp_decoder_workspace_bytes(N, K, S, Ld, U, Dy)
        ws = L.workspace(x.device, nbytes)
        L.check(L.lib().vmp_decoder_loglike_bwd(L.ptr(x), L.ptr(y), L.ptr(gA), *[L.ptr(p) for p in params], N, K, S, Ld, Dy,
                                                U, L.ptr(dx), L.ptr(dp), None, L.ptr(ws), nbytes, L.stream()),
                'vmp_decoder_loglike_bwd')
        return (None, dx) + tuple(_split_flat(dp, params))


def _split_flat(dp, params):
    grads, o = [], 0
    for p in params:
        grads.append(dp[o:o + p.numel()].reshape(p.shape))
        o += p.numel()
    return grads


class DecoderWeightedLoglikeFn(torch.autograd.Function):
    """(y, x (N,K,S,L), w (N,K), 9 decoder parameters) -> scalar  sum_nk w_nk A_nk  with A as in DecoderLoglikeFn -
    the contraction einsum('nksd,nk->') of reference vae.py:240.  Because dLoss/dA_nk = w_nk is an INPUT, the value
    and every gradient come out of ONE launch of the backward kernel (which recomputes the forward anyway): the
    separate forward launch of DecoderLoglikeFn is saved.  Gradients: x, w (= A) and the 9 parameters."""

    @staticmethod
    def forward(ctx, y, x, w, *params):
        x = _c(x, 'x_k_samples')
        if x.dim() != 4:
            raise L.VmpError('x must have shape (N,K,S,L)')
        N, K, S, _ = x.shape
        params = [_c(p, n) for p, n in zip(params, DECODER_PARAM_NAMES)]
        Ld, U, Dy = _decoder_dims(x, params)
        y = _c(y, 'y', (N, Dy))
        w = _c(w, 'weights', (N, K))
        ll = torch.empty(N, K, S, dtype=torch.float32, device=x.device)
        args = [L.ptr(p) for p in params]
        if not any(ctx.needs_input_grad):
            L.check(L.lib().vmp_decoder_loglike_fwd(L.ptr(x), L.ptr(y), *args, N, K, S, Ld, Dy, U, L.ptr(ll), None, None,
                                                    L.stream()), 'vmp_decoder_loglike_fwd')
            return (ll.sum(-1) * w).sum()
        dx = torch.empty_like(x)
        dp = torch.empty(L.lib().vmp_decoder_param_words(Ld, U, Dy), dtype=torch.float32, device=x.device)
        nbytes = L.lib().vmp_decoder_workspace_bytes(N, K, S, Ld, U, Dy)
        ws = L.workspace(x.device, nbytes)
        L.check(L.lib().vmp_decoder_loglike_bwd(L.ptr(x), L.ptr(y), L.ptr(w), *args, N, K, S, Ld, Dy, U, L.ptr(dx),
                                                L.ptr(dp), L.ptr(ll), L.ptr(ws), nbytes, L.stream()),
                'vmp_decoder_loglike_bwd')
        A = ll.sum(-1)
        ctx.save_for_backward(A, dx, dp)
        ctx.pshapes = [tuple(p.shape) for p in params]
        return (A * w).sum()

    @staticmethod
    def backward(ctx, g):
        A, dx, dp = ctx.saved_tensors
        # dx is (N,K,S,L)-sized: rescaling it costs a full pass over HBM, so the (usual) upstream gradient of exactly
        # 1 - compute_elbo folds its -1/2S into the weights - is detected with one scalar read-back instead
        if torch.cuda.is_current_stream_capturing():     # no host read-back inside a graph capture
            dx, dp = dx * g, dp * g
        else:
            gs = float(g)
            if gs != 1.0:
                dx = dx * gs
                dp = dp * gs
        grads, o = [], 0
        for shp in ctx.pshapes:
            n = 1
            for v in shp:
                n *= v
            grads.append(dp[o:o + n].reshape(shp))
            o += n
        return (None, dx, A * g) + tuple(grads)


_TAIL_WS = {}
_TAIL_WS_MAX = 16
_TAIL_LOCK = threading.Lock()


def _tail_workspace(device):
    """Scratch of vmp_svae_elbo_tail / vmp_decoder_elbo (per-block partial sums), one per device, stream and host thread;
    least recently used entries beyond _TAIL_WS_MAX are dropped (streams that went away do not pin memory)."""
    key = (device.index, L._raw_stream(device.index), threading.get_ident())
    with _TAIL_LOCK:
        ws = _TAIL_WS.pop(key, None)
        if ws is None:
            ws = torch.empty(L.lib().vmp_svae_elbo_tail_workspace_bytes(), dtype=torch.uint8, device=device)
        _TAIL_WS[key] = ws
        while len(_TAIL_WS) > _TAIL_WS_MAX:
            del _TAIL_WS[next(iter(_TAIL_WS))]
    return ws


def release_tail_workspaces(stream):
    """Forget (and return) the tail scratch of `stream`: a graph capture takes ownership of the buffer its captured launch
    points into; a temporary warm-up stream's buffer is simply dropped."""
    sid = stream.cuda_stream
    with _TAIL_LOCK:
        return [_TAIL_WS.pop(k) for k in [k for k in _TAIL_WS if k[1] == sid]]


class FusedElboFn(torch.autograd.Function):
    """(y, x (N,K,S,L), log_z (N,K), T' (N,K), 9 decoder parameters) -> (elbo, rec, reg, r): compute_elbo of reference
    svae.py:199-262 for the fused decoder in two launches (decoder value + gradients with r = exp(log_z) formed inside
    the kernel; then the reduction of its per-block parameter partials beside the scalar tail) instead of those plus ~24
    (N,K)-sized torch launches.  `seed` is the upstream gradient the caller WILL pass for elbo (a 0-dim tensor, e.g. -1 for
    loss = -elbo; None = +1): the stored gradients are pre-multiplied by it, and backward only rescales - with a few
    launches - when handed a different tensor.  rec, reg and r are returned for reporting / the M-step and carry no
    gradient."""

    @staticmethod
    def forward(ctx, y, x, lz, Tp, seed, sigma, *params):
        x = _c(x, 'x_k_samples')
        if x.dim() != 4:
            raise L.VmpError('x must have shape (N,K,S,L)')
        N, K, S, _ = x.shape
        params = [_c(p, n) for p, n in zip(params, DECODER_PARAM_NAMES)]
        Ld, U, Dy = _decoder_dims(x, params)
        y = _c(y, 'y', (N, Dy))
        lz = _c(lz, 'log_z', (N, K))
        Tp = _c(Tp, 'T_prime', (N, K))
        f32 = dict(dtype=torch.float32, device=x.device)
        ll = torch.empty(N, K, S, **f32)
        dx = torch.empty_like(x)
        dp = torch.empty(L.lib().vmp_decoder_param_words(Ld, U, Dy), **f32)
        nbytes = L.lib().vmp_decoder_workspace_bytes(N, K, S, Ld, U, Dy)
        ws = L.workspace(x.device, nbytes)
        sigma = float(sigma)
        scal = torch.empty(3, **f32)
        g_lz, g_Tp, r = torch.empty(N, K, **f32), torch.empty(N, K, **f32), torch.empty(N, K, **f32)
        tws = _tail_workspace(x.device)
        if N > 0:
            # decoder value + gradients (r = exp(log_z) formed in the kernel), then partial reduction and scalar tail together
            L.check(L.lib().vmp_decoder_elbo(L.ptr(x), L.ptr(y), L.ptr(lz), L.ptr(Tp), sigma, *[L.ptr(p) for p in params], N, K, S,
                                             Ld, Dy, U, L.ptr(dx), L.ptr(dp), L.ptr(ll), L.ptr(scal), L.ptr(g_lz), L.ptr(g_Tp),
                                             L.ptr(r), L.ptr(ws), nbytes, L.ptr(tws), tws.numel(), L.stream()), 'vmp_decoder_elbo')
        else:
            L.check(L.lib().vmp_decoder_loglike_bwd_logw(L.ptr(x), L.ptr(y), L.ptr(lz), -sigma * 0.5 / S, *[L.ptr(p) for p in params],
                                                         N, K, S, Ld, Dy, U, L.ptr(dx), L.ptr(dp), L.ptr(ll), L.ptr(ws), nbytes,
                                                         L.stream()), 'vmp_decoder_loglike_bwd_logw')
            L.check(L.lib().vmp_svae_elbo_tail(L.ptr(lz), L.ptr(Tp), L.ptr(ll), N, K, S, Dy, sigma, L.ptr(scal), L.ptr(g_lz),
                                               L.ptr(g_Tp), L.ptr(r), L.ptr(tws), tws.numel(), L.stream()), 'vmp_svae_elbo_tail')
        ctx.save_for_backward(dx, dp, g_lz, g_Tp)
        ctx.pshapes = [tuple(p.shape) for p in params]
        ctx.seed, ctx.sigma = seed, sigma
        ctx.seed_version = None if seed is None else seed._version      # the announced tensor must not be written to afterwards
        elbo, rec, reg = scal[0], scal[1], scal[2]
        ctx.mark_non_differentiable(rec, reg, r)
        ctx.set_materialize_grads(False)        # no zero-filled gradients for the three reporting outputs
        return elbo, rec, reg, r

    @staticmethod
    def backward(ctx, g, _g_rec, _g_reg, _g_r):
        dx, dp, g_lz, g_Tp = ctx.saved_tensors
        if g is None:
            return (None,) * (6 + len(ctx.pshapes))
        # recognised = the very tensor that was announced (same storage) AND untouched since (an in-place rescale or sign flip of
        # the announced tensor bumps its version counter: its content is then no longer the sigma folded into the gradients)
        same = ctx.seed is not None and g.data_ptr() == ctx.seed.data_ptr() and ctx.seed._version == ctx.seed_version
        if not same:
            # the stored gradients are sigma * d elbo and the caller differentiates with another upstream tensor: rescale
            # (dx is (N,K,S,L)-sized - outside a graph capture one scalar read-back decides whether that pass is needed)
            if torch.cuda.is_current_stream_capturing():
                f = g / ctx.sigma
            else:
                f = float(g) / ctx.sigma
            if not (isinstance(f, float) and f == 1.0):
                dx, dp, g_lz, g_Tp = dx * f, dp * f, g_lz * f, g_Tp * f
        grads, o = [], 0
        for shp in ctx.pshapes:
            n = 1
            for v in shp:
                n *= v
            grads.append(dp[o:o + n].reshape(shp))
            o += n
        return (None, dx, g_lz, g_Tp, None, None) + tuple(grads)


def decoder_outputs(x, params):
    """(mean, var) of the decoder on x (..., L) through the fused forward kernel (no gradient)."""
    shape = tuple(x.shape)
    x2 = _c(x.detach().reshape(-1, 1, 1, shape[-1]), 'x')
    params = [_c(p.detach(), n) for p, n in zip(params, DECODER_PARAM_NAMES)]
    Ld, U, Dy = _decoder_dims(x2, params)
    R = x2.shape[0]
    mean = torch.empty(R, Dy, dtype=torch.float32, device=x2.device)
    var = torch.empty(R, Dy, dtype=torch.float32, device=x2.device)
    L.check(L.lib().vmp_decoder_loglike_fwd(L.ptr(x2), None, *[L.ptr(p) for p in params], R, 1, 1, Ld, Dy, U, None,
                                            L.ptr(mean), L.ptr(var), L.stream()), 'vmp_decoder_loglike_fwd')
    return mean.reshape(shape[:-1] + (Dy,)), var.reshape(shape[:-1] + (Dy,))


class PhiPrepFn(torch.autograd.Function):
    """(mu_k, L_k_raw, pi_k_raw [, the 5 natural GMM theta tensors]) -> (h_k = mu_k, P_k = L L^T, bias_k [, m_k, W_k, kappa_k])
    - the K-sized inputs of the fused E-step - one launch forward, one backward (reference svae.py:342-358 and :70-92;
    csrc/vmp_prep.hip).  With theta the packed theta (theta_pack_gmm: no gradient, as the reference's stop_gradient,
    svae.py:211-214) comes out of the SAME launch."""

    @staticmethod
    def forward(ctx, mu_k, L_raw, pi_raw, *theta):
        mu_k = _c(mu_k, 'phi_gmm/mu_k')
        K, Ld = mu_k.shape
        L_raw = _c(L_raw, 'phi_gmm/L_k', (K, Ld, Ld))
        pi_raw = _c(pi_raw, 'phi_gmm/log_pi_k', (K,))
        f32 = dict(dtype=torch.float32, device=mu_k.device)
        Lk, P, bias = torch.empty(K, Ld, Ld, **f32), torch.empty(K, Ld, Ld, **f32), torch.empty(K, **f32)
        ctx.save_for_backward(mu_k, L_raw, pi_raw)
        ctx.n_theta = len(theta)
        ctx.set_materialize_grads(False)           # no zero-filled gradients for the packed-theta outputs
        if theta:
            if len(theta) != 5:
                raise L.VmpError('PhiPrepFn: theta must be the 5 natural NIW / Dirichlet tensors')
            alpha, A, b, beta, v_hat = [_c(t, n) for t, n in zip(theta, ('alpha', 'A', 'b', 'beta', 'v_hat'))]
            m, W, kappa = torch.empty(K, Ld, **f32), torch.empty(K, Ld, Ld, **f32), torch.empty(K, **f32)
            L.check(L.lib().vmp_svae_prep_fwd(L.ptr(mu_k), L.ptr(L_raw), L.ptr(pi_raw), L.ptr(alpha), L.ptr(A), L.ptr(b),
                                              L.ptr(beta), L.ptr(v_hat), K, Ld, L.ptr(Lk), L.ptr(P), L.ptr(bias), L.ptr(m),
                                              L.ptr(W), L.ptr(kappa), L.stream()), 'vmp_svae_prep_fwd')
            ctx.mark_non_differentiable(m, W, kappa)
            return mu_k.view_as(mu_k), P, bias, m, W, kappa
        L.check(L.lib().vmp_svae_phi_prep_fwd(L.ptr(mu_k), L.ptr(L_raw), L.ptr(pi_raw), K, Ld, L.ptr(Lk), L.ptr(P),
                                              L.ptr(bias), L.stream()), 'vmp_svae_phi_prep_fwd')
        return mu_k.view_as(mu_k), P, bias         # h_k IS mu_k (svae.py:345): a view, not a copy launch

    @staticmethod
    def backward(ctx, g_hk, g_P, g_bias, *_unused):
        mu_k, L_raw, pi_raw = ctx.saved_tensors
        K, Ld = mu_k.shape
        z = lambda g, ref: torch.zeros_like(ref) if g is None else g.contiguous().float()
        g_hk = z(g_hk, mu_k)
        g_P = z(g_P, L_raw)
        g_bias = z(g_bias, pi_raw)
        g_mu, g_L, g_pi = torch.empty_like(mu_k), torch.empty_like(L_raw), torch.empty_like(pi_raw)
        L.check(L.lib().vmp_svae_phi_prep_bwd(L.ptr(mu_k), L.ptr(L_raw), L.ptr(pi_raw), L.ptr(g_hk), L.ptr(g_P),
                                              L.ptr(g_bias), K, Ld, L.ptr(g_mu), L.ptr(g_L), L.ptr(g_pi), L.stream()),
                'vmp_svae_phi_prep_bwd')
        return (g_mu, g_L, g_pi) + (None,) * ctx.n_theta


def theta_pack_gmm(theta):
    """(m_k, W_k, kappa_k) of a natural NIW / Dirichlet theta (no gradient, as the reference's stop_gradient)."""
    alpha, A, b, beta, v_hat = [_c(t.detach(), n) for t, n in zip(theta, ('alpha', 'A', 'b', 'beta', 'v_hat'))]
    K, Ld = b.shape
    f32 = dict(dtype=torch.float32, device=b.device)
    m, W, kappa = torch.empty(K, Ld, **f32), torch.empty(K, Ld, Ld, **f32), torch.empty(K, **f32)
    L.check(L.lib().vmp_svae_theta_pack(L.ptr(alpha), L.ptr(A), L.ptr(b), L.ptr(beta), L.ptr(v_hat), K, Ld, L.ptr(m),
                                        L.ptr(W), L.ptr(kappa), L.stream()), 'vmp_svae_theta_pack')
    return m, W, kappa


def cvi_update(gmm_prior, theta, stats, rho, want_star=True, rho_dev=None):
    """theta <- (1 - rho) theta + rho theta*, theta* = prior + raw moments (+1 on v_hat), in place, one launch.
    Returns theta* (5 tensors) when want_star."""
    pri = [_c(t.detach(), 'prior') for t in gmm_prior]
    K, Ld = pri[2].shape
    for t in theta:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise L.VmpError('theta must be contiguous fp32 GPU tensors')
    stats = stats.contiguous()
    if stats.dtype != torch.float64 or tuple(stats.shape) != (K, 2 + Ld + Ld * Ld):
        raise L.VmpError('stats must be fp64 (K, 2+L+L*L)')
    star = [torch.empty_like(t) for t in theta] if want_star else [None] * 5
    L.check(L.lib().vmp_svae_cvi_update(L.ptr(stats), *[L.ptr(t) for t in pri], *[L.ptr(t) for t in theta],
                                        *[L.ptr(t) for t in star], L.ptr(rho_dev), float(rho), K, Ld, L.stream()),
            'vmp_svae_cvi_update')
    for t in theta:
        torch.autograd.graph.increment_version(t)
    return star if want_star else None


def mom_cvi(mom, gmm_prior=None, theta=None, rho=0.0, want_star=True, rho_dev=None, want_stats=True):
    """The fused E-step kernel's moment partials (PhiloxNoise.mom, (blocks,16,48) fp64) -> raw moments (K, 2+L+L*L) fp64 and, when
    theta is given, svae.m_step + update_gmm_params from them (svae.py:154-176, 376-403) in the same launch (vmp_svae_mom_cvi).
    Returns (stats or None, theta* or None)."""
    nb, K, Ld = mom.shape[0], 16, 8
    stats = torch.empty(K, 2 + Ld + Ld * Ld, dtype=torch.float64, device=mom.device) if (want_stats or theta is None) else None
    if theta is None:
        L.check(L.lib().vmp_svae_mom_cvi(L.ptr(mom), nb, *([None] * 15), None, 0.0, K, Ld, L.ptr(stats), L.stream()), 'vmp_svae_mom_cvi')
        return stats, None
    pri = [_c(t.detach(), 'prior') for t in gmm_prior]
    for t in theta:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise L.VmpError('theta must be contiguous fp32 GPU tensors')
    star = [torch.empty_like(t) for t in theta] if want_star else [None] * 5
    L.check(L.lib().vmp_svae_mom_cvi(L.ptr(mom), nb, *[L.ptr(t) for t in pri], *[L.ptr(t) for t in theta],
                                     *[L.ptr(t) for t in star], L.ptr(rho_dev), float(rho), K, Ld, L.ptr(stats), L.stream()),
            'vmp_svae_mom_cvi')
    for t in theta:
        torch.autograd.graph.increment_version(t)
    return stats, (star if want_star else None)


STATS_CVI_MAX_ROWS = 512          # SMALL_STATS_MAX_N of the library


def stats_cvi(x_samples, r_nk, gmm_prior, theta, rho, want_star=True, rho_dev=None):
    """M-step moments of a small batch (<= 512 rows) AND cvi_update from them in one launch (vmp_svae_stats_cvi): the
    single-process whole-minibatch training step.  Returns (stats (K, 2+L+L*L) fp64, theta* or None)."""
    x = _c(x_samples.detach(), 'x_samples')
    N, Ld = x.shape
    pri = [_c(t.detach(), 'prior') for t in gmm_prior]
    K = pri[2].shape[0]
    r = _c(r_nk.detach(), 'r_nk', (N, K))
    for t in theta:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise L.VmpError('theta must be contiguous fp32 GPU tensors')
    stats = torch.empty(K, 2 + Ld + Ld * Ld, dtype=torch.float64, device=x.device)
    star = [torch.empty_like(t) for t in theta] if want_star else [None] * 5
    L.check(L.lib().vmp_svae_stats_cvi(L.ptr(x), L.ptr(r), N, *[L.ptr(t) for t in pri], *[L.ptr(t) for t in theta],
                                       *[L.ptr(t) for t in star], L.ptr(rho_dev), float(rho), K, Ld, L.ptr(stats), L.stream()),
            'vmp_svae_stats_cvi')
    for t in theta:
        torch.autograd.graph.increment_version(t)
    return stats, (star if want_star else None)


class GaussMLPFn(torch.autograd.Function):
    """(x (R,L), head scale, 9 parameters) -> (out1, out2) (R,Dy) of the two-tanh-layer Gaussian-head MLP with shortcut
    (reference vae.py:75-128) through the fused MFMA kernels, differentiable: one launch forward, one (+ the partial
    reduce) backward.  Head scale 1: 'standard' (mean, var); -1/2: the encoder's 'natparam' head (eta1, -1/2 var)
    (vae.py:38-42,108-111) - applied inside the kernels."""

    @staticmethod
    def forward(ctx, x, var_scale, *params):
        x = _c(x, 'mlp input')
        if x.dim() != 2:
            raise L.VmpError('input must have shape (R, L)')
        params = [_c(p, n) for p, n in zip(params, DECODER_PARAM_NAMES)]
        Ld, U, Dy = _decoder_dims(x, params)
        R = x.shape[0]
        mean = torch.empty(R, Dy, dtype=torch.float32, device=x.device)
        var = torch.empty(R, Dy, dtype=torch.float32, device=x.device)
        L.check(L.lib().vmp_mlp_gauss_head_fwd(L.ptr(x), *[L.ptr(p) for p in params], R, Ld, Dy, U, float(var_scale), L.ptr(mean),
                                               L.ptr(var), L.stream()), 'vmp_mlp_gauss_head_fwd')
        ctx.save_for_backward(x, *params)
        ctx.dims = (R, Ld, Dy, U)
        ctx.var_scale = float(var_scale)
        return mean, var

    @staticmethod
    def backward(ctx, g_mean, g_var):
        sv = ctx.saved_tensors
        x, params = sv[0], sv[1:]
        R, Ld, Dy, U = ctx.dims
        g_mean = torch.zeros(R, Dy, dtype=torch.float32, device=x.device) if g_mean is None else g_mean.contiguous().float()
        g_var = torch.zeros(R, Dy, dtype=torch.float32, device=x.device) if g_var is None else g_var.contiguous().float()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dp = torch.empty(L.lib().vmp_decoder_param_words(Ld, U, Dy), dtype=torch.float32, device=x.device)
        nbytes = L.lib().vmp_decoder_workspace_bytes(R, 1, 1, Ld, U, Dy)
        ws = L.workspace(x.device, nbytes)
        L.check(L.lib().vmp_mlp_gauss_head_bwd(L.ptr(x), L.ptr(g_mean), L.ptr(g_var), ctx.var_scale, *[L.ptr(p) for p in params],
                                               R, Ld, Dy, U, L.ptr(dx), L.ptr(dp), L.ptr(ws), nbytes, L.stream()),
                'vmp_mlp_gauss_head_bwd')
        return (dx, None) + tuple(_split_flat(dp, params))
