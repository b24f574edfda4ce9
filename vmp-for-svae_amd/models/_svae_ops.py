"""torch.autograd wrappers of the T2 / reconstruction HIP kernels (C ABI: include/vmp_hip.h).
No CPU fallback: tensors must be contiguous fp32 GPU tensors."""
import torch

from .. import _lib as L


def _c(t, name, shape=None):
    return L.dev_f32(t, name, shape)


class SvaeEStepFn(torch.autograd.Function):
    """(eta1, eta2d, hk, Pk, bias, noise, mk, Uk, kappa) -> (x (N,K,S,L), log_z (N,K), T' (N,K)).
    Gradients flow to eta1, eta2d (N,L) and hk, Pk, bias (K-sized, summed over n); theta-side inputs
    (mk, Uk, kappa) are treated as constants (reference svae.py:211-214 stop_gradient)."""

    @staticmethod
    def forward(ctx, eta1, eta2d, hk, Pk, bias, noise, mk, Uk, kappa):
        eta1 = _c(eta1, 'eta1')
        N, Ld = eta1.shape
        eta2d = _c(eta2d, 'eta2_diag', (N, Ld))
        K = hk.shape[0]
        hk, Pk, bias = _c(hk, 'eta1_phi2', (K, Ld)), _c(Pk, 'P_k', (K, Ld, Ld)), _c(bias, 'bias_k', (K,))
        noise = _c(noise, 'noise')
        if noise.dim() != 4 or tuple(noise.shape[:3]) != (N, K, Ld):
            raise L.VmpError('noise must have shape (N,K,L,S), got %s' % (tuple(noise.shape),))
        S = noise.shape[3]
        mk, Uk, kappa = _c(mk, 'm_k', (K, Ld)), _c(Uk, 'U_k', (K, Ld, Ld)), _c(kappa, 'kappa_k', (K,))
        f32 = dict(dtype=torch.float32, device=eta1.device)
        x = torch.empty(N, K, S, Ld, **f32)
        lz = torch.empty(N, K, **f32)
        Tp = torch.empty(N, K, **f32)
        L.check(L.lib().vmp_svae_estep_fwd(L.ptr(eta1), L.ptr(eta2d), L.ptr(hk), L.ptr(Pk), L.ptr(bias), L.ptr(noise),
                                           L.ptr(mk), L.ptr(Uk), L.ptr(kappa), N, K, Ld, S, L.ptr(x), L.ptr(lz),
                                           L.ptr(Tp), L.stream()), 'vmp_svae_estep_fwd')
        ctx.save_for_backward(eta1, eta2d, hk, Pk, bias, mk, Uk, x, lz)
        ctx.dims = (N, K, Ld, S)
        return x, lz, Tp

    @staticmethod
    def backward(ctx, g_x, g_lz, g_T):
        eta1, eta2d, hk, Pk, bias, mk, Uk, x, lz = ctx.saved_tensors
        N, K, Ld, S = ctx.dims
        f32 = dict(dtype=torch.float32, device=eta1.device)
        g_x = torch.zeros_like(x) if g_x is None else g_x.contiguous()
        g_lz = torch.zeros_like(lz) if g_lz is None else g_lz.contiguous()
        g_T = torch.zeros_like(lz) if g_T is None else g_T.contiguous()
        g_eta1 = torch.empty(N, Ld, **f32)
        g_eta2d = torch.empty(N, Ld, **f32)
        nblk = L.lib().vmp_svae_bwd_blocks(N, K)
        PW = L.lib().vmp_svae_bwd_partial_words(Ld)
        partials = torch.empty(nblk, K, PW, **f32)
        L.check(L.lib().vmp_svae_estep_bwd(L.ptr(eta1), L.ptr(eta2d), L.ptr(hk), L.ptr(Pk), L.ptr(bias), L.ptr(mk),
                                           L.ptr(Uk), L.ptr(x), L.ptr(lz), L.ptr(g_x), L.ptr(g_lz), L.ptr(g_T), N, K, Ld,
                                           S, L.ptr(g_eta1), L.ptr(g_eta2d), L.ptr(partials), partials.numel() * 4,
                                           L.stream()), 'vmp_svae_estep_bwd')
        red = partials.double().sum(0)                       # (K, PW): K-sized, fixed order
        g_hk = red[:, :Ld].float()
        tri = red[:, Ld:Ld + Ld * (Ld + 1) // 2]
        il = torch.tril_indices(Ld, Ld, device=eta1.device)
        g_P = torch.zeros(K, Ld, Ld, dtype=torch.float64, device=eta1.device)
        g_P[:, il[0], il[1]] = tri
        g_P = g_P + g_P.transpose(1, 2) - torch.diag_embed(torch.diagonal(g_P, dim1=1, dim2=2))
        g_bias = red[:, -1].float()
        return g_eta1, g_eta2d, g_hk, g_P.float(), g_bias, None, None, None, None


class DiagGaussLoglikeFn(torch.autograd.Function):
    """A_nk = sum_{s,d} (y - mean)^2 / var + log(var + 1e-8)   (reference vae.py:240), with gradients to mean, var."""

    @staticmethod
    def forward(ctx, y, mean, var):
        y = _c(y, 'y')
        mean = _c(mean, 'means')
        var = _c(var, 'vars', tuple(mean.shape))
        N, K, S, Dy = mean.shape
        if tuple(y.shape) != (N, Dy):
            raise L.VmpError('y must have shape (N,Dy)')
        A = torch.empty(N, K, dtype=torch.float32, device=y.device)
        L.check(L.lib().vmp_diag_gauss_loglike_fwd(L.ptr(y), L.ptr(mean), L.ptr(var), N, K, S, Dy, L.ptr(A), L.stream()),
                'vmp_diag_gauss_loglike_fwd')
        ctx.save_for_backward(y, mean, var)
        return A

    @staticmethod
    def backward(ctx, gA):
        y, mean, var = ctx.saved_tensors
        N, K, S, Dy = mean.shape
        gA = gA.contiguous()
        gm, gv = torch.empty_like(mean), torch.empty_like(var)
        L.check(L.lib().vmp_diag_gauss_loglike_bwd(L.ptr(y), L.ptr(mean), L.ptr(var), L.ptr(gA), N, K, S, Dy, L.ptr(gm),
                                                   L.ptr(gv), L.stream()), 'vmp_diag_gauss_loglike_bwd')
        return None, gm, gv


def gauss_logprob_nat(x, eta1, eta2, weights=None):
    raise NotImplementedError('stand-alone gaussian.log_probability_nat: fused into vmp_svae_estep_fwd '
                              '(use models.svae.e_step / compute_log_z_given_y)')


def gauss_logprob_per_samp(x_samps, eta1, eta2):
    raise NotImplementedError('stand-alone gaussian.log_probability_nat_per_samp: fused into vmp_svae_estep_fwd '
                              '(use models.svae.e_step(..., theta=theta) + compute_elbo)')


def student_t_logprob(y, mu, sigma, v):
    raise NotImplementedError('student_t.log_probability_per_samp: scheduled (SURVEY 8a row a8)')
