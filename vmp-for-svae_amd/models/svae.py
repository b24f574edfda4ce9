"""SVAE assembly - mirror of reference models/svae.py:14-516 (GMM-structured latent space).

N-sized arithmetic (per-(n,k) Cholesky, log-dets, triangular solves, responsibilities, reparameterised
samples, per-sample log-densities of the ELBO, and all of their gradients) runs in csrc/vmp_svae.hip; on the
training path the K-sized parameter maps (unpack_recognition_gmm + bias terms and their adjoint, the theta side of
compute_elbo, m_step + update_gmm_params) are the single-launch kernels of csrc/vmp_prep.hip; the stand-alone API
functions (unpack_recognition_gmm, compute_log_z_given_y, m_step, the SMM theta) remain torch with autograd.
"""
import math

import torch

from .. import _klinalg, _lib as L
from ..distributions import dirichlet, gaussian, niw
from . import _mix, _svae_ops, gmm, vae


def _tril_softplus(L_raw):
    Lt = torch.tril(L_raw)
    dg = torch.diagonal(Lt, dim1=-2, dim2=-1)
    return Lt - torch.diag_embed(dg) + torch.diag_embed(torch.nn.functional.softplus(dg, threshold=30.0))


def unpack_recognition_gmm(phi_gmm, name='unpack_phi2'):
    """reference svae.py:342-358: (eta1 = mu_k as-is, eta2 = -1/2 L L^T with softplus diagonal, pi = softmax)."""
    eta1, L_k_raw, pi_k_raw = phi_gmm
    Lk = _tril_softplus(L_k_raw)
    return eta1, -0.5 * (Lk @ Lk.transpose(-1, -2)), torch.softmax(pi_k_raw, dim=-1)


def unpack_recognition_gmm_debug(phi_gmm, name='unpack_phi2'):
    """reference svae.py:325-339 (used by experiments.py:424): as unpack_recognition_gmm, plus the raw factor."""
    eta1, eta2, pi_k = unpack_recognition_gmm(phi_gmm, name)
    return eta1, eta2, pi_k, phi_gmm[1]


def unpack_smm(theta_smm, name='unpack_theta_smm'):
    """reference svae.py:361-373."""
    mu, L_k_raw = theta_smm
    Lk = _tril_softplus(L_k_raw)
    return mu, Lk @ Lk.transpose(-1, -2)


def _recognition_bias(eta1_k, eta2_k, pi_k):
    """bias_k = B_k + log pi_k with B_k = -1/2 h_k^T P_k^-1 h_k + 1/2 log det P_k: the
    k-only part of log N(mu_n; mu_k, Sigma_n + Sigma_k) the reference builds in svae.py:70-92."""
    P = -2.0 * eta2_k
    Lc = _klinalg.cholesky(P)
    sol = torch.linalg.solve_triangular(Lc, eta1_k.unsqueeze(-1), upper=False).squeeze(-1)
    B = -0.5 * (sol * sol).sum(-1) + torch.log(torch.diagonal(Lc, dim1=-2, dim2=-1)).sum(-1)
    return P, B + torch.log(pi_k)


def _theta_pack(theta):
    """(m_k, W_k, kappa_k, nu_k) such that log p(x, z=k | theta) = kappa_k - 1/2 f(|W_k (x - m_k)|^2), W lower.
    GMM theta = natural NIW / Dirichlet 5-tuple (reference svae.py:205-214, stop_gradient): f = identity,
    W = chol(E[Sigma])^-1.  SMM theta = (alpha_nat, mu_k, L_k_raw, DoF) (svae.py:268-277): Student-t,
    f(d) = (nu+L) log1p(d/nu), W = L_k^-1, gradients flow to mu_k and L_k."""
    if len(theta) == 4:
        alpha_nat, mu_k, L_raw, dof = theta
        Ld = mu_k.shape[-1]
        Lk = _tril_softplus(L_raw)
        eye = torch.eye(Ld, dtype=Lk.dtype, device=Lk.device).expand_as(Lk)
        W = torch.linalg.solve_triangular(Lk, eye, upper=False)
        elp = dirichlet.expected_log_pi(dirichlet.natural_to_standard(alpha_nat)).detach()
        nu = dof.detach().float()
        kappa = (torch.lgamma(0.5 * (nu + Ld)) - torch.lgamma(0.5 * nu) - 0.5 * Ld * torch.log(math.pi * nu)
                 - torch.log(torch.diagonal(Lk, dim1=-2, dim2=-1)).sum(-1) + elp)
        return mu_k, W, kappa, nu.contiguous()
    m, W, kappa = _svae_ops.theta_pack_gmm(theta)           # one launch (csrc/vmp_prep.hip)
    return m, W, kappa, None


def _neutral_theta(K, Ld, device):
    f32 = dict(dtype=torch.float32, device=device)
    return torch.zeros(K, Ld, **f32), torch.zeros(K, Ld, Ld, **f32), torch.zeros(K, **f32), None


class PhiTilde(object):
    """What e_step returns as `phi_tilde`: behaves like the reference's tuple (eta1 (N,K,L,1), eta2 (N,K,L,L))
    - materialised only if indexed - and carries the fused per-cell ELBO terms for compute_elbo.  When e_step was not
    given theta (the reference's own call order: inference(...) and THEN compute_elbo(..., theta, phi_tilde, ...),
    experiments.py:209-229), theta_term() evaluates them afterwards from what e_step already had."""

    def __init__(self, eta1_phi1, eta2_diag, eta1_phi2, P_phi2, T_prime, theta_key, bias=None, noise=None, x=None):
        self._p = (eta1_phi1, eta2_diag, eta1_phi2, P_phi2)
        self.T_prime = T_prime
        self.theta_key = theta_key
        self._bias, self._noise, self._x = bias, noise, x
        # what the E-step kernel's epilogue already produced (in-kernel noise; None otherwise): the one-draw sub-sample
        # x_samples (N,L) with the key it was drawn with, r = exp(log_z), per-block partials of the M-step moments
        self.x_samples = self.r_nk = self.mom = self.draw_key = None

    def theta_term(self, theta, x_k_samps):
        """T'_nk = mean_s[log N(x_nks; phi~_nk) - log p(x_nks, z=k | theta)] (reference svae.py:229-243) for a theta the
        E-step did not see: one more launch of the fused forward kernel on the SAME (phi_enc, phi_gmm, noise) - it
        reproduces the samples bit for bit - with this theta's (m_k, W_k, kappa_k); autograd reaches phi_enc / phi_gmm
        (and a Student-t theta) through the fused backward kernel.  Backward is linear in the upstream gradients, so the
        gradients of this call (upstream dT') and of the original e_step call (upstream dx, dlog_z) add up to exactly
        what the one-pass form e_step(theta=theta) delivers."""
        own_x = self._x is not None and (x_k_samps is self._x or (torch.is_tensor(x_k_samps) and x_k_samps.data_ptr() == self._x.data_ptr()
                                                                  and tuple(x_k_samps.shape) == tuple(self._x.shape)))
        if self._noise is None or not own_x:
            # other samples than the E-step's own (or a phi_tilde built for another theta): the reference's literal
            # formulation (svae.py:229-243 / 288-300) on the stand-alone, differentiable density kernels
            return _theta_term_literal(theta, self, x_k_samps)
        e1, e2d, e1k, Pk = self._p
        mk, Wk, kap, nu = _theta_pack(theta)
        _, _, Tp = _svae_ops.SvaeEStepFn.apply(e1, e2d, e1k, Pk, self._bias, self._noise, mk, Wk, kap, nu)
        self.T_prime, self.theta_key = Tp, _theta_key(theta)
        return Tp

    def __len__(self):
        return 2

    def __getitem__(self, i):
        e1, e2d, e1k, Pk = self._p
        e2k = -0.5 * Pk
        if i == 0:
            return (e1.unsqueeze(1) + e1k.unsqueeze(0)).unsqueeze(-1)
        if i == 1:
            return torch.diag_embed(e2d).unsqueeze(1) + e2k.unsqueeze(0)
        raise IndexError(i)

    def __iter__(self):
        return iter((self[0], self[1]))


def _theta_term_literal(theta, phi_tilde, x_k_samps):
    """mean_s[ log N(x_s; phi~_nk) - log p(x_s, z=k | theta) ] exactly as the reference writes it (svae.py:229-243; SMM:
    288-300): gaussian.log_probability_nat_per_samp on the materialised phi_tilde, and the same function on the tiled
    E[theta] (stop_gradient, svae.py:211-214) resp. student_t.log_probability_per_samp - stand-alone HIP kernels with
    backward kernels (csrc/vmp_density.hip), so gradients reach x_k_samps, phi_tilde and a Student-t theta."""
    from ..distributions import student_t
    N, K, S, Ld = x_k_samps.shape
    x = x_k_samps.contiguous()
    num = gaussian.log_probability_nat_per_samp(x, phi_tilde[0].squeeze(-1).contiguous(), phi_tilde[1].contiguous())
    if len(theta) == 4:
        alpha_nat, mu_k, L_raw, dof = theta
        mu, sigma = unpack_smm((mu_k, L_raw))
        den = student_t.log_probability_per_samp(x, mu, sigma, dof)
        elp = dirichlet.expected_log_pi(dirichlet.natural_to_standard(alpha_nat)).detach()
    else:
        beta_k, m_k, C_k, v_k = niw.natural_to_standard(*theta[1:])
        mu, sigma = niw.expected_values((beta_k, m_k, C_k, v_k))
        e1t, e2t = gaussian.standard_to_natural(mu, sigma)
        e1t, e2t = e1t.detach().float(), e2t.detach().float()
        den = gaussian.log_probability_nat_per_samp(x, e1t.unsqueeze(0).expand(N, K, Ld).contiguous(),
                                                    e2t.unsqueeze(0).expand(N, K, Ld, Ld).contiguous())
        elp = dirichlet.expected_log_pi(dirichlet.natural_to_standard(theta[0])).detach()
    return (num - den - elp.view(1, K, 1)).mean(-1)


def _theta_key(theta):
    return tuple((t.data_ptr(), t._version) for t in theta) if theta is not None else None


def compute_log_z_given_y(eta1_phi1, eta2_phi1, eta1_phi2, eta2_phi2, pi_phi2, name='log_q_z_given_y_phi'):
    """reference svae.py:50-92.  eta2_phi1 is the (N,L,L) DIAGONAL matrix the reference passes (svae.py:29).
    Returns (log q(z|y), (w_eta1, w_eta2)) - the debug pair is not produced by the fused kernel (None)."""
    N, Ld = eta1_phi1.shape
    K = eta1_phi2.shape[0]
    e2d = torch.diagonal(eta2_phi1, dim1=-2, dim2=-1).contiguous()
    P, bias = _recognition_bias(eta1_phi2, eta2_phi2, pi_phi2)
    noise = torch.zeros(N, K, Ld, 1, dtype=torch.float32, device=eta1_phi1.device)
    mk, Wk, kap, nu = _neutral_theta(K, Ld, eta1_phi1.device)
    _, lz, _ = _svae_ops.SvaeEStepFn.apply(eta1_phi1, e2d, eta1_phi2.contiguous(), P.contiguous(), bias, noise, mk, Wk, kap, nu)
    return lz, (None, None)


def recognition_prep(phi_gmm, theta=None):
    """unpack_recognition_gmm + the k-only part of compute_log_z_given_y in one launch (autograd: one more); a natural
    GMM theta is packed by the same launch (reference svae.py:342-358, 70-92)"""
    gmm_theta = theta is not None and len(theta) == 5
    return _svae_ops.PhiPrepFn.apply(*phi_gmm, *([t.detach() for t in theta] if gmm_theta else []))


def e_step(phi_enc, phi_gmm, nb_samples, seed=0, name="e_step", noise=None, theta=None, prep=None):
    """reference svae.py:14-47.  Returns (x_k_samples (N,K,S,L), log_z (N,K), phi_tilde, dbg).
    `noise` (N,K,L,S) replaces tf.random_normal (default: torch.randn with `seed`); noise='philox' draws eps INSIDE the
    kernel (Philox4x32-7 keyed by `seed`, as the reference's tf.random_normal does inside its step, svae.py:113-114):
    no (N,K,L,S) tensor is written or read.  When `theta` (natural NIW / Dirichlet parameters) is given, the per-sample
    densities compute_elbo needs are evaluated in the same pass."""
    eta1_phi1, eta2_diag = phi_enc
    N, Ld = eta1_phi1.shape
    # (`prep`: recognition_prep(phi_gmm, theta) already launched by the caller - a graph-captured step runs it beside the encoder)
    gmm_theta = theta is not None and len(theta) == 5
    if prep is None:
        prep = recognition_prep(phi_gmm, theta)
    eta1_phi2, P, bias = prep[:3]
    K = eta1_phi2.shape[0]
    if isinstance(noise, str):
        if noise != 'philox':
            raise ValueError("noise must be a tensor, None or 'philox'")
        noise = _svae_ops.PhiloxNoise(seed, nb_samples, epilogue=True)
    elif isinstance(noise, _svae_ops.PhiloxNoise):
        pass                                                 # caller-built (e.g. with the key in a device word)
    elif noise is None:
        g = torch.Generator(device=eta1_phi1.device).manual_seed(int(seed))
        noise = torch.randn(N, K, Ld, nb_samples, generator=g, device=eta1_phi1.device)
    if gmm_theta:
        mk, Wk, kap, nu = prep[3], prep[4], prep[5], None
    else:
        mk, Wk, kap, nu = _theta_pack(theta) if theta is not None else _neutral_theta(K, Ld, eta1_phi1.device)
    x, lz, Tp = _svae_ops.SvaeEStepFn.apply(eta1_phi1, eta2_diag, eta1_phi2, P, bias, noise, mk, Wk, kap, nu)
    # without theta the (phi, noise) the E-step ran on stay attached, so that compute_elbo can evaluate the theta term
    # afterwards; with theta (the training path) nothing extra is kept alive
    keep = dict(bias=bias, noise=noise, x=x) if theta is None else {}
    phi_tilde = PhiTilde(eta1_phi1, eta2_diag, eta1_phi2, P, Tp if theta is not None else None, _theta_key(theta), **keep)
    if isinstance(noise, _svae_ops.PhiloxNoise) and noise.x_samples is not None:
        phi_tilde.x_samples, phi_tilde.r_nk, phi_tilde.mom = noise.x_samples, noise.r_nk, noise.mom
        phi_tilde.draw_key = (noise.seed, noise.seed_dev)
        noise.x_samples = noise.r_nk = noise.mom = None      # (a caller-built object may be reused: do not pin the tensors)
    return x, lz, phi_tilde, (None, None)


def sample_x_per_comp(eta1, eta2, nb_samples, seed=0, noise=None):
    """reference svae.py:95-119 as a stand-alone function for GENERAL (N,K,L,L) eta2 (API parity; torch batched
    factorisations): x = Sigma eta1 + L^-T eps, L = chol(-2 eta2), eps (N,K,L,S); returns (N,K,S,L).  The training path
    never calls it: e_step's fused kernel does this per cell with the structured eta2 = diag(encoder) + P_k."""
    N, K, Ld, _ = eta2.shape
    inv_sigma = -2.0 * eta2
    Lc = torch.linalg.cholesky(inv_sigma)
    if noise is None:
        g = torch.Generator(device=eta2.device).manual_seed(int(seed))
        noise = torch.randn(N, K, Ld, nb_samples, generator=g, device=eta2.device, dtype=eta2.dtype)
    nz = torch.linalg.solve_triangular(Lc.transpose(-1, -2), noise, upper=True)
    mu = torch.cholesky_solve(eta1, Lc)
    return (mu + nz).transpose(-1, -2)


def subsample_x(x_k_samples, log_q_z_given_y, seed=0, z_draws=None, nb_out=None, u=None, return_z=False):
    """reference svae.py:122-151: z_ns ~ Cat(exp log_q), gather x[n, z_ns, s].  HIP kernel vmp_svae_subsample;
    `z_draws` (N,S) replaces tf.multinomial (default: inverse CDF of torch.rand with `seed`; u='philox': of uniforms
    drawn inside the kernel from Philox4x32-7 keyed by `seed`).  `nb_out` < S only
    produces the first nb_out sample columns (the reference's caller keeps s = 0, svae.py:514).  return_z: also the drawn
    component indices (N, nb_out) int64 (the reference's z_samps)."""
    x = L.dev_f32(x_k_samples.detach(), 'x_k_samples')
    N, K, S, Ld = x.shape
    lz = L.dev_f32(log_q_z_given_y.detach(), 'log_q_z_given_y', (N, K))
    So = S if nb_out is None else int(nb_out)
    z = None
    if z_draws is None and (isinstance(u, str) or isinstance(u, _svae_ops.PhiloxNoise)):
        # uniforms drawn inside the kernel (Philox4x32-7 keyed by `seed`, or by the device word of a PhiloxNoise)
        if isinstance(u, str) and u != 'philox':
            raise ValueError("u must be a tensor, None or 'philox'")
        sd = u.seed_dev if isinstance(u, _svae_ops.PhiloxNoise) else None
        key = (u.seed if isinstance(u, _svae_ops.PhiloxNoise) else int(seed)) & 0xFFFFFFFFFFFFFFFF
        out = torch.empty(N, So, Ld, dtype=torch.float32, device=x.device)
        zo = torch.empty(N, So, dtype=torch.int64, device=x.device) if return_z else None
        L.check(L.lib().vmp_svae_subsample_rng(L.ptr(x), L.ptr(lz), key, L.ptr(sd), N, K, S, Ld, So, L.ptr(out), L.ptr(zo), L.stream()),
                'vmp_svae_subsample_rng')
        return (out, zo) if return_z else out
    if z_draws is not None:
        u = None
        z = z_draws[:, :So].to(torch.int64).contiguous()
    elif u is not None:                                  # supplied uniforms (N,So) - e.g. a static graph input
        u = L.dev_f32(u, 'u', (N, So))
    else:
        g = torch.Generator(device=x.device).manual_seed(int(seed))
        u = torch.rand(N, So, generator=g, device=x.device)
    out = torch.empty(N, So, Ld, dtype=torch.float32, device=x.device)
    zo = torch.empty(N, So, dtype=torch.int64, device=x.device) if return_z else None
    L.check(L.lib().vmp_svae_subsample(L.ptr(x), L.ptr(lz), L.ptr(u), L.ptr(z), N, K, S, Ld, So, L.ptr(out), L.ptr(zo),
                                       L.stream()), 'vmp_svae_subsample')
    return (out, zo) if return_z else out


def m_step(gmm_prior, x_samples, r_nk):
    """reference svae.py:154-176: theta* (natural) from the GMM M-step on (x_samples (N,L), r_nk)."""
    beta_0, m_0, C_0, v_0 = niw.natural_to_standard(*gmm_prior[1:])
    alpha_0 = dirichlet.natural_to_standard(gmm_prior[0])
    alpha_k, beta_k, m_k, C_k, v_k, _, _ = gmm.m_step(x_samples.detach().contiguous(), r_nk.detach().contiguous(),
                                                       alpha_0, beta_0, m_0, C_0, v_0, name='gmm_m_step')
    A, b, beta, v_hat = niw.standard_to_natural(beta_k, m_k, C_k, v_k)
    return [dirichlet.standard_to_natural(alpha_k), A, b, beta, v_hat]


def m_step_from_stats(gmm_prior, stats):
    """theta* directly in natural parameters from raw moments [N_k | W_k | sum r x | sum r x x^T] (SURVEY appendix
    A.6) - the form the data-parallel driver uses after all-reducing the moments."""
    K = stats.shape[0]
    Ld = gmm_prior[2].shape[1]
    Nk = stats[:, 0].float()
    sx = stats[:, 2:2 + Ld].float()
    sxx = stats[:, 2 + Ld:].reshape(K, Ld, Ld).float()
    alpha, A, b, beta, v_hat = gmm_prior
    return [alpha + Nk, A + sxx, b + sx, beta + Nk, v_hat + Nk + 1.0]


def cvi_update_from_stats(gmm_prior, theta, stats, step_size, want_star=True, step_size_dev=None):
    """m_step_from_stats + update_gmm_params (reference svae.py:154-176, 376-403) in one launch; theta in place."""
    return _svae_ops.cvi_update(gmm_prior, theta, stats, step_size, want_star, step_size_dev)


def m_step_smm(smm_prior, r_nk):
    """reference svae.py:179-196: only the Dirichlet parameter is updated."""
    alpha_0 = dirichlet.natural_to_standard(smm_prior[0] if isinstance(smm_prior, (list, tuple)) else smm_prior)
    N_k = gmm.update_Nk(r_nk.detach().contiguous())
    return dirichlet.standard_to_natural(alpha_0 + N_k)


def compute_elbo(y, reconstructions, theta, phi_tilde, x_k_samps, log_z_given_y_phi, decoder_type, grad_seed=None):
    """reference svae.py:199-262.  Returns (elbo, (neg_rec_err, numerator, denominator, regulariser)).
    grad_seed (_svae_ops.GradSeed, optional): the upstream gradient the caller will differentiate elbo with (the trainer:
    -1, loss = -elbo) - lets the fused path skip every rescaling launch."""
    if decoder_type not in ('standard', 'bernoulli'):
        raise NotImplementedError("decoder_type '%s'" % decoder_type)
    if not isinstance(phi_tilde, PhiTilde):
        raise L.VmpError('compute_elbo: phi_tilde must be the object e_step / inference returned')
    if decoder_type == 'standard' and isinstance(reconstructions, vae.LazyReconstruction):
        # fused decoder: value, the three scalars and every gradient seed from three launches (_svae_ops.FusedElboFn)
        if phi_tilde.T_prime is not None and phi_tilde.theta_key == _theta_key(theta):
            Tp = phi_tilde.T_prime
        else:
            Tp = phi_tilde.theta_term(theta, x_k_samps)
        seed_t = None if grad_seed is None else grad_seed.tensor
        sigma = 1.0 if grad_seed is None else grad_seed.value
        elbo, rec, reg, r_nk = _svae_ops.FusedElboFn.apply(y, reconstructions.x, log_z_given_y_phi, Tp, seed_t, sigma,
                                                          *reconstructions.params)
        details = ElboDetails(rec, reg, phi_tilde, x_k_samps, log_z_given_y_phi)
        details.r_nk = r_nk
        details.mom = phi_tilde.mom
        return elbo, details
    r_nk = torch.exp(log_z_given_y_phi)
    if decoder_type == 'bernoulli':                               # svae.py:222-223: out_2 = logits
        rec = vae.expected_bernoulli_loglike(y, reconstructions[1], r_nk=r_nk)
    elif isinstance(reconstructions, vae.LazyReconstruction):   # fused decoder + reconstruction term
        rec = vae.expected_diagonal_gaussian_loglike(y, reconstructions, None, weights=r_nk)
    else:
        means, out_2 = reconstructions
        rec = vae.expected_diagonal_gaussian_loglike(y, means, out_2, weights=r_nk)
    if phi_tilde.T_prime is not None and phi_tilde.theta_key == _theta_key(theta):
        Tp = phi_tilde.T_prime                                    # e_step(theta=theta) evaluated them in its own pass
    else:
        Tp = phi_tilde.theta_term(theta, x_k_samps)               # the reference's call order (experiments.py:209-229)
    reg = (r_nk * (Tp + log_z_given_y_phi)).sum()
    elbo = rec - reg
    return elbo, ElboDetails(rec, reg, phi_tilde, x_k_samps, log_z_given_y_phi)


class ElboDetails(object):
    """The `details` 4-tuple of reference svae.py:256-260 (neg_rec_err, sum r mean_s numerator, sum r mean_s denominator,
    regulariser).  In the reference these are graph nodes evaluated only when fetched; here the two middle debug scalars
    - which the fused kernel does not keep apart (it returns mean_s[numerator - denominator]) - are computed on first
    access: numerator from the stand-alone density kernel on the materialised phi_tilde (gaussian.py:74-105),
    denominator = numerator - regulariser."""

    def __init__(self, rec, reg, phi_tilde, x_k, log_z):
        self._rec, self._reg = rec, reg
        self.r_nk = None            # exp(log_z) when the fused tail produced it (the M-step reuses it)
        self.mom = None             # per-block partials of the M-step moments when the E-step kernel's epilogue produced them
        self._lazy = (phi_tilde, x_k, log_z)
        self._nd = None

    def _split(self):
        if self._nd is None:
            phi_tilde, x_k, log_z = self._lazy
            with torch.no_grad():
                lp = gaussian.log_probability_nat_per_samp(x_k.detach().contiguous(), phi_tilde[0].squeeze(-1).detach().contiguous(),
                                                           phi_tilde[1].detach().contiguous())
                num = (torch.exp(log_z) * (lp.mean(-1) + log_z)).sum().detach()
                self._nd = (num, num - self._reg.detach())
        return self._nd

    def __len__(self):
        return 4

    def __getitem__(self, i):
        if i in (0, -4):
            return self._rec
        if i in (3, -1):
            return self._reg
        if i in (1, -3):
            return self._split()[0]
        if i in (2, -2):
            return self._split()[1]
        raise IndexError(i)

    def __iter__(self):
        return iter((self[0], self[1], self[2], self[3]))


def compute_elbo_smm(y, reconstructions, theta, phi_tilde, x_k_samps, log_z_given_y_phi, decoder_type, grad_seed=None):
    """reference svae.py:265-322: as compute_elbo with the Student-t density of theta = (alpha, mu_k, L_k, DoF)
    (distributions/student_t.py:7-39); the per-sample densities come from the fused E-step (e_step(theta=theta))."""
    return compute_elbo(y, reconstructions, theta, phi_tilde, x_k_samps, log_z_given_y_phi, decoder_type, grad_seed=grad_seed)


def update_gmm_params(current_gmm_params, gmm_params_star, step_size, name='cvi_update_theta'):
    """reference svae.py:376-403: theta <- (1 - rho) theta + rho theta*, in place."""
    with torch.no_grad():
        for cur, star in zip(current_gmm_params, gmm_params_star):
            cur.mul_(1.0 - step_size).add_(star.to(cur.dtype), alpha=float(step_size))
    return current_gmm_params


def init_mm_params(nb_components, latent_dims, alpha_scale=.1, beta_scale=1e-5, v_init=10., m_scale=1., C_scale=10.,
                   seed=0, as_variables=True, trainable=False, device='cuda', name='gmm', m_uniform=None):
    """reference svae.py:433-458.  `m_uniform` in [0,1) (K,L) replaces tf.random_uniform."""
    f32 = dict(dtype=torch.float32, device=device)
    K, Ld = nb_components, latent_dims
    alpha = alpha_scale * torch.ones(K, **f32)
    beta = beta_scale * torch.ones(K, **f32)
    v = torch.full((K,), float(Ld + v_init), **f32)
    if m_uniform is None:
        g = torch.Generator(device='cpu').manual_seed(int(seed))
        m_uniform = torch.rand(K, Ld, generator=g)
    m = m_scale * (m_uniform.to(**f32) * 2.0 - 1.0)
    C = C_scale * torch.eye(Ld, **f32).expand(K, Ld, Ld).contiguous()
    A, b, beta, v_hat = niw.standard_to_natural(beta, m, C, v)
    return dirichlet.standard_to_natural(alpha), A, b, beta, v_hat


def init_mm(nb_components, latent_dims, seed=0, param_device='cuda', name='init_mm', theta_as_variable=True,
            m_uniform=None):
    """reference svae.py:461-471."""
    prior = init_mm_params(nb_components, latent_dims, alpha_scale=0.05 / nb_components, beta_scale=0.5, m_scale=0,
                           C_scale=latent_dims + 0.5, v_init=latent_dims + 0.5, seed=seed, device=param_device,
                           m_uniform=m_uniform)
    theta = init_mm_params(nb_components, latent_dims, alpha_scale=1., beta_scale=1., m_scale=5.,
                           C_scale=2 * latent_dims, v_init=latent_dims + 1., seed=seed, device=param_device,
                           m_uniform=m_uniform)
    return prior, [t.clone() for t in theta]


def make_loc_scale_variables(theta, param_device='cuda', name='copy_m_v'):
    """reference svae.py:474-485."""
    # K-sized, once per model: the factorisations run on the host (_klinalg: torch's GPU Cholesky returned a wrong factor on its
    # first call after HIP-graph replays in a process that shares its GPU with another rank - rounds 5 / 6, tools/r6_dpg_repro.py)
    std = niw.natural_to_standard(theta[1].detach(), theta[2].detach(), theta[3].detach(), theta[4].detach())
    mu, sigma = niw.expected_values(std)
    # contiguous copies: cholesky returns a column-major batch, and one oddly-strided parameter drops the optimiser's
    # multi-tensor updates onto the per-tensor slow path
    dev = theta[1].device
    return (torch.nn.Parameter(mu.clone(memory_format=torch.contiguous_format).to(dev)),
            torch.nn.Parameter(_klinalg.cholesky(sigma).clone(memory_format=torch.contiguous_format).to(dev)))


def init_recognition_params(theta, nb_components, seed=0, param_device='cuda', var_scope='phi_gmm', pi_normal=None):
    """reference svae.py:488-496.  `pi_normal` (K,) replaces tf.random_normal."""
    if pi_normal is None:
        g = torch.Generator(device='cpu').manual_seed(int(seed))
        pi_normal = torch.randn(nb_components, generator=g)
    mu_k, L_k = make_loc_scale_variables(theta, param_device)
    pi_k = torch.nn.Parameter(torch.softmax(pi_normal.to(mu_k.device, torch.float32), dim=-1))
    return mu_k, L_k, pi_k


def inference(y, phi_gmm, encoder_layers, decoder_layers, nb_samples=10, stddev_init_nn=0.01, seed=0, name='inference',
              param_device='cuda', noise=None, z_draws=None, theta=None, lazy_decoder=False, u=None, prep=None):
    """reference svae.py:499-516.  Returns (y_reconstruction, x_given_y_phi, x_k_samples, x_samples, log_z, phi_gmm,
    phi_tilde).  lazy_decoder=True: y_reconstruction is a vae.LazyReconstruction (fused decoder kernels)."""
    x_given_y_phi = vae.make_encoder(y, layerspecs=encoder_layers, stddev_init=stddev_init_nn, seed=seed)
    x_k_samples, log_z, phi_tilde, _ = e_step(x_given_y_phi, phi_gmm, nb_samples, seed=seed, noise=noise, theta=theta, prep=prep)
    y_rec = vae.make_decoder(x_k_samples, layerspecs=decoder_layers, stddev_init=stddev_init_nn, seed=seed,
                             lazy=lazy_decoder)
    own_draw = u is None and z_draws is None and (isinstance(noise, str) or isinstance(noise, _svae_ops.PhiloxNoise))
    if own_draw:
        u = noise                                            # in-kernel noise: the draw's uniforms come from the same generator
    key = None if not own_draw else ((int(seed) & 0xFFFFFFFFFFFFFFFF, None) if isinstance(noise, str) else (noise.seed, noise.seed_dev))
    dk = phi_tilde.draw_key
    if own_draw and phi_tilde.x_samples is not None and dk is not None and dk[0] == key[0] and dk[1] is key[1]:
        x_samples = phi_tilde.x_samples                      # drawn in the E-step kernel's epilogue: same uniforms, same arithmetic
    else:
        x_samples = subsample_x(x_k_samples, log_z, seed, z_draws=z_draws, nb_out=1, u=u)[:, 0, :]
    return y_rec, x_given_y_phi, x_k_samples, x_samples, log_z, phi_gmm, phi_tilde


def predict(y, phi_gmm, encoder_layers, decoder_layers, seed=0, noise=None, z_draws=None):
    """reference svae.py:406-430: encode, E-step with ONE sample per component, draw z ~ q(z|y), decode the drawn
    sample; returns (y_mean, argmax_k log r_nk).  `noise` (N,K,L,1) / `z_draws` (N,1) replace the two TF draws."""
    phi_enc = vae.make_encoder(y, layerspecs=encoder_layers)
    x_k_samples, log_r_nk, _, _ = e_step(phi_enc, phi_gmm, 1, seed=seed, noise=noise)
    x_samples = subsample_x(x_k_samples, log_r_nk, seed, z_draws=z_draws)[:, 0, :]
    y_mean, _ = vae.make_decoder(x_samples, layerspecs=decoder_layers)
    return y_mean, torch.argmax(log_r_nk, dim=1)


def identity_transform(input, nb_components, nb_samples, type='standard', name='debug_nn'):
    """reference svae.py:519-535 (debug helper that freezes the network): mu_n = x_n, Sigma_n = 0.1 I (2-D data)."""
    nn_var = 1e-1
    mu = input
    sigma = nn_var * torch.eye(2, dtype=input.dtype, device=input.device).expand(mu.shape[0], 2, 2)
    if type == 'natparam':
        eta1, eta2 = gaussian.standard_to_natural(mu, sigma)
        return eta1, torch.diagonal(eta2, dim1=-2, dim2=-1)
    sig = torch.full_like(mu[:, :2], nn_var)              # diagonal of sigma, (N,2)
    if tuple(sig.shape) != tuple(input.shape):            # svae.py:532-533
        sig = sig[:, None, None, :].expand(-1, nb_components, nb_samples, -1)
    return mu, sig
