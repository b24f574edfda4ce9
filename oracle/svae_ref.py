"""Oracle: SVAE assembly (reference models/svae.py:14-516), literal formulation.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Random draws (noise, categorical draws) are arguments.
"""
import torch

from . import dists, mixtures, nets


def unpack_recognition_gmm(phi_gmm):
    """svae.py:342-358: eta1 = mu_k variable as-is; L = tril(raw) with softplus diagonal; P = L L^T;
    eta2 = -P/2; pi = softmax(raw)."""
    eta1, L_raw, pi_raw = phi_gmm
    L = torch.tril(L_raw)
    dg = torch.diagonal(L, dim1=-2, dim2=-1)
    L = L - torch.diag_embed(dg) + torch.diag_embed(nets.softplus(dg))
    P = L @ L.transpose(-1, -2)
    return eta1, -0.5 * P, torch.softmax(pi_raw, dim=-1)


def unpack_smm(theta_smm):
    """svae.py:361-373: (mu, Sigma = L L^T) with softplus diagonal."""
    mu, L_raw = theta_smm
    L = torch.tril(L_raw)
    dg = torch.diagonal(L, dim1=-2, dim2=-1)
    L = L - torch.diag_embed(dg) + torch.diag_embed(nets.softplus(dg))
    return mu, L @ L.transpose(-1, -2)


def compute_log_z_given_y(eta1_phi1, eta2_phi1, eta1_phi2, eta2_phi2, pi_phi2):
    """svae.py:50-92 (two LU solves per (n,k) cell, symmetrisation, then gaussian.log_probability_nat)."""
    N, L = eta1_phi1.shape
    K = eta1_phi2.shape[0]
    eta2_tilde = eta2_phi1.unsqueeze(1) + eta2_phi2.unsqueeze(0)
    solved = dists.solve(eta2_tilde, eta2_phi2.unsqueeze(0).repeat(N, 1, 1, 1))
    w_eta2 = torch.einsum('nju,nkui->nkij', eta2_phi1, solved)
    w_eta2 = (w_eta2 + w_eta2.transpose(-1, -2)) / 2.
    rhs = eta1_phi2.unsqueeze(0).unsqueeze(-1).repeat(N, 1, 1, 1)
    w_eta1 = torch.einsum('nuj,nkuv->nkj', eta2_phi1, dists.solve(eta2_tilde, rhs))
    mu_phi1, _ = dists.gauss_natural_to_standard(eta1_phi1, eta2_phi1)
    return dists.gauss_log_probability_nat(mu_phi1, w_eta1, w_eta2, pi_phi2), (w_eta1, w_eta2)


def sample_x_per_comp(eta1, eta2, noise):
    """svae.py:95-119; `noise` (N,K,L,S) replaces tf.random_normal.  Output (N,K,S,L)."""
    inv_sigma = -2 * eta2
    Lc = dists.chol(inv_sigma)
    nz = dists.solve(Lc.transpose(-1, -2), noise)
    return (dists.solve(inv_sigma, eta1) + nz).permute(0, 1, 3, 2)


def e_step(phi_enc, phi_gmm, noise):
    """svae.py:14-47.  Returns (x_k_samples (N,K,S,L), log_z (N,K), phi_tilde, (w_eta1, w_eta2))."""
    eta1_phi1, eta2_diag = phi_enc
    eta2_phi1 = torch.diag_embed(eta2_diag)
    eta1_phi2, eta2_phi2, pi_phi2 = unpack_recognition_gmm(phi_gmm)
    log_z, dbg = compute_log_z_given_y(eta1_phi1, eta2_phi1, eta1_phi2, eta2_phi2, pi_phi2)
    eta1_t = (eta1_phi1.unsqueeze(1) + eta1_phi2.unsqueeze(0)).unsqueeze(-1)
    eta2_t = eta2_phi1.unsqueeze(1) + eta2_phi2.unsqueeze(0)
    x_k = sample_x_per_comp(eta1_t, eta2_t, noise)
    return x_k, log_z, (eta1_t, eta2_t), dbg


def subsample_x(x_k_samples, z_draws):
    """svae.py:122-151; `z_draws` (N,S) int replaces tf.multinomial.  Output (N,S,L)."""
    N, K, S, L = x_k_samples.shape
    n_idx = torch.arange(N).view(-1, 1).expand(N, S)
    s_idx = torch.arange(S).view(1, -1).expand(N, S)
    return x_k_samples[n_idx, z_draws.long(), s_idx]


def m_step(gmm_prior, x_samples, r_nk):
    """svae.py:154-176: natural prior -> standard, gmm.m_step, -> natural theta*."""
    beta_0, m_0, C_0, v_0 = dists.niw_natural_to_standard(*gmm_prior[1:])
    alpha_0 = dists.dir_natural_to_standard(gmm_prior[0])
    alpha_k, beta_k, m_k, C_k, v_k, _, _ = mixtures.gmm_m_step(x_samples, r_nk, alpha_0, beta_0, m_0, C_0, v_0)
    A, b, beta, v_hat = dists.niw_standard_to_natural(beta_k, m_k, C_k, v_k)
    return [dists.dir_standard_to_natural(alpha_k), A, b, beta, v_hat]


def m_step_smm(alpha_prior_nat, r_nk):
    """svae.py:179-196: only the Dirichlet parameter is updated."""
    return dists.dir_standard_to_natural(dists.dir_natural_to_standard(alpha_prior_nat) + r_nk.sum(0))


def _regulariser(r_nk, log_num, log_den):
    """svae.py:245-260 (shared by both ELBOs)."""
    reg = (r_nk.unsqueeze(2) * (log_num - log_den)).sum(1).sum(0).mean()
    d1 = (r_nk * log_num.mean(-1)).sum()
    d2 = (r_nk * log_den.mean(-1)).sum()
    return reg, d1, d2


def _reconstruction(y, reconstructions, r_nk):
    """the decoder term of both ELBOs (vae.py:233-248); reconstructions=None leaves it out (the T2 unit of SURVEY 8d: VMP
    step without the MLPs and without the reconstruction term)."""
    if reconstructions is None:
        return r_nk.new_zeros(())
    means, var = reconstructions
    return nets.expected_diagonal_gaussian_loglike(y, means, var, weights=r_nk)


def compute_elbo(y, reconstructions, theta, phi_tilde, x_k, log_z):
    """svae.py:199-262 (Gaussian decoder).  Returns (elbo, (rec, num, den, reg))."""
    beta_k, m_k, C_k, v_k = dists.niw_natural_to_standard(*theta[1:])
    mu, sigma = dists.niw_expected_values(beta_k, m_k, C_k, v_k)
    eta1_th, eta2_th = dists.gauss_standard_to_natural(mu, sigma)
    elp = dists.dir_expected_log_pi(dists.dir_natural_to_standard(theta[0]))
    eta1_th, eta2_th, elp = eta1_th.detach(), eta2_th.detach(), elp.detach()
    r_nk = torch.exp(log_z)
    rec = _reconstruction(y, reconstructions, r_nk)
    eta1_t, eta2_t = phi_tilde
    N, K, L, _ = eta2_t.shape
    log_num = dists.gauss_log_probability_nat_per_samp(x_k, eta1_t.reshape(N, K, L), eta2_t) + log_z.unsqueeze(2)
    log_den = dists.gauss_log_probability_nat_per_samp(x_k, eta1_th.unsqueeze(0).repeat(N, 1, 1),
                                                       eta2_th.unsqueeze(0).repeat(N, 1, 1, 1))
    log_den = log_den + elp.view(1, K, 1)
    reg, d1, d2 = _regulariser(r_nk, log_num, log_den)
    return rec - reg, (rec, d1, d2, reg)


def compute_elbo_smm(y, reconstructions, theta, phi_tilde, x_k, log_z):
    """svae.py:265-322; theta = (alpha_nat, mu_k, L_k_raw, DoF)."""
    mu_th, sigma_th = unpack_smm(theta[1:3])
    elp = dists.dir_expected_log_pi(dists.dir_natural_to_standard(theta[0])).detach()
    dof = theta[3].detach()
    r_nk = torch.exp(log_z)
    rec = _reconstruction(y, reconstructions, r_nk)
    eta1_t, eta2_t = phi_tilde
    N, K, L, _ = eta2_t.shape
    log_num = dists.gauss_log_probability_nat_per_samp(x_k, eta1_t.reshape(N, K, L), eta2_t) + log_z.unsqueeze(2)
    log_den = dists.student_t_log_probability_per_samp(x_k, mu_th, sigma_th, dof) + elp.view(1, K, 1)
    reg, d1, d2 = _regulariser(r_nk, log_num, log_den)
    return rec - reg, (rec, d1, d2, reg)


def update_gmm_params(current, star, step_size):
    """svae.py:376-403: convex combination per tensor."""
    return [(1 - step_size) * c + step_size * s for c, s in zip(current, star)]


def init_mm(K, L, m_uniform, dtype=torch.float32):
    """svae.py:461-471: (prior, theta) natural parameters; `m_uniform` is the injected U[0,1) draw."""
    prior = mixtures.init_mm_params(K, L, alpha_scale=0.05 / K, beta_scale=0.5, m_scale=0, C_scale=L + 0.5,
                                    v_init=L + 0.5, m_uniform=m_uniform, dtype=dtype)
    theta = mixtures.init_mm_params(K, L, alpha_scale=1., beta_scale=1., m_scale=5., C_scale=2 * L, v_init=L + 1.,
                                    m_uniform=m_uniform, dtype=dtype)
    return prior, theta


def make_loc_scale(theta):
    """svae.py:474-485: (E[mu], chol(E[Sigma])) of the NIW `theta` (natural, 5-tuple)."""
    std = dists.niw_natural_to_standard(*theta[1:])
    mu, sigma = dists.niw_expected_values(*std)
    return mu, dists.chol(sigma)


def init_recognition_params(theta, pi_normal):
    """svae.py:488-496; `pi_normal` is the injected N(0,1) draw (K,)."""
    mu_k, L_k = make_loc_scale(theta)
    return [mu_k, L_k, torch.softmax(pi_normal, dim=-1)]


def inference(y, phi_gmm, enc_w, dec_w, noise, z_draws):
    """svae.py:499-516.  Returns (y_rec, phi_enc, x_k, x_samples (N,L), log_z, phi_gmm, phi_tilde)."""
    phi_enc = nets.encoder(y, enc_w)
    x_k, log_z, phi_tilde, _ = e_step(phi_enc, phi_gmm, noise)
    y_rec = nets.decoder(x_k, dec_w)
    x_s = subsample_x(x_k, z_draws)[:, 0, :]
    return y_rec, phi_enc, x_k, x_s, log_z, phi_gmm, phi_tilde
