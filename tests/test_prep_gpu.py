"""K-sized prep kernels (csrc/vmp_prep.hip) against the oracle's fp64 restatement of the reference's parameter maps:
unpack_recognition_gmm + the k-only part of compute_log_z_given_y (svae.py:342-358, 70-92) with torch autograd as the
gradient truth; the theta side of compute_elbo (niw.py:8-43, dirichlet.py:8-22); m_step + update_gmm_params
(svae.py:154-176, 376-403)."""
import math

import numpy as np
import pytest
import torch

import parity_log

pytestmark = pytest.mark.gpu


def relerr(got, want):
    want = want.detach().double().cpu()
    return parity_log.record('rel', ((got.detach().double().cpu() - want).abs().max() / want.abs().max().clamp_min(1e-300)).item())


@pytest.mark.parametrize('K,Ld', [(5, 2), (10, 6), (16, 8), (1, 1), (64, 3)])
def test_phi_prep_fwd_bwd(K, Ld):
    from oracle import svae_ref
    from vmp_for_svae_amd.models import _svae_ops
    rng = np.random.Generator(np.random.PCG64(K * 10 + Ld))
    mu, Lraw, pi = rng.standard_normal((K, Ld)) * 2, rng.standard_normal((K, Ld, Ld)), rng.standard_normal(K)
    g_h, g_P, g_b = rng.standard_normal((K, Ld)), rng.standard_normal((K, Ld, Ld)), rng.standard_normal(K)
    # truth: the oracle's unpack (tril / softplus / L L^T / softmax) + closed-form bias, fp64 autograd
    t = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (mu, Lraw, pi)]
    e1, e2, pik = svae_ref.unpack_recognition_gmm(t)
    P = -2.0 * e2
    Lc = torch.linalg.cholesky(P)
    sol = torch.linalg.solve_triangular(Lc, e1.unsqueeze(-1), upper=False).squeeze(-1)
    bias = -0.5 * (sol * sol).sum(-1) + torch.log(torch.diagonal(Lc, dim1=-2, dim2=-1)).sum(-1) + torch.log(pik)
    loss = (e1 * torch.tensor(g_h)).sum() + (P * torch.tensor(g_P)).sum() + (bias * torch.tensor(g_b)).sum()
    gt = torch.autograd.grad(loss, t)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device='cuda')
    d = [f32(a).requires_grad_(True) for a in (mu, Lraw, pi)]
    hk, Pd, bd = _svae_ops.PhiPrepFn.apply(*d)
    assert relerr(hk, e1) < 1e-6 and relerr(Pd, P) < 2e-6 and relerr(bd, bias) < 2e-6
    gd = torch.autograd.grad((hk * f32(g_h)).sum() + (Pd * f32(g_P)).sum() + (bd * f32(g_b)).sum(), d)
    for n_, a, b in zip(('mu_k', 'L_k', 'log_pi_k'), gd, gt):
        assert relerr(a, b) < 5e-6, (n_, relerr(a, b))
    assert torch.count_nonzero(torch.triu(gd[1], diagonal=1)) == 0


@pytest.mark.parametrize('K,Ld', [(5, 2), (10, 6), (16, 8)])
def test_theta_pack_and_cvi_update(K, Ld):
    from oracle import dists, svae_ref
    from vmp_for_svae_amd.models import _svae_ops, svae
    rng = np.random.Generator(np.random.PCG64(K + Ld))
    prior, theta = svae_ref.init_mm(K, Ld, torch.tensor(rng.random((K, Ld))), torch.float64)
    # a generic theta: random standard NIW / Dirichlet parameters (C SPD) mapped to natural form
    Mr = torch.tensor(rng.standard_normal((K, Ld, Ld)))
    C_ = Mr @ Mr.transpose(-1, -2) + Ld * torch.eye(Ld, dtype=torch.float64)
    beta_, m_, v_ = torch.tensor(0.5 + rng.random(K) * 30), torch.tensor(rng.standard_normal((K, Ld)) * 4), torch.tensor(Ld + 1.5 + rng.random(K) * 40)
    theta = [torch.tensor(rng.random(K) * 20)] + list(dists.niw_standard_to_natural(beta_, m_, C_, v_))
    theta = [t.float().double() for t in theta]          # the kernel sees fp32 parameters: truth from the same values
    # truth for the pack
    beta, m, C, v = dists.niw_natural_to_standard(*theta[1:])
    mu, sigma = dists.niw_expected_values(beta, m, C, v)
    W = torch.linalg.inv(torch.linalg.cholesky(0.5 * (sigma + sigma.transpose(-1, -2))))
    elp = dists.dir_expected_log_pi(dists.dir_natural_to_standard(theta[0]))
    kappa = torch.log(torch.diagonal(W, dim1=-2, dim2=-1)).sum(-1) - 0.5 * Ld * math.log(2 * math.pi) + elp
    f32 = lambda a: a.to('cuda', torch.float32).contiguous()
    th_d = [f32(t) for t in theta]
    md, Wd, kd = _svae_ops.theta_pack_gmm(th_d)
    assert relerr(md, mu) < 1e-6 and relerr(Wd, W) < 5e-6 and relerr(kd, kappa) < 2e-6
    # m-step in natural parameters + convex update
    stats = torch.tensor(rng.random((K, 2 + Ld + Ld * Ld)) * 50)
    Nk, sx, sxx = stats[:, 0], stats[:, 2:2 + Ld], stats[:, 2 + Ld:].reshape(K, Ld, Ld)
    star = [prior[0] + Nk, prior[1] + sxx, prior[2] + sx, prior[3] + Nk, prior[4] + Nk + 1.0]
    rho = 0.17
    want = svae_ref.update_gmm_params(theta, star, rho)
    pr_d = [f32(t) for t in prior]
    vers = [t._version for t in th_d]
    star_d = svae.cvi_update_from_stats(pr_d, th_d, stats.cuda(), rho)
    for a, b in zip(star_d, star):
        assert relerr(a, b) < 1e-6
    for a, b in zip(th_d, want):
        assert relerr(a, b) < 1e-6
    assert all(t._version > v0 for t, v0 in zip(th_d, vers))
