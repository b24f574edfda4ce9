"""T2/T3 parity on the GPU: the SVAE E-step, ELBO, all 21 gradients and full training steps (TF-Adam, CVI update)
through the reference-shaped surface (models.svae / models.vae) against the golden vectors produced by the
reference itself.  Tolerances: ELBO 1e-5 relative, responsibilities 1e-5 absolute, everything else
1e-5 relative to the fp64 truth - or 3x the reference's own fp32-vs-fp64 error where that is larger."""
import os

import numpy as np
import pytest
import torch

import parity_log

pytestmark = pytest.mark.gpu
NET_VARS = ('layer_0/kernel', 'layer_0/bias', 'layer_1/kernel', 'layer_1/bias', 'gaussian_output/kernel',
            'gaussian_output/bias', 'shortcut/W', 'shortcut/b1', 'shortcut/b2')


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a)).to('cuda', dtype)


def rel(got, want, what=None, tol=None):
    want = np.asarray(want, dtype=np.float64)
    got = got.detach().double().cpu().numpy()
    return parity_log.record('rel', np.abs(got - want).max() / max(np.abs(want).max(), 1e-300), tol, what)


def bar(g, key, base):
    if key + '__f32' not in g.files:          # slim fixtures keep the big per-sample outputs in fp64 only
        return base
    a, b = g[key], g[key + '__f32'].astype(np.float64)
    return max(base, 3 * np.abs(a - b).max() / max(np.abs(a).max(), 1e-300))


def make_trainer(g, **kw):
    from vmp_for_svae_amd.models import vae
    from vmp_for_svae_amd.training import SVAETrainer
    N, K, Ld, S, Dy, U, steps, smm = [int(v) for v in g['in_dims']]
    vae.reset_variables()
    for scope in ('encoder_net', 'decoder_net'):
        for v in NET_VARS:
            vae.VARIABLES[scope + '/' + v] = torch.nn.Parameter(dev(g['in_w_%s/%s' % (scope, v)]))
    tr = SVAETrainer(K, Ld, U, Dy, nb_samples=S, lr=float(g['in_lr']), lrcvi=float(g['in_lrcvi']),
                     decay_rate=float(g['in_decay']), m_uniform=dev(g['in_m_unif']), pi_normal=dev(g['in_pi_norm']),
                     smm=bool(smm), dof=float(g['in_dof0']), **kw)
    with torch.no_grad():
        tr.phi_gmm[1].add_(dev(g['in_Lk_low']))
    return tr, (N, K, Ld, S, Dy, U, steps)


@pytest.mark.parametrize('case', ['svae_tiny', 'svae_paper', 'svae_c1', 'svae_l8', 'svae_auto'])
def test_init_matches_reference(golden, case):
    g = golden(case)
    tr, _ = make_trainer(g)
    for n_, t in zip(('mu_k', 'L_k', 'log_pi_k'), tr.phi_gmm):
        assert rel(t, g['phi_init_' + n_]) < 1e-6, n_
    for n_, p, t in zip(('alpha', 'A', 'b', 'beta', 'vhat'), tr.gmm_prior, tr.theta):
        assert rel(p, g['prior_' + n_]) < 1e-6 and rel(t, g['theta_init_' + n_]) < 1e-6, n_


@pytest.mark.parametrize('case,literal', [('svae_tiny', False), ('svae_paper', False), ('svae_c1', False), ('svae_l8', False),
                                          ('svae_auto', False), ('svae_paper', True), ('svae_auto', True)])
def test_training_steps_vs_reference(golden, case, literal):
    """literal=True: the reference's own call order (experiments.py:209-229): svae.inference(y, phi_gmm, enc, dec, S)
    WITHOUT theta, then svae.compute_elbo(_smm)(y, y_k_rec, theta, phi_tilde, x_k_samples, log_z, ...) - same bars."""
    g = golden(case)
    tr, (N, K, Ld, S, Dy, U, steps) = make_trainer(g, reference_call_order=literal)
    y = dev(g['in_y'])
    for it in range(steps):
        pre = 'step%d_' % it
        noise, zd = dev(g['in_noise'][it]), dev(g['in_zdraw'][it], torch.int64)
        out = tr.step(y, noise=noise, z_draws=zd)
        slack = 1 + 2 * it                                  # free-running: errors compound through Adam / CVI
        assert rel(out['x_k'], g[pre + 'x_k']) <= slack * bar(g, pre + 'x_k', 1e-5), (it, 'x_k')
        r_err = np.abs(np.exp(out['log_z'].double().cpu().numpy()) - np.exp(g[pre + 'log_z'])).max()
        r_ref = np.abs(np.exp(g[pre + 'log_z']) - np.exp(g[pre + 'log_z__f32'].astype(np.float64))).max()
        parity_log.record('abs', r_err, slack * max(1e-5, 3 * r_ref), 'r_nk')
        assert r_err <= slack * max(1e-5, 3 * r_ref), (it, 'r_nk', r_err)
        e_true, e_f32 = float(g[pre + 'elbo']), float(g[pre + 'elbo__f32'])
        parity_log.record('rel', abs(out['elbo'].item() - e_true) / abs(e_true),
                          slack * max(1e-5, 3 * abs(e_f32 - e_true) / abs(e_true)), 'elbo')
        assert abs(out['elbo'].item() - e_true) <= slack * max(1e-5 * abs(e_true), 3 * abs(e_f32 - e_true)), (it, 'elbo')
        det = g[pre + 'details']
        assert abs(out['neg_rec_err'].item() - det[0]) <= slack * 2e-5 * abs(det[0])
        assert abs(out['regulariser'].item() - det[3]) <= slack * 2e-5 * max(abs(det[3]), abs(det[0]) * 0.1)
        assert rel(out['x_samples'], g[pre + 'x_s']) <= slack * bar(g, pre + 'x_s', 1e-5)
        for n_, gr in out['grads'].items():
            e = rel(gr, g[pre + 'grad_' + n_])
            assert e <= slack * bar(g, pre + 'grad_' + n_, 3e-5), (it, 'grad', n_, e)
        for n_, ts, t in zip(('alpha', 'A', 'b', 'beta', 'vhat'), out['theta_star'], tr.theta):
            assert rel(ts, g[pre + 'theta_star_' + n_]) <= slack * bar(g, pre + 'theta_star_' + n_, 1e-5), (it, 'theta*', n_)
            assert rel(t, g[pre + 'theta_' + n_]) <= slack * bar(g, pre + 'theta_' + n_, 1e-5), (it, 'theta', n_)
        names, params = tr.trainables()
        for n_, p in zip(names, params):
            assert rel(p, g[pre + 'param_' + n_]) <= slack * bar(g, pre + 'param_' + n_, 2e-5), (it, 'param', n_)


def test_estep_vs_oracle_shapes():
    """ragged K (not dividing 64), L=1..8, S odd, N not a multiple of the wave tile - forward and backward against
    the oracle's literal formulation in fp64 (autograd)."""
    from oracle import svae_ref, dists
    from vmp_for_svae_amd.models import svae
    rng = np.random.Generator(np.random.PCG64(5))
    for (N, K, Ld, S) in [(5, 3, 2, 3), (37, 10, 6, 10), (64, 16, 8, 10), (130, 7, 5, 4), (9, 33, 3, 2), (20, 5, 1, 7), (3, 64, 4, 5),
                          (21, 10, 2, 100), (11, 5, 8, 100), (7, 16, 6, 37),    # these three: S-chunked forward (L*S > 144)
                          # K = 16, even L >= 4, even S >= 4: the LDS-ring backward kernel (partial last tile, L/2 = 2, 3, 4
                          # pieces per pair, the shortest and a long sample loop)
                          (33, 16, 4, 10), (50, 16, 6, 8), (13, 16, 8, 4), (9, 16, 8, 100), (257, 16, 8, 10),
                          # K != 16 with even L, S (the round-2 kernel: a run-time-K variant of the ring kernel measured SLOWER
                          # than it at K = 10, 2.41 vs 2.14 ms at N = 1e6)
                          (64, 10, 8, 10), (21, 3, 4, 6), (40, 19, 8, 10), (70, 1, 8, 10),
                          # 8 <= K < 16 on the ring kernel (round 4: 64 // K whole rows per wave tile), several blocks, ragged ends
                          (2050, 10, 8, 10), (700, 12, 6, 4), (333, 9, 4, 8), (1000, 8, 8, 6), (515, 15, 8, 10), (1201, 13, 6, 10),
                          # ... which takes over from the generic kernel's one-tile-per-wave form above 2048 wave tiles only:
                          (12500, 10, 8, 4), (8301, 15, 8, 4), (16501, 8, 6, 4), (14403, 9, 4, 6), (10302, 12, 8, 4), (8303, 13, 6, 4),
                          (8205, 16, 8, 4),
                          # L = 8 with S / 2 odd: cells that start in mid-line take their pairs in rotated order (3 and 7 pairs)
                          (300, 16, 8, 6), (50, 16, 8, 14), (12501, 10, 8, 6),
                          # >= 1024 wave tiles with S / 2 >= 4: the shapes a four-stage build (-DVMP_RING_STAGES=4) takes
                          (4101, 16, 8, 10), (6201, 10, 8, 8),
                          # odd K (an odd number of cells per tile: the parity of a tile's first cell alternates) with rotated pairs
                          (8301, 15, 8, 6), (14405, 9, 8, 6),
                          # small L at streaming sizes (C2's latent dimension): several blocks per CU since round 6 (> 512 backward blocks)
                          (40000, 10, 2, 10), (30000, 7, 3, 4), (25000, 16, 1, 6)]:
        e1 = rng.standard_normal((N, Ld))
        e2 = -0.5 * (0.3 + rng.random((N, Ld)))
        mu_k = rng.standard_normal((K, Ld)) * 2
        Lraw = rng.standard_normal((K, Ld, Ld)) * 0.4
        pir = rng.standard_normal(K)
        noise = rng.standard_normal((N, K, Ld, S))
        m_unif = rng.random((K, Ld))

        def to(a, g=False):
            return torch.tensor(a, dtype=torch.float64, requires_grad=g)
        oe1, oe2, omu, oL, opi = to(e1, True), to(e2, True), to(mu_k, True), to(Lraw, True), to(pir, True)
        prior, theta = svae_ref.init_mm(K, Ld, to(m_unif), torch.float64)
        x_o, lz_o, pt_o, _ = svae_ref.e_step((oe1, oe2), [omu, oL, opi], to(noise))
        bk, mk, Ck, vk = dists.niw_natural_to_standard(*theta[1:])
        mu_t, sig_t = dists.niw_expected_values(bk, mk, Ck, vk)
        e1t, e2t = dists.gauss_standard_to_natural(mu_t, sig_t)
        elp = dists.dir_expected_log_pi(dists.dir_natural_to_standard(theta[0]))
        num = dists.gauss_log_probability_nat_per_samp(x_o, pt_o[0].reshape(N, K, Ld), pt_o[1])
        den = dists.gauss_log_probability_nat_per_samp(x_o, e1t.unsqueeze(0).repeat(N, 1, 1),
                                                       e2t.unsqueeze(0).repeat(N, 1, 1, 1)) + elp.view(1, K, 1)
        Tp_o = (num - den).mean(-1)
        wx = to(rng.standard_normal((N, K, S, Ld)))
        loss_o = (x_o * wx).sum() + (torch.exp(lz_o) * (Tp_o + lz_o)).sum()
        go = torch.autograd.grad(loss_o, [oe1, oe2, omu, oL, opi])

        def f(a, g=False):
            return torch.tensor(a, dtype=torch.float32, device='cuda', requires_grad=g)
        pe1, pe2, pmu, pL, ppi = f(e1, True), f(e2, True), f(mu_k, True), f(Lraw, True), f(pir, True)
        th = [t.float().cuda() for t in theta]
        x_p, lz_p, pt_p, _ = svae.e_step((pe1, pe2), [pmu, pL, ppi], S, noise=f(noise), theta=th)
        loss_p = (x_p * wx.float().cuda()).sum() + (torch.exp(lz_p) * (pt_p.T_prime + lz_p)).sum()
        gp = torch.autograd.grad(loss_p, [pe1, pe2, pmu, pL, ppi])
        tag = (N, K, Ld, S)
        # bars: the stated 1e-5 (x, T' relative to their largest entry; r absolute), or 3 x what the reference's OWN fp32 arithmetic
        # loses on this very shape where that is more - the oracle's literal graph in fp32, measured here
        with torch.no_grad():
            f32c = lambda a: torch.tensor(a, dtype=torch.float32)
            pr32, th32 = svae_ref.init_mm(K, Ld, f32c(m_unif), torch.float32)
            x32, lz32, pt32, _ = svae_ref.e_step((f32c(e1), f32c(e2)), [f32c(mu_k), f32c(Lraw), f32c(pir)], f32c(noise))
            b32, m32, C32, v32 = dists.niw_natural_to_standard(*th32[1:])
            mu32, sig32 = dists.niw_expected_values(b32, m32, C32, v32)
            e1t32, e2t32 = dists.gauss_standard_to_natural(mu32, sig32)
            elp32 = dists.dir_expected_log_pi(dists.dir_natural_to_standard(th32[0]))
            Tp32 = (dists.gauss_log_probability_nat_per_samp(x32, pt32[0].reshape(N, K, Ld), pt32[1])
                    - dists.gauss_log_probability_nat_per_samp(x32, e1t32.unsqueeze(0).repeat(N, 1, 1), e2t32.unsqueeze(0).repeat(N, 1, 1, 1))
                    - elp32.view(1, K, 1)).mean(-1)
        ref_x = float((x32.double() - x_o.detach()).abs().max() / x_o.detach().abs().max())
        ref_r = float((torch.exp(lz32).double() - torch.exp(lz_o.detach())).abs().max())
        ref_T = float((Tp32.double() - Tp_o.detach()).abs().max() / Tp_o.detach().abs().max())
        bx, br, bT = max(1e-5, 3 * ref_x), max(1e-5, 3 * ref_r), max(1e-5, 3 * ref_T)
        assert rel(x_p, x_o.detach().numpy(), 'e_step shapes x', bx) <= bx, (tag, ref_x)
        e_r = np.abs(np.exp(lz_p.detach().double().cpu().numpy()) - np.exp(lz_o.detach().numpy())).max()
        parity_log.record('abs', e_r, br, 'e_step shapes r_nk')
        assert e_r <= br, (tag, e_r, ref_r)
        assert rel(pt_p.T_prime, Tp_o.detach().numpy(), "e_step shapes T'", bT) <= bT, (tag, ref_T)
        for a_, b_, n_ in zip(gp, go, ('eta1', 'eta2d', 'mu_k', 'L_k', 'log_pi_k')):
            # fp32 accumulation over N*S sample adjoints: tolerance 2e-4 at S<=10, growing as sqrt(S/10)
            assert rel(a_, b_.numpy()) < 2e-4 * max(1.0, S / 10.0) ** 0.5, (tag, n_, rel(a_, b_.numpy()))


@pytest.mark.parametrize('N,K,Ld,S', [(64, 10, 8, 10), (100, 16, 8, 10), (7, 3, 2, 3), (130, 7, 5, 4), (300, 5, 8, 16), (64, 10, 6, 1), (1000, 16, 8, 10)])
def test_minibatch_backward_form_equals_the_streaming_forms(N, K, Ld, S):
    """svae_estep_bwd1_kernel (round 6: one block per tile, one wave per sample pair; vmp_svae_estep_bwd_n with
    nblk = vmp_svae_bwd_blocks_for) against the generic / ring kernels (vmp_svae_estep_bwd) on the same inputs: the N-sized
    gradients and the reduced K-sized gradients agree to fp32 rounding (the per-cell sums over samples are formed per pair first)."""
    from vmp_for_svae_amd import _lib as L
    from vmp_for_svae_amd.models import svae
    g = torch.Generator(device='cuda').manual_seed(N + K)
    f32 = dict(dtype=torch.float32, device='cuda')
    eta1 = torch.randn(N, Ld, generator=g, **f32)
    eta2d = -0.5 * torch.nn.functional.softplus(torch.randn(N, Ld, generator=g, **f32))
    prior, theta = svae.init_mm(K, Ld, seed=0, param_device='cuda')
    phi = list(svae.init_recognition_params(theta, K, seed=0, param_device='cuda'))
    with torch.no_grad():
        hk, P, bias, mk, Wk, kap = svae.recognition_prep(phi, theta)
        noise = torch.randn(N, K, Ld, S, generator=g, **f32)
        x, lz, pt, _ = svae.e_step((eta1, eta2d), phi, S, noise=noise, theta=theta)
    Gx, Glz, GT = torch.randn(N, K, S, Ld, generator=g, **f32), torch.randn(N, K, generator=g, **f32), torch.rand(N, K, generator=g, **f32)
    PW = L.lib().vmp_svae_bwd_partial_words(Ld)
    outs = []
    for mode in ('minibatch', 'streaming'):
        nblk = L.lib().vmp_svae_bwd_blocks_for(N, K, Ld, S, 0) if mode == 'minibatch' else L.lib().vmp_svae_bwd_blocks(N, K)
        g1, g2, part = torch.empty(N, Ld, **f32), torch.empty(N, Ld, **f32), torch.empty(nblk, K, PW, **f32)
        L.check(L.lib().vmp_svae_estep_bwd_n(L.ptr(eta1), L.ptr(eta2d), L.ptr(hk.contiguous()), L.ptr(P.contiguous()), L.ptr(bias), L.ptr(mk), L.ptr(Wk), None,
                                             L.ptr(x), L.ptr(lz), L.ptr(Gx), L.ptr(Glz), L.ptr(GT), N, K, Ld, S, L.ptr(g1), L.ptr(g2), L.ptr(part),
                                             part.numel() * 4, nblk, L.stream()), 'vmp_svae_estep_bwd_n')
        outs.append((g1, g2, part.double().sum(0)[:, :PW // 2]))
    assert L.lib().vmp_svae_bwd_blocks_for(N, K, Ld, S, 0) == (N + 64 // K - 1) // (64 // K)      # one partial row per tile
    for a_, b_, n_ in zip(outs[0], outs[1], ('g_eta1', 'g_eta2d', 'K-sized sums')):
        e = ((a_ - b_).abs().max() / b_.abs().max().clamp_min(1e-30)).item()
        assert e < 2e-5, (n_, e)


def test_estep_vs_oracle_shapes_student_t():
    """The same comparison with a Student-t theta (svae.py:265-322, student_t.py:7-39: the theta term of T' is
    (nu+L)/2 log1p(delta^2/nu), theta/mu_k and theta/L_k are trainable): shapes with SEVERAL blocks of the backward kernels, so
    that the N*K*S-reduced gradients of mu_k / L_k cross the per-block partials and the reduce kernel (the goldens hold
    N = 7 and N = 10: one block).  K = 16 and 8 <= K < 16 with even L, S take the LDS-ring kernel, the rest the generic one."""
    from oracle import svae_ref, dists
    from vmp_for_svae_amd.models import svae
    rng = np.random.Generator(np.random.PCG64(15))
    for (N, K, Ld, S) in [(300, 16, 8, 10), (1500, 16, 8, 10), (1031, 16, 6, 4), (700, 10, 8, 10), (2050, 10, 8, 10), (999, 12, 4, 6),
                          (411, 7, 5, 3), (800, 16, 8, 5), (513, 9, 8, 8), (257, 33, 2, 4),
                          # 8 <= K < 16: the ring kernel takes over above 2048 wave tiles (64 // K rows each)
                          (12500, 10, 8, 4), (8301, 15, 6, 4), (16501, 8, 8, 4), (10302, 12, 4, 6), (8303, 13, 8, 4),
                          (300, 16, 8, 6), (12501, 10, 8, 6),          # rotated pair order (S / 2 odd at L = 8)
                          (4101, 16, 8, 10), (6201, 10, 8, 8),         # >= 1024 wave tiles with S / 2 >= 4 (four-stage builds)
                          (8301, 15, 8, 6)]:                           # odd K with rotated pairs
        e1 = rng.standard_normal((N, Ld))
        e2 = -0.5 * (0.3 + rng.random((N, Ld)))
        mu_k = rng.standard_normal((K, Ld)) * 2
        Lraw = rng.standard_normal((K, Ld, Ld)) * 0.4
        pir = rng.standard_normal(K)
        noise = rng.standard_normal((N, K, Ld, S))
        th_mu = rng.standard_normal((K, Ld)) * 1.5
        th_L = np.tril(rng.standard_normal((K, Ld, Ld)) * 0.4) + 0.8 * np.eye(Ld)
        alpha_nat = rng.random(K) * 3.0
        dof = 2.5 + 5.0 * rng.random(K)

        def to(a, g=False):
            return torch.tensor(a, dtype=torch.float64, requires_grad=g)
        oe1, oe2, omu, oL, opi, otm, otL = to(e1, True), to(e2, True), to(mu_k, True), to(Lraw, True), to(pir, True), to(th_mu, True), to(th_L, True)
        x_o, lz_o, pt_o, _ = svae_ref.e_step((oe1, oe2), [omu, oL, opi], to(noise))
        mu_th, sig_th = svae_ref.unpack_smm([otm, otL])
        elp = dists.dir_expected_log_pi(dists.dir_natural_to_standard(to(alpha_nat)))
        num = dists.gauss_log_probability_nat_per_samp(x_o, pt_o[0].reshape(N, K, Ld), pt_o[1])
        den = dists.student_t_log_probability_per_samp(x_o, mu_th, sig_th, to(dof)) + elp.view(1, K, 1)
        Tp_o = (num - den).mean(-1)
        wx = to(rng.standard_normal((N, K, S, Ld)))
        loss_o = (x_o * wx).sum() + (torch.exp(lz_o) * (Tp_o + lz_o)).sum()
        go = torch.autograd.grad(loss_o, [oe1, oe2, omu, oL, opi, otm, otL])

        def f(a, g=False):
            return torch.tensor(a, dtype=torch.float32, device='cuda', requires_grad=g)
        pe1, pe2, pmu, pL, ppi, ptm, ptL = f(e1, True), f(e2, True), f(mu_k, True), f(Lraw, True), f(pir, True), f(th_mu, True), f(th_L, True)
        th = [f(alpha_nat), ptm, ptL, f(dof)]
        x_p, lz_p, pt_p, _ = svae.e_step((pe1, pe2), [pmu, pL, ppi], S, noise=f(noise), theta=th)
        loss_p = (x_p * wx.float().cuda()).sum() + (torch.exp(lz_p) * (pt_p.T_prime + lz_p)).sum()
        gp = torch.autograd.grad(loss_p, [pe1, pe2, pmu, pL, ppi, ptm, ptL])
        tag = (N, K, Ld, S)
        # bars: the stated 1e-5 (x, T' relative to their largest entry; r absolute), or 3 x what the reference's OWN fp32 arithmetic
        # loses on this very shape where that is more - the oracle's literal graph in fp32, measured here
        with torch.no_grad():
            f32c = lambda a: torch.tensor(a, dtype=torch.float32)
            x32, lz32, pt32, _ = svae_ref.e_step((f32c(e1), f32c(e2)), [f32c(mu_k), f32c(Lraw), f32c(pir)], f32c(noise))
            mu32, sig32 = svae_ref.unpack_smm([f32c(th_mu), f32c(th_L)])
            elp32 = dists.dir_expected_log_pi(dists.dir_natural_to_standard(f32c(alpha_nat)))
            Tp32 = (dists.gauss_log_probability_nat_per_samp(x32, pt32[0].reshape(N, K, Ld), pt32[1])
                    - dists.student_t_log_probability_per_samp(x32, mu32, sig32, f32c(dof)) - elp32.view(1, K, 1)).mean(-1)
        ref_x = float((x32.double() - x_o.detach()).abs().max() / x_o.detach().abs().max())
        ref_r = float((torch.exp(lz32).double() - torch.exp(lz_o.detach())).abs().max())
        ref_T = float((Tp32.double() - Tp_o.detach()).abs().max() / Tp_o.detach().abs().max())
        bx, br, bT = max(1e-5, 3 * ref_x), max(1e-5, 3 * ref_r), max(1e-5, 3 * ref_T)
        assert rel(x_p, x_o.detach().numpy(), 'e_step shapes (Student-t) x', bx) <= bx, (tag, ref_x)
        e_r = np.abs(np.exp(lz_p.detach().double().cpu().numpy()) - np.exp(lz_o.detach().numpy())).max()
        parity_log.record('abs', e_r, br, 'e_step shapes (Student-t) r_nk')
        assert e_r <= br, (tag, e_r, ref_r)
        assert rel(pt_p.T_prime, Tp_o.detach().numpy(), "e_step shapes (Student-t) T'", bT) <= bT, (tag, ref_T)
        for a_, b_, n_ in zip(gp, go, ('eta1', 'eta2d', 'mu_k', 'L_k', 'log_pi_k', 'theta/mu_k', 'theta/L_k')):
            e = rel(a_, b_.numpy())
            assert e < 2e-4 * max(1.0, S / 10.0) ** 0.5, (tag, n_, e)


@pytest.mark.parametrize('case,literal', [('svae_smm_tiny', False), ('svae_smm_l8', False), ('svae_smm_l8', True)])
def test_smm_training_steps_vs_reference(golden, case, literal):
    """Student-t mixture SVAE (BASELINE config 5 model): compute_elbo_smm, trainable theta/mu_k, theta/L_k,
    Dirichlet-only CVI update (experiments.py:154-176, 252-256; svae.py:265-322; student_t.py:7-39).
    literal=True: inference(...) without theta, then compute_elbo_smm(..., theta, phi_tilde, ...) as experiments.py:209-224."""
    g = golden(case)
    tr, (N, K, Ld, S, Dy, U, steps) = make_trainer(g, reference_call_order=literal)
    assert rel(tr.gmm_prior, g['prior_alpha']) < 1e-6
    for n_, t in zip(('alpha', 'mu', 'L', 'dof'), tr.theta):
        assert rel(t, g['theta_init_' + n_]) < 1e-6, n_
    y = dev(g['in_y'])
    for it in range(steps):
        pre = 'step%d_' % it
        out = tr.step(y, noise=dev(g['in_noise'][it]), z_draws=dev(g['in_zdraw'][it], torch.int64))
        slack = 1 + 2 * it
        e_true, e_f32 = float(g[pre + 'elbo']), float(g[pre + 'elbo__f32'])
        assert abs(out['elbo'].item() - e_true) <= slack * max(1e-5 * abs(e_true), 3 * abs(e_f32 - e_true)), (it, 'elbo')
        det = g[pre + 'details']
        assert abs(out['regulariser'].item() - det[3]) <= slack * 2e-5 * max(abs(det[3]), abs(det[0]) * 0.1)
        assert rel(out['x_k'], g[pre + 'x_k']) <= slack * bar(g, pre + 'x_k', 1e-5)
        for n_, gr in out['grads'].items():
            e = rel(gr, g[pre + 'grad_' + n_])
            assert e <= slack * bar(g, pre + 'grad_' + n_, 3e-5), (it, 'grad', n_, e)
        assert rel(out['theta_star'][0], g[pre + 'theta_star_alpha']) <= slack * 1e-5
        assert rel(tr.theta[0], g[pre + 'theta_alpha']) <= slack * 1e-5
        names, params = tr.trainables()
        assert 'theta/mu_k' in names and 'theta/L_k' in names
        for n_, p in zip(names, params):
            assert rel(p, g[pre + 'param_' + n_]) <= slack * bar(g, pre + 'param_' + n_, 2e-5), (it, 'param', n_)


def test_t2_full_size_properties():
    """BASELINE config-3 shaped E-step (L=8, K=16, S=10) at N=2e5 rows (3.2e6 cells): size-independent invariants."""
    from vmp_for_svae_amd.models import svae
    N, K, Ld, S = 200_000, 16, 8, 10
    g = torch.Generator(device='cuda').manual_seed(3)
    e1 = torch.randn(N, Ld, device='cuda', generator=g).requires_grad_(True)
    e2 = (-0.5 * torch.nn.functional.softplus(torch.randn(N, Ld, device='cuda', generator=g))).requires_grad_(True)
    prior, theta = svae.init_mm(K, Ld, seed=0, param_device='cuda')
    phi = [p.detach().requires_grad_(True) for p in svae.init_recognition_params(theta, K, seed=0, param_device='cuda')]
    with torch.no_grad():
        phi[1].add_(torch.tril(torch.randn(K, Ld, Ld, device='cuda', generator=g) * 0.2, -1))
    # (a) zero noise: every sample equals the cell mean mu~ = Pt^-1 ht   (checked as Pt x = ht in fp64)
    x0, lz0, pt0, _ = svae.e_step((e1, e2), phi, S, noise=torch.zeros(N, K, Ld, S, device='cuda'), theta=theta)
    assert (x0[:, :, 1:, :] - x0[:, :, :1, :]).abs().max().item() == 0.0
    eta1_k, eta2_k, _ = svae.unpack_recognition_gmm(phi)
    idx = torch.randint(0, N, (4096,), device='cuda', generator=g)
    Pt = (torch.diag_embed(-2.0 * e2[idx]).unsqueeze(1) + (-2.0 * eta2_k).unsqueeze(0)).double()
    ht = (e1[idx].unsqueeze(1) + eta1_k.unsqueeze(0)).double()
    res = torch.einsum('nkij,nkj->nki', Pt, x0[idx, :, 0, :].double()) - ht
    assert (res.abs().max() / ht.abs().max()).item() < 2e-5
    # (b) responsibilities are distributions; T' finite
    r = torch.exp(lz0.double())
    assert (r.sum(1) - 1).abs().max().item() < 1e-5 and torch.isfinite(pt0.T_prime).all()
    # (c) run-to-run determinism, forward and backward
    noise = torch.randn(N, K, Ld, S, device='cuda', generator=g)
    Gx = torch.randn(N, K, S, Ld, device='cuda', generator=g)
    Glz = torch.randn(N, K, device='cuda', generator=g)

    def run(scale):
        x, lz, pt, _ = svae.e_step((e1, e2), phi, S, noise=noise, theta=theta)
        gr = torch.autograd.grad([x, lz, pt.T_prime], [e1, e2] + phi, [scale * Gx, scale * Glz, scale * torch.exp(lz.detach())])
        return x, lz, gr
    x1, lz1, g1 = run(1.0)
    x2, lz2, g2 = run(1.0)
    assert torch.equal(x1, x2) and torch.equal(lz1, lz2)
    for a_, b_ in zip(g1, g2):
        assert torch.equal(a_, b_)
    # (d) the backward pass is linear in the upstream gradients
    _, _, g3 = run(2.0)
    for a_, b_ in zip(g1, g3):
        assert ((2 * a_ - b_).abs().max() / b_.abs().max()).item() < 1e-5
    # (e) shift invariance of the softmax: adding a constant to the component log-weights leaves log_z unchanged
    phi_s = [phi[0], phi[1], (phi[2] + 3.0).detach()]
    _, lz_s, _, _ = svae.e_step((e1, e2), phi_s, S, noise=noise, theta=theta)
    assert (torch.exp(lz_s) - torch.exp(lz1)).abs().max().item() < 1e-6


@pytest.mark.parametrize('case', ['metrics', 'metrics_s100'])
def test_eval_metrics_golden(golden, case):
    """weighted_mse, diagonal_gaussian_logprob (incl. per-sample weights and the missing-data mask)
    and purity vs the reference run (losses.py)."""
    from vmp_for_svae_amd import losses
    g = golden(case)
    y, mean, var, lw, lws = [dev(g['in_' + k]) for k in ('y', 'mean', 'var', 'lw', 'lws')]
    r = torch.exp(lw)
    assert rel(losses.weighted_mse(y, mean, r), g['weighted_mse']) < 1e-5
    assert rel(losses.diagonal_gaussian_logprob(y, mean, var, lw), g['loli']) < 1e-5
    assert rel(losses.diagonal_gaussian_logprob(y, mean, var, lws), g['loli_s']) < 1e-5
    assert rel(losses.diagonal_gaussian_logprob(y, mean, var, lw, mask=dev(g['in_mask'], torch.bool)), g['loli_mask']) < 1e-5
    e, p_ = losses.purity(r, dev(g['in_labels']))
    assert rel(e, g['entropy']) < 1e-5 and rel(p_, g['purity']) < 1e-5


def test_imputation_golden(golden):
    """generate_missing_data_mask, perturb_data, imputation_mse, imputation_losses
    (reference losses.py:148-310) vs the reference run; the (N,K,S,D)-sized parts in the eval HIP kernel."""
    from vmp_for_svae_amd import losses
    g = golden('imputation')
    N, K, S, D, P = [int(v) for v in g['in_dims']]
    y, noise = dev(g['in_y']), dev(g['in_noise'])
    mask = losses.generate_missing_data_mask(y, float(g['in_mask_ratio']), seed=int(g['in_mask_seed']))
    assert mask.dtype == torch.bool and np.array_equal(mask.cpu().numpy(), g['mask'].astype(bool))
    for typ, n_, key in (('quarter', 3, 'mask_quarter'), ('left_half', 2, 'mask_left_half')):
        m2 = losses.generate_missing_data_mask(torch.zeros(n_, 16, device='cuda'), mask_type=typ)
        assert np.array_equal(m2.cpu().numpy(), g[key].astype(bool))
    assert rel(losses.perturb_data(y, mask, 0, noise=noise[0]), g['perturbed0']) < 1e-6
    assert rel(losses.imputation_mse(y, dev(g['in_y_pred']), dev(g['in_r']), mask), g['imputation_mse']) < 1e-5
    a_ks, b_ksd, Wr = [dev(g['in_' + k]) for k in ('a_ks', 'b_ksd', 'Wr')]

    def method(y_pert):
        mean = (y_pert[:, None, None, :] * a_ks[None, :, :, None] + b_ksd[None]).contiguous()
        var = (0.3 + 0.5 * (y_pert[:, None, None, :] * a_ks[None, :, :, None]) ** 2).contiguous()
        logits = y_pert @ Wr
        return mean, var, logits - torch.logsumexp(logits, dim=1, keepdim=True)
    mse, ll = losses.imputation_losses(y, mask, method, nb_samples_pert=P, nb_samples_rec=S, noise=noise)
    assert rel(mse, g['imp_mse']) < 1e-5 and rel(ll, g['imp_loglike']) < 1e-5
    # default noise path: different perturbations, finite results
    mse2, ll2 = losses.imputation_losses(y, mask, method, nb_samples_pert=3, nb_samples_rec=S, seed=1)
    assert torch.isfinite(mse2) and torch.isfinite(ll2)


def test_checkpoint_resume_and_imputation_eval(tmp_path):
    """8f rank 3/2: a run restored from save_checkpoint continues bit-for-bit (parameters, theta, Adam slots, step
    counters); the imputation measurement of experiments.py:361-377 runs on the trained model."""
    from vmp_for_svae_amd import experiments, losses
    from vmp_for_svae_amd.models import vae
    from vmp_for_svae_amd.training import SVAETrainer
    K, Ld, U, Dy, S, N = 6, 3, 20, 4, 5, 96
    g = torch.Generator(device='cuda').manual_seed(11)
    y = torch.randn(N, Dy, device='cuda', generator=g) * 2
    noises = [torch.randn(N, K, Ld, S, device='cuda', generator=g) for _ in range(5)]
    zd = [torch.randint(0, K, (N, S), device='cuda', generator=g) for _ in range(5)]

    def fresh():
        vae.reset_variables()
        return SVAETrainer(K, Ld, U, Dy, nb_samples=S, lr=1e-2, lrcvi=0.3, decay_rate=0.95, stddev_init_nn=0.1)
    tr = fresh()
    for i in range(3):
        tr.step(y, noise=noises[i], z_draws=zd[i])
    path = experiments.save_checkpoint(tr, str(tmp_path / 'ck.npz'))
    for i in range(3, 5):
        tr.step(y, noise=noises[i], z_draws=zd[i])
    want = {k: v.copy() for k, v in experiments.checkpoint_state(tr).items()}
    tr2 = experiments.load_checkpoint(fresh(), path)
    assert tr2.global_step == 3
    for i in range(3, 5):
        tr2.step(y, noise=noises[i], z_draws=zd[i])
    got = experiments.checkpoint_state(tr2)
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    mask = losses.generate_missing_data_mask(y, 0.25, seed=0)
    m = experiments.evaluate_imputation(tr2, y, mask, nb_samples_pert=3, nb_samples_te=7, seed=0)
    assert np.isfinite(m['imp_mse']) and np.isfinite(m['imp_logprob']) and m['imp_mse'] > 0


def test_graphed_step_matches_eager():
    """The HIP-graph replay of the training step (training.GraphedSVAEStep) follows the eager step: same noise and
    uniforms in, same parameters / theta / ELBO out (Adam's update is algebraically the same, rounded differently)."""
    from vmp_for_svae_amd.models import vae
    from vmp_for_svae_amd.training import SVAETrainer, GraphedSVAEStep
    K, Ld, U, Dy, S, N = 10, 6, 50, 6, 10, 64
    g = torch.Generator(device='cuda').manual_seed(5)
    ys = [torch.randn(N, Dy, device='cuda', generator=g) * 2 for _ in range(4)]

    def fresh():
        vae.reset_variables()
        return SVAETrainer(K, Ld, U, Dy, nb_samples=S, lr=3e-3, lrcvi=0.2, decay_rate=0.95, stddev_init_nn=0.1, seed=3, rng='torch')
    # eager reference run: call i of the graphed stepper must be training step i (the warm-up steps of the capture do
    # not train and do not consume the generator)
    tr = fresh()
    gen = torch.Generator(device='cuda').manual_seed(3)
    noise, u = torch.empty(N, K, Ld, S, device='cuda'), torch.empty(N, 1, device='cuda')
    elbos, lrcvis = [], []
    for i in range(4):
        noise.normal_(generator=gen)
        u.uniform_(generator=gen)
        o = tr.step(ys[i], noise=noise, u=u)
        elbos.append(o['elbo'].item())
        lrcvis.append(o['lrcvi'])
    _, want = tr.trainables()
    want = [p.detach().clone() for p in want] + [t.clone() for t in tr.theta]
    # graphed run
    tr2 = fresh()
    gs = GraphedSVAEStep(tr2, ys[0], warmup=3)
    assert tr2.global_step == 0 and tr2.opt.t == 0
    got_elbo = []
    for i in range(4):
        out = gs(ys[i])
        got_elbo.append(out['elbo'].item())
        assert out['lrcvi'] == lrcvis[i]
    assert tr2.global_step == 4 and tr2.opt.t == 4
    _, got = tr2.trainables()
    got = list(got) + list(tr2.theta)
    for a, b in zip(got_elbo, elbos):
        assert abs(a - b) <= 2e-5 * abs(b), (a, b)
    for a, b in zip(got, want):
        assert rel(a, b.double().cpu().numpy()) < 2e-5
    # scratch buffers are private to (device, stream, thread): the capture stream's buffers are not the eager stream's,
    # so a later, larger eager step (which re-allocates ITS buffer) cannot touch what the graph points into
    import vmp_for_svae_amd as V
    cur = torch.cuda.current_stream().cuda_stream
    before = dict(V._lib._WS)
    assert any(k[2] != cur for k in gs._ws_refs), 'capture used the eager stream\'s scratch'
    tr2.step(torch.randn(512, Dy, device='cuda', generator=g))          # eager step, larger workspace
    assert all(V._lib._WS[k] is before[k] for k in before if k[2] != cur)
    out = gs(ys[0])
    assert torch.isfinite(out['elbo']) and all(torch.isfinite(p).all() for p in tr2.trainables()[1])


@pytest.mark.parametrize('N,K,Ld,U,Dy,S', [(64, 10, 8, 50, 6, 10), (100, 16, 8, 64, 8, 10), (37, 7, 4, 33, 3, 8), (5, 3, 2, 16, 1, 4),
                                          (512, 8, 6, 50, 6, 10), (40, 20, 4, 32, 4, 6), (30, 33, 2, 16, 2, 4), (8, 64, 8, 64, 8, 10),
                                          (64, 10, 8, 50, 6, 16), (1, 1, 1, 1, 1, 4)])
def test_direct_minibatch_step_equals_the_autograd_step(N, K, Ld, U, Dy, S):
    """Round 6: SVAETrainer._step_direct (7 launches: lazy partial reductions, the ELBO tail inside the E-step backward, ONE closing
    launch for partial rows -> phi_gmm gradients, both reductions, Adam, moments + CVI, scalars) against the autograd step
    over the stand-alone launches of the same device functions: every gradient, parameter, Adam slot, theta tensor and moment
    BIT-identical over 4 steps; the three ELBO scalars (summed per tile instead of per tail block, fp64) to 1e-6.  Also the graphed
    replay of the direct step against the eager one (same Philox stream: call i == step i)."""
    from vmp_for_svae_amd.models import vae
    from vmp_for_svae_amd.training import SVAETrainer, GraphedSVAEStep
    g = torch.Generator(device='cuda').manual_seed(N + K)
    ys = [torch.randn(N, Dy, device='cuda', generator=g) * 2 for _ in range(4)]

    def run(direct, graphed=False):
        vae.reset_variables()
        tr = SVAETrainer(K, Ld, U, Dy, nb_samples=S, lr=3e-3, lrcvi=0.2, decay_rate=0.95, stddev_init_nn=0.1, seed=3, direct_step=direct)
        assert tr._direct_ok(ys[0], None, None, None, None) == direct
        stepper = GraphedSVAEStep(tr, ys[0]) if graphed else tr.step
        outs = []
        for i in range(4):
            o = stepper(ys[i])
            outs.append(dict(elbo=[o[k].item() for k in ('elbo', 'neg_rec_err', 'regulariser')],
                             grads={k: v.detach().clone() for k, v in o['grads'].items()},
                             star=[t.clone() for t in o['theta_star']], log_z=o['log_z'].clone(), xs=o['x_samples'].clone()))
        assert tr.global_step == 4 and tr.opt.t == 4
        state = [p.detach().clone() for p in tr.trainables()[1]] + [t.clone() for t in tr.theta] + [t.clone() for t in tr.opt.m] + \
                [t.clone() for t in tr.opt.v]
        return outs, state
    want_o, want_s = run(False)
    for graphed in (False, True):
        got_o, got_s = run(True, graphed)
        for i, (a, b) in enumerate(zip(got_o, want_o)):
            for x, y_ in zip(a['elbo'], b['elbo']):
                assert abs(x - y_) <= 1e-6 * max(abs(b['elbo'][1]), abs(b['elbo'][2])), (i, a['elbo'], b['elbo'])
            for k in b['grads']:
                assert torch.equal(a['grads'][k], b['grads'][k]), (graphed, i, k, rel(a['grads'][k], b['grads'][k].double().cpu().numpy()))
            for x, y_ in zip(a['star'], b['star']):
                assert torch.equal(x, y_), (graphed, i)
            assert torch.equal(a['log_z'], b['log_z']) and torch.equal(a['xs'], b['xs'])
        for j, (a, b) in enumerate(zip(got_s, want_s)):
            assert torch.equal(a, b), (graphed, j)


def test_graphed_step_scalar_table_refills_and_follows_the_trainer():
    """Round 6: a replayed direct step takes [Philox key | CVI step size | Adam step size] from a table the host fills for the coming
    steps (no eager launch per replay).  With a 3-row table, 9 replays with an eager tr.step() in between (the trainer's counters move
    outside the graph object: the table must be rebuilt) leave exactly the state of 10 eager steps; the minibatch is written
    straight into the static input for some calls (no copy launch) and passed as another tensor for others."""
    from vmp_for_svae_amd.models import vae
    from vmp_for_svae_amd.training import SVAETrainer, GraphedSVAEStep
    N, K, Ld, U, Dy, S = 64, 10, 8, 50, 6, 10
    g = torch.Generator(device='cuda').manual_seed(77)
    ys = [torch.randn(N, Dy, device='cuda', generator=g) * 2 for _ in range(10)]

    def mk():
        vae.reset_variables()
        return SVAETrainer(K, Ld, U, Dy, nb_samples=S, lr=3e-3, lrcvi=0.2, decay_rate=0.5, stddev_init_nn=0.1, seed=11)
    tr = mk()
    want_elbo = [tr.step(y)['elbo'].item() for y in ys]
    want = [p.detach().clone() for p in tr.trainables()[1]] + [t.clone() for t in tr.theta]
    tr2 = mk()
    gs = GraphedSVAEStep(tr2, ys[0])
    assert gs.table_mode
    gs.TABLE_ROWS = 3                                   # (instance attribute: the table tensor keeps its 1024 rows, 3 are used)
    got_elbo = []
    for i, y in enumerate(ys):
        if i == 4:
            got_elbo.append(tr2.step(y)['elbo'].item())                 # eager step in between
        elif i % 2:
            gs.y.copy_(y)
            got_elbo.append(gs(gs.y)['elbo'].item())                    # minibatch already in the static input
        else:
            got_elbo.append(gs(y)['elbo'].item())
    assert tr2.global_step == 10 and tr2.opt.t == 10
    assert got_elbo == want_elbo
    for a, b in zip([p.detach() for p in tr2.trainables()[1]] + list(tr2.theta), want):
        assert torch.equal(a, b)


def test_graphed_step_across_the_scalar_table_boundary():
    """1 100 replayed steps (the table holds 1 024 rows: one refill on the way) and 278 replays of four steps (1 112 steps, refill with the
    table not used up to its last row) leave exactly the parameters of as many eager steps."""
    from vmp_for_svae_amd.models import vae
    from vmp_for_svae_amd.training import SVAETrainer, GraphedSVAEStep
    N, K, Ld, U, Dy, S = 32, 5, 4, 16, 3, 4
    g = torch.Generator(device='cuda').manual_seed(31)
    y = torch.randn(N, Dy, device='cuda', generator=g)

    def mk():
        vae.reset_variables()
        return SVAETrainer(K, Ld, U, Dy, nb_samples=S, lr=1e-3, lrcvi=0.2, decay_rate=0.95, stddev_init_nn=0.1, seed=2)
    for n, calls in ((1, 1100), (4, 278)):
        tr = mk()
        for _ in range(n * calls):
            tr.step(y)
        want = [p.detach().clone() for p in tr.trainables()[1]] + [t.clone() for t in tr.theta]
        tr2 = mk()
        gs = GraphedSVAEStep(tr2, y, steps_per_replay=n)
        src = gs.ys if n > 1 else gs.y
        for _ in range(calls):
            gs(src)
        assert tr2.global_step == n * calls and gs._table_base[0] > 0          # the table was rebuilt on the way
        for a, b in zip([p.detach() for p in tr2.trainables()[1]] + list(tr2.theta), want):
            assert torch.equal(a, b), n


def test_graphed_multi_step_replay_equals_eager_steps():
    """Round 6: GraphedSVAEStep(steps_per_replay=4) captures FOUR consecutive training steps in one graph (step i reads minibatch i of the
    static input and the next row of the scalar table): two replays == eight eager steps, every parameter / Adam slot / theta tensor and
    all eight ELBOs bit-identical; an eager step between the replays keeps the table aligned."""
    from vmp_for_svae_amd.models import vae
    from vmp_for_svae_amd.training import SVAETrainer, GraphedSVAEStep
    N, K, Ld, U, Dy, S = 64, 10, 8, 50, 6, 10
    g = torch.Generator(device='cuda').manual_seed(123)
    ys = torch.randn(9, N, Dy, device='cuda', generator=g) * 2

    def mk():
        vae.reset_variables()
        return SVAETrainer(K, Ld, U, Dy, nb_samples=S, lr=3e-3, lrcvi=0.2, decay_rate=0.9, stddev_init_nn=0.1, seed=5)
    tr = mk()
    want_elbo = [tr.step(ys[i])['elbo'].item() for i in range(9)]
    want = [p.detach().clone() for p in tr.trainables()[1]] + [t.clone() for t in tr.theta] + [t.clone() for t in tr.opt.m + tr.opt.v]
    tr2 = mk()
    gs = GraphedSVAEStep(tr2, ys[0], steps_per_replay=4)
    got = [o['elbo'].item() for o in gs(ys[0:4])]
    got.append(tr2.step(ys[4])['elbo'].item())                        # an eager step in between
    gs.ys.copy_(ys[5:9])
    outs = gs(gs.ys)                                                   # minibatches already in the static input
    got += [o['elbo'].item() for o in outs]
    assert [o['lrcvi'] for o in outs] == [0.2 * 0.9 ** ((5 + i) / 1000.0) for i in range(4)]
    assert tr2.global_step == 9 and tr2.opt.t == 9
    assert got == want_elbo
    have = [p.detach() for p in tr2.trainables()[1]] + list(tr2.theta) + tr2.opt.m + tr2.opt.v
    for a, b in zip(have, want):
        assert torch.equal(a, b)
    with pytest.raises(Exception):
        GraphedSVAEStep(SVAETrainer(K, Ld, U, Dy, nb_samples=S, rng='torch'), ys[0], steps_per_replay=2)   # no table mode: refused


def test_sample_x_per_comp_standalone_matches_fused(golden):
    """svae.sample_x_per_comp (svae.py:95-119) on the materialised phi_tilde reproduces the samples of the fused E-step
    kernel and the reference's x_k."""
    from vmp_for_svae_amd.models import svae
    g = golden('svae_tiny')
    tr, (N, K, Ld, S, Dy, U, steps) = make_trainer(g)
    from vmp_for_svae_amd.models import vae
    y, noise = dev(g['in_y']), dev(g['in_noise'][0])
    with torch.no_grad():
        phi_enc = vae.make_encoder(y, tr.encoder_layers, tr.stddev_init_nn)
        x_k, log_z, phi_tilde, _ = svae.e_step(phi_enc, tr.phi_gmm, S, noise=noise)
        x2 = svae.sample_x_per_comp(phi_tilde[0], phi_tilde[1], S, noise=noise)
    assert tuple(x2.shape) == (N, K, S, Ld)
    assert rel(x2, x_k.double().cpu().numpy()) < 1e-5
    assert rel(x2, g['step0_x_k']) < 1e-4


def test_elbo_debug_details_on_demand(golden):
    """details[1], details[2] of svae.compute_elbo (svae.py:256-260: sum r mean_s log-numerator / log-denominator),
    computed on access through the stand-alone density kernel, vs the reference run."""
    from vmp_for_svae_amd.models import svae
    g = golden('svae_paper')
    tr, (N, K, Ld, S, Dy, U, steps) = make_trainer(g)
    y, noise, zd = dev(g['in_y']), dev(g['in_noise'][0]), dev(g['in_zdraw'][0], torch.int64)
    elbo, details, x_k, x_s, log_z = tr.forward(y, noise, zd)
    det = g['step0_details']
    assert len(details) == 4
    rec, num, den, reg = details
    for got, want in ((rec, det[0]), (num, det[1]), (den, det[2]), (reg, det[3])):
        got = float(got.detach())
        assert abs(got - want) <= 3e-5 * max(abs(want), abs(det[0]) * 0.1), (got, want)


def test_driver_pinwheel_converges():
    """End-to-end (8f rank 3): the experiments driver on pinwheel (reference config: K=10, L=2, U=50, minibatch 100,
    lr 0.01, lrcvi 0.1).  The negative normalised ELBO must fall and the held-out log-likelihood must rise."""
    from vmp_for_svae_amd import experiments
    cfg = {'dataset': 'pinwheel', 'method': 'svae-cvi', 'lr': 0.01, 'lrcvi': 0.1, 'K': 10, 'L': 2, 'U': 50, 'seed': 0}
    tr, hist, log_id = experiments.run(cfg, nb_iters=400, measurement_freq=100, verbose=False)
    assert log_id.startswith('svae-cvi_pinwheel_K10')
    first, last = hist[0], hist[-1]
    assert all(np.isfinite(list(h.values())).all() for h in hist)
    assert last['neg_normed_elbo'] < first['neg_normed_elbo'] - 1.0
    assert last['loli'] > first['loli'] + 1.0
    assert last['mse'] < 0.5 * first['mse']
    assert 0.2 <= last['purity'] <= 1.0


def test_driver_multi_step_replays_equal_single_step_replays(tmp_path):
    """experiments.run(steps_per_replay=4): up to four consecutive iterations that no measurement / checkpoint iteration interrupts run
    from one graph replay - the same steps on the same minibatches: history, final parameters and the checkpoints written on the way
    are bit-identical to the one-step-per-replay run (pinwheel: 350 training rows, minibatches of 100 -> a ragged batch every epoch)."""
    from vmp_for_svae_amd import experiments
    cfg = {'dataset': 'pinwheel', 'method': 'svae-cvi', 'lr': 0.01, 'lrcvi': 0.1, 'K': 10, 'L': 2, 'U': 50, 'seed': 0}
    res = []
    for n in (1, 4):
        d = tmp_path / ('n%d' % n)
        tr, hist, log_id = experiments.run(cfg, nb_iters=43, measurement_freq=10, verbose=False, steps_per_replay=n,
                                           checkpoint_freq=15, checkpoint_dir=str(d))
        ck = {f: dict(np.load(os.path.join(str(d), f))) for f in sorted(os.listdir(str(d)))}
        res.append((hist, [p.detach().clone() for p in tr.trainables()[1]] + [t.clone() for t in tr.theta], ck))
    (h1, p1, c1), (h4, p4, c4) = res
    assert [sorted(h.items()) for h in h1] == [sorted(h.items()) for h in h4]
    for a, b in zip(p1, p4):
        assert torch.equal(a, b)
    assert sorted(c1) == sorted(c4) and len(c1) == 4
    for f in c1:
        for k in c1[f]:
            assert np.array_equal(c1[f][k], c4[f][k]), (f, k)


def test_predict_vs_oracle(golden):
    """svae.predict (reference svae.py:406-430) against the oracle's literal restatement with the same injected draws:
    reconstruction mean of the sub-sampled latent and the arg-max component; plus the seeded default path."""
    from oracle import nets, svae_ref
    from vmp_for_svae_amd.models import svae
    g = golden('svae_paper')
    tr, (N, K, Ld, S, Dy, U, steps) = make_trainer(g)
    y = dev(g['in_y'])
    gen = torch.Generator(device='cuda').manual_seed(5)
    noise = torch.randn(N, K, Ld, 1, device='cuda', generator=gen)
    zd = torch.randint(0, K, (N, 1), device='cuda', generator=gen)
    with torch.no_grad():
        y_mean, z_hat = svae.predict(y, tr.phi_gmm, tr.encoder_layers, tr.decoder_layers, seed=3, noise=noise, z_draws=zd)
    T = lambda a: torch.as_tensor(np.asarray(a)).double()
    enc_w = {v: T(g['in_w_encoder_net/' + v]) for v in NET_VARS}
    dec_w = {v: T(g['in_w_decoder_net/' + v]) for v in NET_VARS}
    phi = [p.detach().double().cpu() for p in tr.phi_gmm]
    phi_enc = nets.encoder(T(g['in_y']), enc_w)
    x_k, log_r, _, _ = svae_ref.e_step(phi_enc, phi, noise.double().cpu())
    x_s = svae_ref.subsample_x(x_k, zd.cpu())[:, 0, :]
    y_o, _ = nets.decoder(x_s, dec_w)
    e = rel(y_mean, y_o.numpy())
    parity_log.record('rel', e, 1e-5, 'predict y_mean')
    assert e <= 1e-5, e
    # arg-max: equal wherever the oracle's top two responsibilities are not within fp32 noise of each other
    top2 = torch.topk(log_r, 2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert clear.sum().item() > 0.9 * N
    assert (z_hat.cpu()[clear] == torch.argmax(log_r, dim=1)[clear]).all()
    # default draws: seeded, so two calls agree bit for bit and another seed gives another sample
    with torch.no_grad():
        a1, k1 = svae.predict(y, tr.phi_gmm, tr.encoder_layers, tr.decoder_layers, seed=7)
        a2, k2 = svae.predict(y, tr.phi_gmm, tr.encoder_layers, tr.decoder_layers, seed=7)
        a3, _ = svae.predict(y, tr.phi_gmm, tr.encoder_layers, tr.decoder_layers, seed=8)
    assert torch.equal(a1, a2) and torch.equal(k1, k2) and torch.equal(k1, z_hat)
    assert not torch.equal(a1, a3)
    assert tuple(a1.shape) == (N, Dy) and k1.dtype == torch.int64


def test_standalone_densities_are_differentiable():
    """Adjoints of the two stand-alone per-sample densities (vmp_gauss_logprob_nat_per_samp_bwd, vmp_student_t_logprob_bwd)
    against the oracle's fp64 autograd of the literal formulas (gaussian.py:74-105, student_t.py:7-39).  eta2 enters
    through a symmetric parametrisation (as every caller builds it), so the comparison is on the leaf gradients."""
    from oracle import dists
    from vmp_for_svae_amd.distributions import gaussian, student_t
    rng = np.random.Generator(np.random.PCG64(21))
    for (N, K, S, D) in [(7, 3, 4, 2), (33, 10, 10, 6), (64, 16, 10, 8), (5, 5, 3, 1), (9, 7, 5, 5)]:
        x = rng.standard_normal((N, K, S, D))
        e1 = rng.standard_normal((N, K, D))
        A = rng.standard_normal((N, K, D, D)) * 0.3
        g = rng.standard_normal((N, K, S))

        def build(t64, dev):
            cast = (lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)) if t64 else \
                   (lambda a: torch.tensor(a, dtype=torch.float32, device='cuda', requires_grad=True))
            xs, e1s, As = cast(x), cast(e1), cast(A)
            eye = torch.eye(D, dtype=As.dtype, device=As.device)
            e2 = -0.5 * (As @ As.transpose(-1, -2) + 0.5 * eye)            # symmetric negative definite
            return xs, e1s, As, e2
        xo, e1o, Ao, e2o = build(True, None)
        lo = dists.gauss_log_probability_nat_per_samp(xo, e1o, e2o)
        go = torch.autograd.grad((lo * torch.tensor(g)).sum(), [xo, e1o, Ao])
        xp, e1p, Ap, e2p = build(False, 'cuda')
        lp = gaussian.log_probability_nat_per_samp(xp, e1p, e2p)
        gp = torch.autograd.grad((lp * torch.tensor(g, dtype=torch.float32, device='cuda')).sum(), [xp, e1p, Ap])
        tag = (N, K, S, D)
        assert rel(lp, lo.detach().numpy()) < 2e-5, tag
        for n_, a_, b_ in zip(('x', 'eta1', 'A(eta2)'), gp, go):
            e = rel(a_, b_.numpy())
            parity_log.record('rel', e, 5e-5, 'gauss per-samp grad ' + n_)
            assert e < 5e-5, (tag, n_, e)
        # Student-t: gradients to y, mu and the scale matrices (through the K-sized Cholesky / inverse in torch)
        mu = rng.standard_normal((K, D)) * 2
        B = rng.standard_normal((K, D, D)) * 0.4
        v = 3.0 + rng.random(K) * 4

        def build_t(t64):
            cast = (lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)) if t64 else \
                   (lambda a: torch.tensor(a, dtype=torch.float32, device='cuda', requires_grad=True))
            ys, mus, Bs = cast(x), cast(mu), cast(B)
            eye = torch.eye(D, dtype=Bs.dtype, device=Bs.device)
            sig = Bs @ Bs.transpose(-1, -2) + 0.5 * eye
            vs = torch.tensor(v, dtype=Bs.dtype, device=Bs.device)
            return ys, mus, Bs, sig, vs
        yo, muo, Bo, sigo, vo = build_t(True)
        to = dists.student_t_log_probability_per_samp(yo, muo, sigo, vo)
        gto = torch.autograd.grad((to * torch.tensor(g)).sum(), [yo, muo, Bo])
        yp, mup, Bp, sigp, vp = build_t(False)
        tp = student_t.log_probability_per_samp(yp, mup, sigp, vp)
        gtp = torch.autograd.grad((tp * torch.tensor(g, dtype=torch.float32, device='cuda')).sum(), [yp, mup, Bp])
        assert rel(tp, to.detach().numpy()) < 2e-5, tag
        for n_, a_, b_ in zip(('y', 'mu', 'B(sigma)'), gtp, gto):
            e = rel(a_, b_.numpy())
            parity_log.record('rel', e, 5e-5, 'student-t grad ' + n_)
            assert e < 5e-5, (tag, n_, e)


@pytest.mark.parametrize('case', ['svae_paper', 'svae_smm_l8'])
def test_compute_elbo_on_other_samples_than_the_e_steps(golden, case):
    """compute_elbo(_smm) evaluated at samples that are NOT the tensor e_step returned (a detached copy with its own
    gradient): the reference's literal formulation on the stand-alone differentiable densities must give the same value
    and the same gradient w.r.t. the samples as the oracle (svae.py:229-252 / 288-300)."""
    from oracle import nets, svae_ref
    from vmp_for_svae_amd.models import svae
    g = golden(case)
    tr, (N, K, Ld, S, Dy, U, steps) = make_trainer(g)
    y = dev(g['in_y'])
    noise = dev(g['in_noise'][0])
    y_rec, phi_enc, x_k, x_s, log_z, _, phi_tilde = svae.inference(y, tr.phi_gmm, tr.encoder_layers, tr.decoder_layers, S,
                                                                   noise=noise, z_draws=dev(g['in_zdraw'][0], torch.int64))
    x2 = (x_k.detach() * 1.01 + 0.02).requires_grad_(True)                  # other samples, own leaf
    fn = svae.compute_elbo_smm if tr.smm else svae.compute_elbo
    elbo, details = fn(y, y_rec, tr.theta, phi_tilde, x2, log_z, 'standard')
    gx, = torch.autograd.grad(-details[3], [x2])
    # oracle, fp64, same inputs
    T = lambda a: torch.as_tensor(np.asarray(a)).double()
    enc_w = {v: T(g['in_w_encoder_net/' + v]) for v in NET_VARS}
    phi = [p.detach().double().cpu() for p in tr.phi_gmm]
    pe = nets.encoder(T(g['in_y']), enc_w)
    xk_o, lz_o, pt_o, _ = svae_ref.e_step(pe, phi, T(g['in_noise'][0]))
    x2o = (xk_o.detach() * 1.01 + 0.02).requires_grad_(True)
    th = [t.detach().double().cpu() for t in tr.theta]
    rec_dummy = (torch.zeros(N, K, S, Dy, dtype=torch.float64), torch.ones(N, K, S, Dy, dtype=torch.float64))
    fo = svae_ref.compute_elbo_smm if tr.smm else svae_ref.compute_elbo
    _, det_o = fo(T(g['in_y']), rec_dummy, th, pt_o, x2o, lz_o)
    gxo, = torch.autograd.grad(-det_o[3], [x2o])
    e = abs(details[3].item() - det_o[3].item()) / abs(det_o[3].item())
    parity_log.record('rel', e, 2e-5, 'regulariser at foreign samples')
    assert e <= 2e-5, e
    eg = rel(gx, gxo.numpy())
    parity_log.record('rel', eg, 1e-4, 'd regulariser / d samples')
    assert eg <= 1e-4, eg


@pytest.mark.parametrize('N,K,S,Ld', [(1000, 16, 10, 8), (777, 10, 4, 6), (50, 3, 2, 1), (130, 64, 3, 4), (9, 33, 5, 7)])
def test_subsample_inverse_cdf(N, K, S, Ld):
    """svae.subsample_x with uniforms (svae.py:122-151: z ~ Cat(exp log_z), x[n, z, s]): the lane-per-cell kernel's draw equals
    a numpy inverse CDF wherever u is not within fp32 rounding of a CDF step, and the gathered rows are exact copies."""
    from vmp_for_svae_amd.models import svae
    rng = np.random.Generator(np.random.PCG64(N + K))
    lz = np.log(rng.dirichlet(np.ones(K) * 0.5, size=N)).astype(np.float32)
    x = rng.standard_normal((N, K, S, Ld)).astype(np.float32)
    u = rng.random((N, S)).astype(np.float32)
    out = svae.subsample_x(dev(x), dev(lz), u=dev(u)).cpu().numpy()
    cdf = np.cumsum(np.exp(lz.astype(np.float64)), axis=1)
    z = np.minimum((cdf[:, None, :-1] <= u[:, :, None]).sum(-1), K - 1)            # (N,S)
    gap = np.abs(cdf[:, None, :] - u[:, :, None]).min(-1)
    clear = gap > 1e-5
    want = x[np.arange(N)[:, None], z, np.arange(S)[None, :]]                    # (N,S,L)
    assert clear.mean() > 0.99
    assert np.array_equal(out[clear], want[clear])
    # every output row is SOME component's sample row of the right (n, s)
    hit = (out[:, None, :, :] == x).all(-1).any(1)
    assert hit.all()
    # supplied indices: exact gather
    zi = rng.integers(0, K, size=(N, S))
    out2 = svae.subsample_x(dev(x), dev(lz), z_draws=dev(zi, torch.int64)).cpu().numpy()
    assert np.array_equal(out2, x[np.arange(N)[:, None], zi, np.arange(S)[None, :]])
