"""The oracle (oracle/*.py) against the golden vectors produced by executing the reference's own
functions (tests/golden/make_fixtures.py), plus independent known-answer checks.  CPU only."""
import os

import numpy as np
import pytest
import scipy.special
import scipy.stats
import torch

from oracle import dists, metrics, mixtures, nets, svae_ref, train_ref

F64_RTOL = 1e-9
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def T(a, dtype=torch.float64):
    return torch.as_tensor(np.asarray(a)).to(dtype) if np.asarray(a).dtype.kind == 'f' else torch.as_tensor(np.asarray(a))


def close(got, want, rtol=F64_RTOL, atol=None, what=''):
    got = got.detach().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = np.asarray(want)
    scale = max(np.abs(want).max(), 1e-300)
    err = np.abs(got - want).max() / scale if atol is None else np.abs(got - want).max()
    tol = rtol if atol is None else atol
    assert err <= tol, '%s: err %.3e > %.1e' % (what, err, tol)


@pytest.mark.parametrize('case', ['dist_tiny', 'dist_l8'])
@pytest.mark.parametrize('dtype,suf,rtol', [(torch.float64, '', 1e-9), (torch.float32, '__f32', 2e-4)])
def test_distributions(golden, case, dtype, suf, rtol):
    g = golden(case)
    i = {k[3:]: T(g[k], dtype) for k in g.files if k.startswith('in_')}
    e1, e2 = dists.gauss_standard_to_natural(i['mu'], i['sigma'])
    close(e1, g['s2n_eta1' + suf], rtol, what='s2n eta1')
    close(e2, g['s2n_eta2' + suf], rtol, what='s2n eta2')
    mu2, sg2 = dists.gauss_natural_to_standard(e1, e2)
    close(mu2, g['n2s_mu' + suf], rtol), close(sg2, g['n2s_sigma' + suf], rtol)
    close(dists.gauss_log_probability_nat(i['x'], i['eta1_nk'], i['eta2_nk'], i['w']), g['logprob_nat' + suf], rtol)
    close(dists.gauss_log_probability_nat(i['x'], i['eta1_nk'], i['eta2_nk']), g['logprob_nat_now' + suf], rtol)
    close(dists.gauss_log_probability_nat_per_samp(i['xs'], i['eta1_nk'], i['eta2_nk']), g['logprob_per_samp' + suf], rtol)
    close(dists.logdet(i['sigma']), g['logdet' + suf], rtol)
    em, eC = dists.niw_expected_values(i['beta'], i['m'], i['C'], i['v'])
    close(em, g['niw_exp_m' + suf], rtol), close(eC, g['niw_exp_C' + suf], rtol)
    A, b, be, vh = dists.niw_standard_to_natural(i['beta'], i['m'], i['C'], i['v'])
    close(A, g['niw_A' + suf], rtol), close(b, g['niw_b' + suf], rtol), close(vh, g['niw_vhat' + suf], rtol)
    _, m2, C2, v2 = dists.niw_natural_to_standard(A, b, be, vh)
    close(m2, g['niw_back_m' + suf], rtol), close(C2, g['niw_back_C' + suf], rtol), close(v2, g['niw_back_v' + suf], rtol)
    close(dists.dir_expected_log_pi(i['alpha']), g['dir_elogpi' + suf], rtol)
    close(dists.dir_standard_to_natural(i['alpha']), g['dir_nat' + suf], rtol)
    close(dists.dir_natural_to_standard(i['alpha']), g['dir_std' + suf], rtol)
    close(dists.student_t_log_probability_per_samp(i['xs'], i['mu'], i['sigma'], i['dof']), g['student_t' + suf], rtol)


def test_known_answers(golden):
    """Independent of the reference: scipy densities / special functions."""
    g = golden('dist_tiny')
    xs, mu, sigma, dof = g['in_xs'], g['in_mu'], g['in_sigma'], g['in_dof']
    N, K, S, L = xs.shape
    got = dists.student_t_log_probability_per_samp(T(xs), T(mu), T(sigma), T(dof)).numpy()
    for k in range(K):
        want = scipy.stats.multivariate_t(loc=mu[k], shape=sigma[k], df=dof[k]).logpdf(xs[:, k].reshape(-1, L))
        assert np.abs(got[:, k].reshape(-1) - want).max() < 1e-12
    e1, e2 = g['in_eta1_nk'], g['in_eta2_nk']
    got = dists.gauss_log_probability_nat_per_samp(T(xs), T(e1), T(e2)).numpy()
    for n in range(N):
        for k in range(K):
            Sg = np.linalg.inv(-2 * e2[n, k])
            want = scipy.stats.multivariate_normal(Sg @ e1[n, k], Sg).logpdf(xs[n, k])
            assert np.abs(got[n, k] - want).max() < 1e-11
    a = g['in_alpha']
    assert np.abs(dists.dir_expected_log_pi(T(a)).numpy() - (scipy.special.digamma(a) - scipy.special.digamma(a.sum()))).max() < 1e-13
    lz = dists.gauss_log_probability_nat(T(g['in_x']), T(e1), T(e2), T(g['in_w']))
    assert np.abs(np.exp(lz.numpy()).sum(1) - 1).max() < 1e-12


@pytest.mark.parametrize('case', ['gmm_tiny', 'gmm_d6k10', 'gmm_d8k16'])
@pytest.mark.parametrize('dtype,suf,rtol,atol_r', [(torch.float64, '', 1e-9, 1e-10), (torch.float32, '__f32', 5e-4, 2e-4)])
def test_gmm_smm_steps(golden, case, dtype, suf, rtol, atol_r):
    g = golden(case)
    x, r = T(g['in_x'], dtype), T(g['in_r0'], dtype)
    for it in range(3):
        r, log_r, theta, (xk, Sk, pi) = mixtures.gmm_inference_step(x, r)
        for n_, t_ in zip(('alpha', 'beta', 'm', 'C', 'v'), theta):
            close(t_, g['gmm%d_%s%s' % (it, n_, suf)], rtol, what='gmm%d %s' % (it, n_))
        close(xk, g['gmm%d_xk%s' % (it, suf)], rtol), close(Sk, g['gmm%d_Sk%s' % (it, suf)], rtol)
        close(pi, g['gmm%d_pi%s' % (it, suf)], rtol)
        close(r, g['gmm%d_r%s' % (it, suf)], atol=atol_r, what='gmm r')
    # stand-alone pieces on step-0 quantities
    prior = mixtures.vmp_prior(x.shape[1] and g['in_r0'].shape[1], x.shape[1], dtype)
    r0 = T(g['in_r0'], dtype)
    ak, bk, mk, Ck, vk, _, _ = mixtures.gmm_m_step(x, r0, *prior)
    Pk = dists.inv(Ck)
    close(Pk, g['P0' + suf], rtol * 10)
    close(mixtures.gmm_expct_log_det_prec(vk, Pk), g['elogdet0' + suf], rtol * 10)
    close(mixtures.gmm_expct_mahalanobis(x, bk, mk, Pk, vk), g['maha0' + suf], rtol * 10)
    rm, pim = mixtures.gmm_e_step(x, ak, bk, mk, Pk, vk, torch.as_tensor(g['in_miss']))
    close(rm, g['miss_r' + suf], atol=atol_r), close(pim, g['miss_pi' + suf], rtol * 10)
    # SMM
    r, u = r0, torch.ones_like(r0)
    for it in range(3):
        r, u, theta, (xk, Sk, pi) = mixtures.smm_inference_step(x, r, u, float(g['in_kappa']))
        for n_, t_ in zip(('alpha', 'beta', 'm', 'C', 'v'), theta):
            close(t_, g['smm%d_%s%s' % (it, n_, suf)], rtol * 10, what='smm%d %s' % (it, n_))
        close(r, g['smm%d_r%s' % (it, suf)], atol=atol_r * 5, what='smm r')
        close(u, g['smm%d_u%s' % (it, suf)], rtol * 50, what='smm u')


def _svae_state(g, dtype, suf=''):
    N, K, L, S, Dy, U, steps, smm = [int(v) for v in g['in_dims']]
    enc = {v: T(g['in_w_encoder_net/' + v], dtype) for v in nets.NET_VARS}
    dec = {v: T(g['in_w_decoder_net/' + v], dtype) for v in nets.NET_VARS}
    prior, theta = svae_ref.init_mm(K, L, T(g['in_m_unif'], dtype), dtype)
    phi = svae_ref.init_recognition_params(theta, T(g['in_pi_norm'], dtype))
    phi[1] = phi[1] + T(g['in_Lk_low'], dtype)
    if smm:
        mu_k, L_k = svae_ref.make_loc_scale(prior)
        theta = [theta[0], mu_k, L_k, torch.full((K,), float(g['in_dof0']), dtype=dtype)]
        prior = prior[0]
    return train_ref.State(phi, enc, dec, theta, prior, smm=bool(smm)), (N, K, L, S, Dy, U, steps, smm)


@pytest.mark.parametrize('case', ['svae_tiny', 'svae_paper', 'svae_c1', 'svae_l8', 'svae_smm_tiny', 'svae_smm_l8', 'svae_auto'])
def test_svae_init_matches_reference(golden, case):
    g = golden(case)
    st, dims = _svae_state(g, torch.float64)
    for n_, t_ in zip(('mu_k', 'L_k', 'log_pi_k'), st.phi_gmm):
        close(t_, g['phi_init_' + n_], what='phi ' + n_)
    if dims[-1]:
        close(st.gmm_prior, g['prior_alpha'])
        for n_, t_ in zip(('alpha', 'mu', 'L', 'dof'), st.theta):
            close(t_, g['theta_init_' + n_], what='theta ' + n_)
    else:
        for n_, p_, t_ in zip(('alpha', 'A', 'b', 'beta', 'vhat'), st.gmm_prior, st.theta):
            close(p_, g['prior_' + n_]), close(t_, g['theta_init_' + n_])


@pytest.mark.parametrize('case', ['svae_tiny', 'svae_paper', 'svae_c1', 'svae_l8', 'svae_smm_tiny', 'svae_smm_l8', 'svae_auto'])
@pytest.mark.parametrize('dtype,suf,rtol', [(torch.float64, '', 1e-8), (torch.float32, '__f32', 2e-3)])
def test_svae_training_steps(golden, case, dtype, suf, rtol):
    """Full training steps (inference, ELBO, 21/23 gradients, CVI update, TF-Adam) vs the reference run."""
    g = golden(case)
    st, (N, K, L, S, Dy, U, steps, smm) = _svae_state(g, dtype)
    y = T(g['in_y'], dtype)
    for it in range(steps):
        pre = 'step%d_' % it
        noise, zd = T(g['in_noise'][it], dtype), T(g['in_zdraw'][it])
        # forward pieces first (before the step mutates the state)
        y_rec, phi_enc, x_k, x_s, log_z, _, phi_tilde = svae_ref.inference(y, st.phi_gmm, st.enc_w, st.dec_w, noise, zd)
        close(phi_enc[0], g[pre + 'enc_eta1' + suf], rtol), close(phi_enc[1], g[pre + 'enc_eta2' + suf], rtol)
        if pre + 'x_k' + suf in g.files:                      # slim fixtures (svae_auto) hold x_k / x_s in fp64 only
            close(x_k, g[pre + 'x_k' + suf], rtol), close(x_s, g[pre + 'x_s' + suf], rtol)
        close(torch.exp(log_z), np.exp(g[pre + 'log_z' + suf]), atol=max(rtol * 0.1, 1e-10), what='r_nk')
        if pre + 'rec_mean' + suf in g.files:
            close(y_rec[0], g[pre + 'rec_mean' + suf], rtol), close(y_rec[1], g[pre + 'rec_var' + suf], rtol)
            close(phi_tilde[0], g[pre + 'phi_tilde_eta1' + suf], rtol), close(phi_tilde[1], g[pre + 'phi_tilde_eta2' + suf], rtol)
        out = train_ref.train_step(st, y, noise, zd, float(g['in_lr']), float(g['in_lrcvi']), float(g['in_decay']))
        close(out['elbo'], g[pre + 'elbo' + suf], rtol, what='elbo')
        close(out['details'], g[pre + 'details' + suf], rtol, what='details')
        assert abs(out['lrcvi'] - float(g[pre + 'lrcvi'])) < 1e-15
        for n_, gr in out['grads'].items():
            close(gr, g[pre + 'grad_' + n_ + suf], rtol * 5, what='grad ' + n_)
        names, params = st.trainables()
        for n_, p in zip(names, params):
            close(p, g[pre + 'param_' + n_ + suf], rtol, what='param ' + n_)
        if smm:
            close(out['theta_star'][0], g[pre + 'theta_star_alpha' + suf], rtol)
            close(st.theta[0], g[pre + 'theta_alpha' + suf], rtol)
        else:
            for n_, ts, t_ in zip(('alpha', 'A', 'b', 'beta', 'vhat'), out['theta_star'], st.theta):
                close(ts, g[pre + 'theta_star_' + n_ + suf], rtol, what='theta* ' + n_)
                close(t_, g[pre + 'theta_' + n_ + suf], rtol, what='theta ' + n_)


def test_auto_fixture_is_the_auto_minibatch():
    """BASELINE configs[3]: svae_auto trains on 64 rows of the Auto training split the reference's loader produced."""
    g, d = np.load(os.path.join(GOLDEN_DIR, 'svae_auto.npz')), np.load(os.path.join(GOLDEN_DIR, 'datasets.npz'))
    assert [int(v) for v in g['in_dims'][:6]] == [64, 10, 8, 10, 6, 50]
    X = d['auto_X_tr'].astype(np.float32)
    assert all((X == row.astype(np.float32)).all(1).any() for row in g['in_y'])


@pytest.mark.parametrize('chunk', [100, 257])
def test_chunked_mixture_steps_equal_the_literal_ones(golden, chunk):
    """The N-chunked evaluation used for the N=1e6 GPU parity tests is the same two-pass update."""
    g = golden('gmm_d8k16')
    x, r = T(g['in_x'], torch.float64), T(g['in_r0'], torch.float64)
    u = torch.ones_like(r)
    for it in range(3):
        a, b = mixtures.gmm_inference_step(x, r), mixtures.gmm_inference_step_chunked(x, r, chunk)
        close(b[0], a[0].numpy(), atol=1e-11)
        for p, q in zip(a[2], b[2]):
            close(q, p.numpy(), 1e-12)
        close(b[0], g['gmm%d_r' % it], atol=1e-10)
        r = a[0]
    r = T(g['in_r0'], torch.float64)
    for it in range(3):
        a, b = mixtures.smm_inference_step(x, r, u, 5.0), mixtures.smm_inference_step_chunked(x, r, u, 5.0, chunk)
        close(b[0], a[0].numpy(), atol=1e-11), close(b[1], a[1].numpy(), 1e-12)
        close(b[0], g['smm%d_r' % it], atol=1e-10)
        r, u = a[0], a[1]


def test_towers_average_gradients(golden):
    """experiments.py:196-265 semantics: G towers -> mean of per-tower gradients, summed ELBO, M-step on the
    concatenation.  (No multi-tower golden exists: the reference needs real GPUs for nb_gpu > 1.)"""
    g = golden('svae_paper')
    st1, (N, K, L, S, Dy, U, steps, smm) = _svae_state(g, torch.float64)
    st2, _ = _svae_state(g, torch.float64)
    y, noise, zd = T(g['in_y']), T(g['in_noise'][0]), T(g['in_zdraw'][0])
    o1 = train_ref.train_step(st1, y, noise, zd, 3e-4, 0.2, 0.95, towers=1)
    o2 = train_ref.train_step(st2, y, noise, zd, 3e-4, 0.2, 0.95, towers=2)
    close(o2['elbo'], o1['elbo'].numpy(), 1e-12)
    for n_ in o1['grads']:
        close(o2['grads'][n_] * 2, o1['grads'][n_].numpy(), 1e-9, what=n_)
    for a, b in zip(st1.theta, st2.theta):
        close(b, a.numpy(), 1e-12)


def test_thread_pool_does_not_change_the_chunked_oracles(golden):
    """The full-size GPU tests run the oracle's row chunks / towers several at a time (workers=8): independent graphs, combined in
    chunk order - the results must be bit-identical to the sequential evaluation."""
    g = golden('svae_paper')
    st1, (N, K, L, S, Dy, U, steps, smm) = _svae_state(g, torch.float64)
    st2, _ = _svae_state(g, torch.float64)
    y, noise, zd = T(g['in_y']), T(g['in_noise'][0]), T(g['in_zdraw'][0])
    o1 = train_ref.train_step(st1, y, noise, zd, 3e-4, 0.2, 0.95, towers=2)
    o2 = train_ref.train_step(st2, y, noise, zd, 3e-4, 0.2, 0.95, towers=2, workers=2)
    assert torch.equal(o1['elbo'], o2['elbo']) and all(torch.equal(o1['grads'][n_], o2['grads'][n_]) for n_ in o1['grads'])
    rng = np.random.Generator(np.random.PCG64(3))
    Nn = 50
    prior, theta = svae_ref.init_mm(K, L, T(rng.random((K, L))), torch.float64)
    phi = svae_ref.init_recognition_params(theta, T(rng.standard_normal(K)))
    e1, e2 = T(rng.standard_normal((Nn, L))), -0.5 * torch.nn.functional.softplus(T(rng.standard_normal((Nn, L))))
    nz, zz = T(rng.standard_normal((Nn, K, L, S))), torch.as_tensor(rng.integers(0, K, size=(Nn, S)))
    Gx, Glz = T(rng.standard_normal((Nn, K, S, L))), T(rng.standard_normal((Nn, K)))
    a = train_ref.vmp_step_t2(phi, theta, prior, e1, e2, nz, zz, Gx, Glz, 0.2, chunk=16)
    b = train_ref.vmp_step_t2(phi, theta, prior, e1, e2, nz, zz, Gx, Glz, 0.2, chunk=16, workers=3)
    assert torch.equal(a['reg'], b['reg']) and torch.equal(a['g_eta1'], b['g_eta1']) and torch.equal(a['x_samples'], b['x_samples'])
    assert all(torch.equal(x_, y_) for x_, y_ in zip(a['g_phi'] + a['theta_new'], b['g_phi'] + b['theta_new']))


@pytest.mark.parametrize('case', ['metrics', 'metrics_s100'])
@pytest.mark.parametrize('dtype,suf,rtol', [(torch.float64, '', 1e-12), (torch.float32, '__f32', 1e-4)])
def test_metrics(golden, case, dtype, suf, rtol):
    g = golden(case)
    y, mean, var, lw, lws = [T(g['in_' + k], dtype) for k in ('y', 'mean', 'var', 'lw', 'lws')]
    r = torch.exp(lw)
    close(metrics.weighted_mse(y, mean, r), g['weighted_mse' + suf], rtol)
    close(metrics.diagonal_gaussian_logprob(y, mean, var, lw), g['loli' + suf], rtol)
    close(metrics.diagonal_gaussian_logprob(y, mean, var, lws), g['loli_s' + suf], rtol)
    close(metrics.diagonal_gaussian_logprob(y, mean, var, lw, mask=T(g['in_mask'])), g['loli_mask' + suf], rtol)
    e, p_ = metrics.purity(r, T(g['in_labels'], dtype))
    close(e, g['entropy' + suf], rtol), close(p_, g['purity' + suf], rtol)


def _toy_impute(g, dtype):
    """The deterministic imputation method of tests/golden/make_fixtures.py::imputation_toy_method (data only)."""
    a_ks, b_ksd, Wr = [T(g['in_' + k], dtype) for k in ('a_ks', 'b_ksd', 'Wr')]

    def method(y_pert):
        mean = y_pert[:, None, None, :] * a_ks[None, :, :, None] + b_ksd[None]
        var = 0.3 + 0.5 * (y_pert[:, None, None, :] * a_ks[None, :, :, None]) ** 2
        logits = y_pert @ Wr
        return mean, var, logits - torch.logsumexp(logits, dim=1, keepdim=True)
    return method


@pytest.mark.parametrize('dtype,suf,rtol', [(torch.float64, '', 1e-12), (torch.float32, '__f32', 1e-4)])
def test_imputation(golden, dtype, suf, rtol):
    """SURVEY 8f rank 2 (losses.py:148-310) vs the reference run."""
    g = golden('imputation')
    N, K, S, D, P = [int(v) for v in g['in_dims']]
    y, noise = T(g['in_y'], dtype), T(g['in_noise'], dtype)
    mask = torch.as_tensor(g['mask'].astype(bool))
    assert int(mask.sum()) == int(N * D * 0.3)
    close(metrics.perturb_data(y, mask, noise[0]), g['perturbed0' + suf], rtol)
    close(metrics.imputation_mse(y, T(g['in_y_pred'], dtype), T(g['in_r'], dtype), mask), g['imputation_mse' + suf], rtol)
    mse, ll = metrics.imputation_losses(y, mask, _toy_impute(g, dtype), noise, S)
    close(mse, g['imp_mse' + suf], rtol), close(ll, g['imp_loglike' + suf], rtol)


def _vb_weights(g, prefix, dtype):
    return {k[len('in_w_' + prefix):]: T(g[k], dtype) for k in g.files if k.startswith('in_w_' + prefix)}


@pytest.mark.parametrize('dtype,suf,rtol', [(torch.float64, '', 1e-10), (torch.float32, '__f32', 2e-4)])
def test_bernoulli_and_plain_vae(golden, dtype, suf, rtol):
    """SURVEY 8f rank 4 (vae.py:53-55,138-198,253-296; losses.py:41-80) vs the reference run."""
    g = golden('vae_bernoulli')
    N, K, S, Ld, D, U = [int(v) for v in g['in_dims']]
    yb, yr, x4, lw, lws = [T(g['in_' + k], dtype) for k in ('y_bin', 'y_real', 'x4', 'lw', 'lws')]
    mask = torch.as_tensor(g['in_mask'])
    probas, logits = nets.decoder_bernoulli(x4, _vb_weights(g, 'dec_bernoulli/', dtype))
    close(probas, g['probas' + suf], rtol), close(logits, g['logits' + suf], rtol)
    close(nets.expected_bernoulli_loglike(yb, logits, torch.exp(lw)), g['ebl_weighted' + suf], rtol)
    close(nets.expected_bernoulli_loglike(yb, logits[:, 0]), g['ebl_plain' + suf], rtol)
    close(metrics.bernoulli_logprob(yb, logits[:, 0]), g['blp_plain' + suf], rtol)
    close(metrics.bernoulli_logprob(yb, logits, lw), g['blp_w' + suf], rtol)
    close(metrics.bernoulli_logprob(yb, logits, lws, mask), g['blp_ws_mask' + suf], rtol)
    pert = torch.where(mask, torch.where(T(g['in_unif_pert'], dtype) < 0.5, 1.0, -1.0).to(dtype) * 0 +
                       torch.where(T(g['in_unif_pert'], dtype) < 0.5, torch.ones_like(yb), -torch.ones_like(yb)), yb)
    close(pert, g['perturbed_bern' + suf], rtol)
    for head, y in (('bernoulli', yb), ('standard', yr)):
        ew = {k: v.clone().requires_grad_(True) for k, v in _vb_weights(g, 'enc/', dtype).items()}
        dw = {k: v.clone().requires_grad_(True) for k, v in _vb_weights(g, 'dec_%s/' % head, dtype).items()}
        mu, var = nets.mlp(y, ew, 'standard')
        xs = nets.reparam_trick_sampling(mu, var, T(g['in_noise_rep'], dtype))
        dec = nets.decoder_bernoulli(xs, dw) if head == 'bernoulli' else nets.decoder(xs, dw)
        elbo = nets.vae_compute_elbo(y, mu, var, dec, head)
        close(xs, g['vae_%s_x%s' % (head, suf)], rtol)
        close(nets.kl_divergence(mu, var), g['vae_%s_kl%s' % (head, suf)], rtol)
        close(elbo, g['vae_%s_elbo%s' % (head, suf)], rtol)
        names = [('encoder_net/' + k, v) for k, v in ew.items()] + [('decoder_net/' + k, v) for k, v in dw.items()]
        grads = torch.autograd.grad(-elbo, [v for _, v in names])
        for (n_, _), gr in zip(names, grads):
            close(gr, g['vae_%s_grad_%s%s' % (head, n_, suf)], 5 * rtol, what=n_)
