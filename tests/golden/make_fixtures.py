#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN FUNCTIONS.

Run in the authoring container only (needs /root/reference; the GPU box never has it):

    python tests/golden/make_fixtures.py            # rewrites tests/golden/*.npz

How: TensorFlow 1.3 is not installable here, so ``tf1_shim/tensorflow`` (an eager, torch-CPU backed
stand-in for the ~90 tf.* calls the hot path uses) is put first on sys.path and the reference's
unmodified ``distributions/*``, ``models/{gmm,smm,svae,vae}`` and ``helpers/tf_utils`` are imported from
/root/reference on top of it.  Every case is evaluated twice on identical (fp32-representable) inputs:
with tf.float32 -> fp64 ("truth", key ``name``) and -> fp32 (the reference's own arithmetic, key
``name__f32``).  All random draws (noise, categorical draws, initial responsibilities, MLP weights) are
generated here with numpy PCG64 and INJECTED, and are stored in the fixture next to the outputs.

Only data is written: inputs and expected outputs.  No reference source text is stored.
The training-step fixtures use the reference's functions for everything except Adam and
exponential_decay (TensorFlow internals, not in the reference tree) which are restated here from the
TF-1.3 documentation: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; var -= lr_t*m/(sqrt(v)+eps).
"""
import collections
import collections.abc
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('VMP_REFERENCE', '/root/reference')


def _install():
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(HERE, 'tf1_shim'))
    # stubs for things the reference imports at module import time but the hot path never uses
    if not hasattr(np, 'int'):
        np.int = int
    collections.Iterable = collections.abc.Iterable
    import matplotlib
    matplotlib.use('Agg')
    for name in ('tensorboard', 'tensorboard.backend', 'tensorboard.backend.event_processing',
                 'tensorboard.backend.event_processing.event_accumulator'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['tensorboard.backend.event_processing.event_accumulator'].EventAccumulator = object
    import tensorflow as tf
    from distributions import gaussian, niw, dirichlet, student_t
    from helpers import tf_utils
    from models import gmm, smm, svae, vae
    import losses

    class _NP(object):          # gmm.py:66,76 / smm.py:70,85 apply np.multiply/np.divide to tensors
        def __getattr__(self, k):
            return getattr(np, k)
        multiply = staticmethod(lambda a, b: tf._T(a) * tf._T(b))
        divide = staticmethod(lambda a, b: tf._T(a) / tf._T(b))
    gmm.np = _NP()
    smm.np = _NP()
    return types.SimpleNamespace(tf=tf, gaussian=gaussian, niw=niw, dirichlet=dirichlet, student_t=student_t,
                                 tf_utils=tf_utils, gmm=gmm, smm=smm, svae=svae, vae=vae, losses=losses)


R = _install()
tf = R.tf


def f32(a):
    """round to fp32-representable values, keep as float64 ndarray"""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def T(a, grad=False):
    t = tf._T(torch.as_tensor(np.asarray(a)).to(torch.get_default_dtype()))
    if grad:
        t = t.detach().clone().as_subclass(tf.Tensor).requires_grad_(True)
    return t


def npy(t):
    if isinstance(t, (list, tuple)):
        return [npy(v) for v in t]
    return t.detach().cpu().numpy().copy() if isinstance(t, torch.Tensor) else np.array(t)


def both(fn):
    """run fn() under fp64 and fp32; merge dicts as name / name__f32"""
    out = {}
    for dt, suf in ((torch.float64, ''), (torch.float32, '__f32')):
        torch.set_default_dtype(dt)
        tf.reset()
        res = fn()
        for k, v in res.items():
            out[k + suf] = np.asarray(v)
    torch.set_default_dtype(torch.float32)
    return out


def spd(rng, k, d, scale=1.0):
    a = rng.standard_normal((k, d, d))
    return f32(scale * (a @ a.transpose(0, 2, 1) / d + 0.5 * np.eye(d)))


# ============================================================================ distributions
def case_distributions(seed, N, K, L, S):
    rng = np.random.Generator(np.random.PCG64(seed))
    mu = f32(rng.standard_normal((K, L)))
    sigma = spd(rng, K, L)
    x = f32(rng.standard_normal((N, L)) * 2)
    # per-(n,k) natural parameters: eta2 = -0.5 * SPD
    P_nk = spd(rng, N * K, L).reshape(N, K, L, L)
    eta2_nk = f32(-0.5 * P_nk)
    eta1_nk = f32(rng.standard_normal((N, K, L)))
    w = rng.random(K) + 0.1
    w = f32(w / w.sum())
    xs = f32(rng.standard_normal((N, K, S, L)) * 1.5)
    beta = f32(rng.random(K) + 0.5)
    m = f32(rng.standard_normal((K, L)))
    C = spd(rng, K, L, 3.0)
    v = f32(L + 1.5 + rng.random(K) * 5)
    alpha = f32(rng.random(K) * 3 + 0.2)
    dof = f32(rng.random(K) * 6 + 2.5)
    inputs = dict(mu=mu, sigma=sigma, x=x, eta1_nk=eta1_nk, eta2_nk=eta2_nk, w=w, xs=xs, beta=beta, m=m, C=C, v=v,
                  alpha=alpha, dof=dof)

    def run():
        o = {}
        e1, e2 = R.gaussian.standard_to_natural(T(mu), T(sigma))
        o['s2n_eta1'], o['s2n_eta2'] = npy(e1), npy(e2)
        mu2, sig2 = R.gaussian.natural_to_standard(e1, e2)
        o['n2s_mu'], o['n2s_sigma'] = npy(mu2), npy(sig2)
        o['logprob_nat'] = npy(R.gaussian.log_probability_nat(T(x), T(eta1_nk), T(eta2_nk), T(w)))
        o['logprob_nat_now'] = npy(R.gaussian.log_probability_nat(T(x), T(eta1_nk), T(eta2_nk), None))
        o['logprob_per_samp'] = npy(R.gaussian.log_probability_nat_per_samp(T(xs), T(eta1_nk), T(eta2_nk)))
        o['logdet'] = npy(R.tf_utils.logdet(T(sigma)))
        em, eC = R.niw.expected_values((T(beta), T(m), T(C), T(v)))
        o['niw_exp_m'], o['niw_exp_C'] = npy(em), npy(eC)
        A, b, be, vh = R.niw.standard_to_natural(T(beta), T(m), T(C), T(v))
        o['niw_A'], o['niw_b'], o['niw_beta'], o['niw_vhat'] = npy(A), npy(b), npy(be), npy(vh)
        be2, m2, C2, v2 = R.niw.natural_to_standard(A, b, be, vh)
        o['niw_back_m'], o['niw_back_C'], o['niw_back_v'] = npy(m2), npy(C2), npy(v2)
        o['dir_elogpi'] = npy(R.dirichlet.expected_log_pi(T(alpha)))
        o['dir_nat'] = npy(R.dirichlet.standard_to_natural(T(alpha)))
        o['dir_std'] = npy(R.dirichlet.natural_to_standard(T(alpha)))
        o['student_t'] = npy(R.student_t.log_probability_per_samp(T(xs), T(mu), T(sigma), T(dof)))
        return o

    out = both(run)
    out.update({'in_' + k: v for k, v in inputs.items()})
    return out


# ============================================================================ pure GMM / SMM VMP
def synth_gmm(rng, N, D, K):
    """SURVEY 8d synthetic mixture: centres ~ N(0, 25 I), unit covariance, uniform labels."""
    c = rng.standard_normal((K, D)) * 5.0
    z = rng.integers(0, K, size=N)
    x = c[z] + rng.standard_normal((N, D))
    r0 = np.exp(3.0 * rng.standard_normal((N, K)))
    r0 /= r0.sum(1, keepdims=True)
    return f32(x), f32(r0), z


def case_gmm(seed, N, D, K, steps=3, kappa=5.0, miss_ratio=0.2):
    rng = np.random.Generator(np.random.PCG64(seed))
    x, r0, z = synth_gmm(rng, N, D, K)
    miss = rng.random((N, D)) < miss_ratio
    inputs = dict(x=x, r0=r0, miss=miss, kappa=np.float64(kappa))

    def run():
        o = {}
        # ---- gmm.inference exactly as gmm.py:230-269 builds it, iterated `steps` times
        r = T(r0)
        for it in range(steps):
            tf.reset()
            tf.INJECT['dirichlet'].append(r)
            step, log_r, theta, aux = R.gmm.inference(T(x), K, seed=0)
            names = ('alpha', 'beta', 'm', 'C', 'v')
            for n_, t_ in zip(names, theta):
                o['gmm%d_%s' % (it, n_)] = npy(t_)
            o['gmm%d_xk' % it], o['gmm%d_Sk' % it], o['gmm%d_pi' % it] = npy(aux[0]), npy(aux[1]), npy(aux[2])
            o['gmm%d_r' % it] = npy(step)
            o['gmm%d_logr' % it] = npy(log_r)
            r = step
        # ---- stand-alone m_step / e_step / e_step_missing_data on the step-0 quantities
        tf.reset()
        alpha, A, b, beta, v_hat = R.svae.init_mm_params(K, D, alpha_scale=0.05 / K, beta_scale=0.5, m_scale=0,
                                                         C_scale=D + 0.5, v_init=D + 0.5, seed=0, name='prior',
                                                         trainable=False)
        o['prior_alpha'], o['prior_A'], o['prior_b'], o['prior_beta'], o['prior_vhat'] = \
            npy(alpha), npy(A), npy(b), npy(beta), npy(v_hat)
        beta_0, m_0, C_0, v_0 = R.niw.natural_to_standard(A, b, beta, v_hat)
        alpha_0 = R.dirichlet.natural_to_standard(alpha)
        ak, bk, mk, Ck, vk, xk, Sk = R.gmm.m_step(T(x), T(r0), alpha_0, beta_0, m_0, C_0, v_0)
        Pk = tf.matrix_inverse(Ck)
        o['P0'] = npy(Pk)
        o['elogdet0'] = npy(R.gmm.compute_expct_log_det_prec(vk, Pk))
        o['maha0'] = npy(R.gmm.compute_expct_mahalanobis_dist(T(x), bk, mk, Pk, vk))
        rm, pim = R.gmm.e_step_missing_data(T(x), ak, bk, mk, Pk, vk, torch.as_tensor(miss))
        o['miss_r'], o['miss_pi'] = npy(rm), npy(pim)
        # ---- SMM (smm.py:199-245), u initialised to ones
        r = T(r0)
        u = T(np.ones((N, K)))
        for it in range(steps):
            ak, bk, mk, Ck, vk, xk, Sk = R.smm.m_step(T(x), r, u, alpha_0, beta_0, m_0, C_0, v_0)
            Pk = tf.matrix_inverse(Ck)
            kap = T(np.full((K,), kappa))
            r, u, pi = R.smm.e_step(T(x), ak, bk, mk, Pk, vk, kap)
            for n_, t_ in zip(('alpha', 'beta', 'm', 'C', 'v', 'xk', 'Sk', 'r', 'u', 'pi'),
                              (ak, bk, mk, Ck, vk, xk, Sk, r, u, pi)):
                o['smm%d_%s' % (it, n_)] = npy(t_)
        return o

    out = both(run)
    out.update({'in_' + k: v for k, v in inputs.items()})
    return out


# ============================================================================ SVAE
NET_VARS = ('layer_0/kernel', 'layer_0/bias', 'layer_1/kernel', 'layer_1/bias',
            'gaussian_output/kernel', 'gaussian_output/bias', 'shortcut/W', 'shortcut/b1', 'shortcut/b2')


def make_weights(rng, Dy, L, U, std=0.3):
    """MLP weights for both nets.  std larger than the reference's 0.01 init so that the nets are far
    from linear and the test is discriminating; shortcut W comes from the reference's own
    rand_partial_isometry (vae.py:58-72)."""
    w = {}
    for net, din, dout in (('encoder_net', Dy, L), ('decoder_net', L, Dy)):
        dims = [din, U, U]
        for i in range(2):
            w['%s/layer_%d/kernel' % (net, i)] = f32(rng.standard_normal((dims[i], U)) * std)
            w['%s/layer_%d/bias' % (net, i)] = f32(rng.standard_normal((U,)) * std)
        w['%s/gaussian_output/kernel' % net] = f32(rng.standard_normal((U, 2 * dout)) * std)
        w['%s/gaussian_output/bias' % net] = f32(rng.standard_normal((2 * dout,)) * std)
        w['%s/shortcut/W' % net] = f32(R.vae.rand_partial_isometry(din, dout, 1., seed=0))
        w['%s/shortcut/b1' % net] = f32(rng.standard_normal((dout,)) * 0.1)
        w['%s/shortcut/b2' % net] = f32(rng.standard_normal((dout,)) * 0.1)
    return w


def load_weights(w):
    for k, v in w.items():
        t = T(v, grad=True)
        t._tf_name = k + ':0'
        tf.VARIABLES[k] = t


def case_svae(seed, N, K, L, S, Dy, U, smm=False, steps=3, lr=3e-4, lrcvi=0.2, decay=0.95, y=None, slim=False):
    """`y` (N,Dy): data rows to train on (default: a synthetic mixture).  `slim`: keep the large per-sample outputs
    (reconstructions, phi_tilde) out of the fixture and store x_k in fp64 only."""
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.standard_normal((K, Dy)) * 2.0
    y_syn = f32(c[rng.integers(0, K, size=N)] + 0.5 * rng.standard_normal((N, Dy)))
    y = y_syn if y is None else f32(y)
    assert y.shape == (N, Dy)
    weights = make_weights(rng, Dy, L, U)
    m_unif = f32(rng.random((K, L)))                      # tf.random_uniform draw of svae.py:440 (pre-scaling)
    pi_norm = f32(rng.standard_normal((K,)))              # tf.random_normal draw of svae.py:491
    Lk_low = f32(np.tril(rng.standard_normal((K, L, L)) * 0.3, -1))   # make recognition L_k non-diagonal
    noise = [f32(rng.standard_normal((N, K, L, S))) for _ in range(steps)]
    zdraw = [rng.integers(0, K, size=(N, S)) for _ in range(steps)]
    dof0 = 5.0
    inputs = dict(y=y, m_unif=m_unif, pi_norm=pi_norm, Lk_low=Lk_low, noise=np.stack(noise), zdraw=np.stack(zdraw),
                  lr=np.float64(lr), lrcvi=np.float64(lrcvi), decay=np.float64(decay), dof0=np.float64(dof0))
    inputs.update({'w_' + k: v for k, v in weights.items()})
    tanh = tf.tanh
    enc_layers = [(U, tanh), (U, tanh), (L, 'natparam')]
    dec_layers = [(U, tanh), (U, tanh), (Dy, 'standard')]

    def init_params():
        """experiments.py:154-181 with the TF-RNG pieces injected"""
        tf.reset()
        load_weights(weights)
        # init_mm_params draws random_uniform twice (prior: m_scale=0, theta: m_scale=5)
        tf.INJECT['random_uniform'] += [T(m_unif), T(m_unif)]
        tf.INJECT['random_normal'] += [T(pi_norm)]
        if smm:
            gmm_prior, theta0 = R.svae.init_mm(K, L, seed=0, theta_as_variable=False)
            with tf.variable_scope('theta'):
                mu_k, L_k = R.svae.make_loc_scale_variables(gmm_prior)
                DoF = T(np.full((K,), dof0))
                alpha_k = T(npy(theta0[0]))
            phi_gmm = R.svae.init_recognition_params(theta0, K, seed=0)
            gmm_prior = gmm_prior[0]
            theta = [alpha_k, mu_k, L_k, DoF]
        else:
            gmm_prior, theta = R.svae.init_mm(K, L, seed=0)
            phi_gmm = R.svae.init_recognition_params(theta, K, seed=0)
            theta = list(theta)
        phi_gmm = list(phi_gmm)
        # asymmetric perturbation of the recognition factor (not in the reference's init; makes the
        # fixture transpose-detecting)
        phi_gmm[1] = (phi_gmm[1].detach() + T(Lk_low)).as_subclass(tf.Tensor).requires_grad_(True)
        tf.VARIABLES['phi_gmm/L_k'] = phi_gmm[1]
        return gmm_prior, theta, phi_gmm

    def trainables(theta, phi_gmm):
        names = ['phi_gmm/mu_k', 'phi_gmm/L_k', 'phi_gmm/log_pi_k']
        ts = [phi_gmm[0], phi_gmm[1], phi_gmm[2]]
        if smm:
            names += ['theta/mu_k', 'theta/L_k']
            ts += [theta[1], theta[2]]
        for net in ('encoder_net', 'decoder_net'):
            for v in NET_VARS:
                names.append(net + '/' + v)
                ts.append(tf.VARIABLES[net + '/' + v])
        return names, ts

    def run():
        o = {}
        gmm_prior, theta, phi_gmm = init_params()
        if smm:
            o['prior_alpha'] = npy(gmm_prior)
            for n_, t_ in zip(('alpha', 'mu', 'L', 'dof'), theta):
                o['theta_init_' + n_] = npy(t_)
        else:
            for n_, t_ in zip(('alpha', 'A', 'b', 'beta', 'vhat'), gmm_prior):
                o['prior_' + n_] = npy(t_)
            for n_, t_ in zip(('alpha', 'A', 'b', 'beta', 'vhat'), theta):
                o['theta_init_' + n_] = npy(t_)
        for n_, t_ in zip(('mu_k', 'L_k', 'log_pi_k'), phi_gmm):
            o['phi_init_' + n_] = npy(t_)
        names, params = trainables(theta, phi_gmm)
        adam_m = [torch.zeros_like(p) for p in params]
        adam_v = [torch.zeros_like(p) for p in params]
        b1, b2, eps = 0.9, 0.999, 1e-8
        for it in range(steps):
            pre = 'step%d_' % it
            tf.INJECT['random_normal'] += [T(noise[it])]
            tf.INJECT['multinomial'] += [torch.as_tensor(zdraw[it])]
            Y = T(y)
            (y_rec, y_enc, x_k, x_s, log_z, _, phi_tilde) = R.svae.inference(Y, phi_gmm, enc_layers, dec_layers, S,
                                                                             stddev_init_nn=0.01, seed=0)
            if smm:
                elbo, details = R.svae.compute_elbo_smm(Y, y_rec, theta, phi_tilde, x_k, log_z, 'standard')
            else:
                elbo, details = R.svae.compute_elbo(Y, y_rec, theta, phi_tilde, x_k, log_z, 'standard')
            grads = torch.autograd.grad(-elbo, params, allow_unused=True)
            o[pre + 'enc_eta1'], o[pre + 'enc_eta2'] = npy(y_enc[0]), npy(y_enc[1])
            o[pre + 'x_k'], o[pre + 'x_s'], o[pre + 'log_z'] = npy(x_k), npy(x_s), npy(log_z)
            if not slim:
                o[pre + 'rec_mean'], o[pre + 'rec_var'] = npy(y_rec[0]), npy(y_rec[1])
                o[pre + 'phi_tilde_eta1'], o[pre + 'phi_tilde_eta2'] = npy(phi_tilde[0]), npy(phi_tilde[1])
            o[pre + 'elbo'] = npy(elbo)
            o[pre + 'details'] = np.stack([npy(d) for d in details])
            for n_, g in zip(names, grads):
                o[pre + 'grad_' + n_] = npy(g)
            # ---- CVI update of theta from OLD values (experiments.py:250-260)
            step_lrcvi = lrcvi * decay ** (it / 1000.0)            # global_step = it before apply_gradients
            o[pre + 'lrcvi'] = np.float64(step_lrcvi)
            r_nk = tf.exp(log_z).detach()
            if smm:
                alpha_star = R.svae.m_step_smm(gmm_prior, r_nk)
                o[pre + 'theta_star_alpha'] = npy(alpha_star)
                new_alpha = ((1 - step_lrcvi) * theta[0] + step_lrcvi * alpha_star).detach()
            else:
                theta_star = R.svae.m_step(gmm_prior, x_s.detach(), r_nk)
                for n_, t_ in zip(('alpha', 'A', 'b', 'beta', 'vhat'), theta_star):
                    o[pre + 'theta_star_' + n_] = npy(t_)
                new_theta = [((1 - step_lrcvi) * c_ + step_lrcvi * s_).detach() for c_, s_ in zip(theta, theta_star)]
            # ---- Adam, TF-1.3 formulation (single tower: average_gradients is the identity)
            t_ = it + 1
            lr_t = lr * np.sqrt(1 - b2 ** t_) / (1 - b1 ** t_)
            with torch.no_grad():
                for p, g, m_, v_ in zip(params, grads, adam_m, adam_v):
                    m_.mul_(b1).add_(g, alpha=1 - b1)
                    v_.mul_(b2).addcmul_(g, g, value=1 - b2)
                    p.sub_(lr_t * m_ / (v_.sqrt() + eps))
            if smm:
                theta[0] = tf._T(new_alpha)
            else:
                theta = [tf._T(t) for t in new_theta]
            for n_, p in zip(names, params):
                o[pre + 'param_' + n_] = npy(p)
            if smm:
                o[pre + 'theta_alpha'] = npy(theta[0])
            else:
                for n_, t_2 in zip(('alpha', 'A', 'b', 'beta', 'vhat'), theta):
                    o[pre + 'theta_' + n_] = npy(t_2)
        return o

    out = both(run)
    if slim:
        for k in [k for k in out if k.endswith('__f32') and (k.endswith('x_k__f32') or k.endswith('x_s__f32'))]:
            del out[k]
    out.update({'in_' + k: v for k, v in inputs.items()})
    out['in_dims'] = np.array([N, K, L, S, Dy, U, steps, int(smm)])
    return out


# ============================================================================ data.py: loaders, split, scaling, perturbation
def case_datasets():
    """Outputs of the reference's own data.py (make_pinwheel_data :216-235, make_minibatch :9-176 with
    size_minibatch=-1 i.e. the full split tensors, perturb_data :238-259) on the dataset files that ship with the
    reference (datasets/Auto/auto-mpg.csv, Aggregation.txt, geyser).  Only the processed arrays are stored."""
    import data as rdata
    tf.reset()
    o = {}
    X, lab = rdata.make_pinwheel_data(0.3, 0.05, 5, 200, 0.25)
    o['pinwheel_data'], o['pinwheel_labels'] = X, lab
    for ds in ('auto', 'aggregation', 'geyser', 'pinwheel', 'noisy-pinwheel'):
        X_tr, y_tr, X_te, y_te = rdata.make_minibatch(ds, ratio_tr=0.7, path_datadir=os.path.join(REF, 'datasets'),
                                                      size_minibatch=-1, size_testbatch=-1, seed_split=0,
                                                      noise_level=0.1)
        key = ds.replace('-', '_')
        o[key + '_X_tr'], o[key + '_X_te'] = npy(X_tr), npy(X_te)
        o[key + '_y_tr'], o[key + '_y_te'] = npy(y_tr), npy(y_te)
    # validation split (data.py:91-105): the training part is split again and the VALIDATION rows come back as "test"
    for ds in ('auto', 'noisy-pinwheel', 'geyser'):
        X_tr, y_tr, X_te, y_te = rdata.make_minibatch(ds, ratio_tr=0.6, ratio_val=0.2, path_datadir=os.path.join(REF, 'datasets'),
                                                      size_minibatch=-1, size_testbatch=-1, seed_split=3, noise_level=0.1)
        key = 'val_' + ds.replace('-', '_')
        o[key + '_X_tr'], o[key + '_X_te'] = npy(X_tr), npy(X_te)
        o[key + '_y_tr'], o[key + '_y_te'] = npy(y_tr), npy(y_te)
    z = np.arange(60, dtype=np.float64).reshape(20, 3)
    o['perturb_in'] = z.copy()
    o['perturb_out'] = rdata.perturb_data(z.copy(), noise_ratio=0.25, noise_mean=1.0, noise_stddev=3.0, seed=7)
    return o


def auto_minibatch(n=64, seed=0):
    """The first minibatch of the Auto training set (reference loader, standardised x5) under the build's PCG64 shuffle."""
    d = np.load(os.path.join(HERE, 'datasets.npz'))
    X = d['auto_X_tr']
    return X[np.random.Generator(np.random.PCG64(seed)).permutation(X.shape[0])[:n]]


# ============================================================================ evaluation metrics (losses.py)
def case_metrics(seed, N, K, S, D, C):
    rng = np.random.Generator(np.random.PCG64(seed))
    y = f32(rng.standard_normal((N, D)))
    mean = f32(y[:, None, None, :] + 0.7 * rng.standard_normal((N, K, S, D)))
    var = f32(0.2 + rng.random((N, K, S, D)))
    lw = rng.standard_normal((N, K))
    lw = f32(lw - np.log(np.exp(lw).sum(1, keepdims=True)))
    lws = f32(lw[:, :, None] + 0.1 * rng.standard_normal((N, K, S)))
    mask = rng.random((N, D)) < 0.4
    lab = np.eye(C)[rng.integers(0, C, size=N)]
    inputs = dict(y=y, mean=mean, var=var, lw=lw, lws=lws, mask=mask, labels=f32(lab))

    def run():
        o = {}
        r = tf.exp(T(lw))
        o['weighted_mse'] = npy(R.losses.weighted_mse(T(y), T(mean), r))
        o['loli'] = npy(R.losses.diagonal_gaussian_logprob(T(y), T(mean), T(var), T(lw)))
        o['loli_s'] = npy(R.losses.diagonal_gaussian_logprob(T(y), T(mean), T(var), T(lws)))
        o['loli_mask'] = npy(R.losses.diagonal_gaussian_logprob(T(y), T(mean), T(var), T(lw), mask=tf._T(torch.as_tensor(mask))))
        e, p_ = R.losses.purity(r, T(lab))
        o['entropy'], o['purity'] = npy(e), npy(p_)
        return o

    out = both(run)
    out.update({'in_' + k: v for k, v in inputs.items()})
    return out

# ============================================================================ missing-data imputation (losses.py:148-310)
def imputation_toy_method(y_pert, K, S, D, a_ks, b_ksd, Wr, lib):
    """A deterministic stand-in for the `impute` closure of experiments.py:365-372 (which runs svae.inference on the
    perturbed data): (mean (N,K,S,D), var (N,K,S,D), log_r_nk (N,K)) as smooth functions of y_perturbed.  `lib` is the
    tensor namespace (the tf shim here, torch in the tests) - only +,*,exp,log, sum are used."""
    mean = y_pert[:, None, None, :] * a_ks[None, :, :, None] + b_ksd[None]
    var = 0.3 + 0.5 * (y_pert[:, None, None, :] * a_ks[None, :, :, None]) ** 2
    logits = y_pert @ Wr
    log_r = logits - lib.log(lib.exp(logits).sum(1, keepdim=True))
    return mean, var, log_r


def case_imputation(seed, N, K, S, D, P):
    rng = np.random.Generator(np.random.PCG64(seed))
    y = f32(rng.standard_normal((N, D)) * 1.5)
    noise = f32(rng.standard_normal((P, N, D)))
    a_ks = f32(0.6 + 0.4 * rng.random((K, S)))
    b_ksd = f32(0.3 * rng.standard_normal((K, S, D)))
    Wr = f32(0.5 * rng.standard_normal((D, K)))
    y_pred = f32(y[:, None, None, :] + 0.7 * rng.standard_normal((N, K, S, D)))
    lw = rng.standard_normal((N, K))
    r = f32(np.exp(lw) / np.exp(lw).sum(1, keepdims=True))
    inputs = dict(y=y, noise=noise, a_ks=a_ks, b_ksd=b_ksd, Wr=Wr, y_pred=y_pred, r=r, dims=np.array([N, K, S, D, P]),
                  mask_seed=np.array(seed), mask_ratio=np.array(0.3))

    def run():
        o = {}
        mask = R.losses.generate_missing_data_mask(T(y), 0.3, seed=seed)
        o['mask'] = npy(mask)
        o['mask_quarter'] = npy(R.losses.generate_missing_data_mask(T(np.zeros((3, 16))), mask_type='quarter'))
        o['mask_left_half'] = npy(R.losses.generate_missing_data_mask(T(np.zeros((2, 16))), mask_type='left_half'))
        tf.INJECT['random_normal'] += [T(noise[0])]
        o['perturbed0'] = npy(R.losses.perturb_data(T(y), mask, seed))
        o['imputation_mse'] = npy(R.losses.imputation_mse(T(y), T(y_pred), T(r), mask))
        tf.INJECT['random_normal'] += [T(noise[p]) for p in range(P)]
        method = lambda yp: imputation_toy_method(yp, K, S, D, T(a_ks), T(b_ksd), T(Wr), torch)
        mse, ll = R.losses.imputation_losses(T(y), mask, method, nb_samples_pert=P, nb_samples_rec=S, seed=seed)
        o['imp_mse'], o['imp_loglike'] = npy(mse), npy(ll)
        return o

    out = both(run)
    out.update({'in_' + k: v for k, v in inputs.items()})
    return out

# ============================================================================ Bernoulli decoder + plain VAE (8f rank 4)
def case_vae_bernoulli(seed, N, K, S, L, D, U):
    rng = np.random.Generator(np.random.PCG64(seed))
    y_bin = f32(np.where(rng.random((N, D)) < 0.5, -1.0, 1.0))
    y_real = f32(rng.standard_normal((N, D)) * 1.5)
    x4 = f32(rng.standard_normal((N, K, S, L)))
    lw = rng.standard_normal((N, K))
    lw = f32(lw - np.log(np.exp(lw).sum(1, keepdims=True)))
    lws = f32(lw[:, :, None] + 0.1 * rng.standard_normal((N, K, S)))
    mask = rng.random((N, D)) < 0.3
    noise_rep = f32(rng.standard_normal((N, S, L)))
    unif_pert = f32(rng.random((N, D)))
    std = 0.4
    w = {}
    for head, dout, key in (('bernoulli', D, 'bernoulli_output'), ('standard', 2 * D, 'gaussian_output')):
        pre = 'dec_%s/' % head
        w[pre + 'layer_0/kernel'] = f32(rng.standard_normal((L, U)) * std)
        w[pre + 'layer_0/bias'] = f32(rng.standard_normal((U,)) * std)
        w[pre + 'layer_1/kernel'] = f32(rng.standard_normal((U, U)) * std)
        w[pre + 'layer_1/bias'] = f32(rng.standard_normal((U,)) * std)
        w[pre + key + '/kernel'] = f32(rng.standard_normal((U, dout)) * std)
        w[pre + key + '/bias'] = f32(rng.standard_normal((dout,)) * std)
        w[pre + 'shortcut/W'] = f32(R.vae.rand_partial_isometry(L, D, 1., seed=0))
        w[pre + 'shortcut/b1'] = f32(rng.standard_normal((D,)) * 0.1)
        if head == 'standard':
            w[pre + 'shortcut/b2'] = f32(rng.standard_normal((D,)) * 0.1)
    pre = 'enc/'
    w[pre + 'layer_0/kernel'] = f32(rng.standard_normal((D, U)) * std)
    w[pre + 'layer_0/bias'] = f32(rng.standard_normal((U,)) * std)
    w[pre + 'layer_1/kernel'] = f32(rng.standard_normal((U, U)) * std)
    w[pre + 'layer_1/bias'] = f32(rng.standard_normal((U,)) * std)
    w[pre + 'gaussian_output/kernel'] = f32(rng.standard_normal((U, 2 * L)) * std)
    w[pre + 'gaussian_output/bias'] = f32(rng.standard_normal((2 * L,)) * std)
    w[pre + 'shortcut/W'] = f32(R.vae.rand_partial_isometry(D, L, 1., seed=0))
    w[pre + 'shortcut/b1'] = f32(rng.standard_normal((L,)) * 0.1)
    w[pre + 'shortcut/b2'] = f32(rng.standard_normal((L,)) * 0.1)
    inputs = dict(y_bin=y_bin, y_real=y_real, x4=x4, lw=lw, lws=lws, mask=mask, noise_rep=noise_rep, unif_pert=unif_pert,
                  dims=np.array([N, K, S, L, D, U]))
    inputs.update({'w_' + k: v for k, v in w.items()})

    def load(net, prefix):
        names = []
        for k, v in w.items():
            if k.startswith(prefix):
                full = net + '/' + k[len(prefix):]
                t = T(v, grad=True)
                t._tf_name = full + ':0'
                tf.VARIABLES[full] = t
                names.append(full)
        return names

    def run():
        o = {}
        tanh = tf.tanh
        # ---- Bernoulli decoder on the SVAE sample tensor (svae.py:511, vae.py:138-151)
        load('decoder_net', 'dec_bernoulli/')
        probas, logits = R.vae.make_decoder(T(x4), [(U, tanh), (U, tanh), (D, 'bernoulli')], stddev_init=0.3)
        o['probas'], o['logits'] = npy(probas), npy(logits)
        r = tf.exp(T(lw))
        o['ebl_weighted'] = npy(R.vae.expected_bernoulli_loglike(T(y_bin), logits, r))
        o['ebl_plain'] = npy(R.vae.expected_bernoulli_loglike(T(y_bin), logits[:, 0]))
        o['blp_plain'] = npy(R.losses.bernoulli_logprob(T(y_bin), logits[:, 0]))
        o['blp_w'] = npy(R.losses.bernoulli_logprob(T(y_bin), logits, T(lw)))
        o['blp_ws_mask'] = npy(R.losses.bernoulli_logprob(T(y_bin), logits, T(lws), tf._T(torch.as_tensor(mask))))
        tf.INJECT['random_uniform'] += [T(unif_pert)]
        o['perturbed_bern'] = npy(R.losses.perturb_data(T(y_bin), tf._T(torch.as_tensor(mask)), 0, decoder_type='bernoulli'))
        # ---- plain VAE (vae.py:131-135, 282-296, 253-279), Bernoulli and Gaussian decoders, with gradients
        for head, yv in (('bernoulli', y_bin), ('standard', y_real)):
            tf.reset()
            enc_names = load('encoder_net', 'enc/')
            dec_names = load('decoder_net', 'dec_%s/' % head)
            mu, var = R.vae.make_encoder(T(yv), [(U, tanh), (U, tanh), (L, 'standard')], stddev_init=0.3)
            tf.INJECT['random_normal'] += [T(noise_rep)]
            xs = R.vae.reparam_trick_sampling(mu, var, S, seed=0)
            dec = R.vae.make_decoder(xs, [(U, tanh), (U, tanh), (D, head)], stddev_init=0.3)
            elbo = R.vae.compute_elbo(T(yv), mu, var, dec, decoder_type=head)
            o['vae_%s_enc_mu' % head], o['vae_%s_enc_var' % head] = npy(mu), npy(var)
            o['vae_%s_x' % head] = npy(xs)
            o['vae_%s_kl' % head] = npy(R.vae.build_kl_divergence(mu, var))
            o['vae_%s_elbo' % head] = npy(elbo)
            names = enc_names + dec_names
            gr = torch.autograd.grad(-elbo, [tf.VARIABLES[n] for n in names])
            for n, g_ in zip(names, gr):
                o['vae_%s_grad_%s' % (head, n)] = npy(g_)
        return o

    out = both(run)
    out.update({'in_' + k: v for k, v in inputs.items()})
    return out


def main():
    cases = {
        'datasets': case_datasets,
        'svae_auto': lambda: case_svae(16, N=64, K=10, L=8, S=10, Dy=6, U=50, steps=3, y=auto_minibatch(), slim=True),
        'dist_tiny': lambda: case_distributions(1, N=6, K=4, L=3, S=5),
        'dist_l8': lambda: case_distributions(2, N=9, K=5, L=8, S=4),
        'gmm_tiny': lambda: case_gmm(3, N=60, D=2, K=3),
        'gmm_d6k10': lambda: case_gmm(4, N=400, D=6, K=10),
        'gmm_d8k16': lambda: case_gmm(5, N=700, D=8, K=16),
        'svae_tiny': lambda: case_svae(6, N=7, K=4, L=3, S=5, Dy=2, U=5),
        'svae_paper': lambda: case_svae(7, N=24, K=10, L=6, S=10, Dy=6, U=50),
        'svae_c1': lambda: case_svae(8, N=40, K=5, L=2, S=10, Dy=2, U=20),
        'svae_l8': lambda: case_svae(9, N=10, K=16, L=8, S=10, Dy=8, U=50, steps=2),
        'svae_smm_tiny': lambda: case_svae(10, N=7, K=4, L=3, S=5, Dy=2, U=5, smm=True),
        'metrics': lambda: case_metrics(12, N=50, K=5, S=7, D=3, C=4),
        'metrics_s100': lambda: case_metrics(13, N=12, K=10, S=100, D=6, C=3),
        'vae_bernoulli': lambda: case_vae_bernoulli(15, N=12, K=4, S=5, L=3, D=40, U=16),
        'imputation': lambda: case_imputation(14, N=40, K=5, S=6, D=6, P=4),
        'svae_smm_l8': lambda: case_svae(11, N=10, K=16, L=8, S=10, Dy=8, U=50, smm=True, steps=2),
    }
    only = sys.argv[1:]
    for name, fn in cases.items():
        if only and name not in only:
            continue
        out = fn()
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **out)
        print('%-16s %4d arrays %8.1f KB' % (name, len(out), os.path.getsize(path) / 1024.))


if __name__ == '__main__':
    main()
