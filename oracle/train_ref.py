"""Oracle: one training step of the reference driver (experiments.py:143-151, 196-267).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Adam / exponential_decay are TensorFlow-1.3 internals
(not in the reference tree); they are restated from the TF-1.3 documentation:
    lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
    var -= lr_t * m / (sqrt(v) + eps)                      (tf.train.AdamOptimizer)
    lrcvi = lrcvi0 * decay_rate ** (global_step / 1000)    (tf.train.exponential_decay, staircase=False)
"""
import math

import torch

from . import nets, svae_ref


class State(object):
    """All mutable state of the reference's training graph."""

    def __init__(self, phi_gmm, enc_w, dec_w, theta, gmm_prior, smm=False):
        self.phi_gmm = [p.detach().clone().requires_grad_(True) for p in phi_gmm]
        self.enc_w = {k: v.detach().clone().requires_grad_(True) for k, v in enc_w.items()}
        self.dec_w = {k: v.detach().clone().requires_grad_(True) for k, v in dec_w.items()}
        self.theta = [t.detach().clone() for t in theta]
        if smm:                                                    # experiments.py:160-161: mu_k, L_k trainable
            self.theta[1].requires_grad_(True)
            self.theta[2].requires_grad_(True)
        self.gmm_prior = gmm_prior
        self.smm = smm
        self.global_step = 0
        self.adam_m = None
        self.adam_v = None

    def trainables(self):
        """Order used by the fixtures: phi_gmm (3), [theta mu_k, L_k], encoder (9), decoder (9)."""
        names = ['phi_gmm/mu_k', 'phi_gmm/L_k', 'phi_gmm/log_pi_k']
        ts = list(self.phi_gmm)
        if self.smm:
            names += ['theta/mu_k', 'theta/L_k']
            ts += [self.theta[1], self.theta[2]]
        for net, w in (('encoder_net', self.enc_w), ('decoder_net', self.dec_w)):
            for v in nets.NET_VARS:
                names.append(net + '/' + v)
                ts.append(w[v])
        return names, ts


def tower_forward(st, y, noise, z_draws):
    """experiments.py:208-229 for one tower: inference + ELBO (sum over the shard)."""
    y_rec, phi_enc, x_k, x_s, log_z, _, phi_tilde = svae_ref.inference(y, st.phi_gmm, st.enc_w, st.dec_w, noise, z_draws)
    if st.smm:
        elbo, details = svae_ref.compute_elbo_smm(y, y_rec, st.theta, phi_tilde, x_k, log_z)
    else:
        elbo, details = svae_ref.compute_elbo(y, y_rec, st.theta, phi_tilde, x_k, log_z)
    return elbo, details, x_s, log_z


def train_step(st, y, noise, z_draws, lr, lrcvi0, decay_rate, towers=1, b1=0.9, b2=0.999, eps=1e-8, workers=None):
    """One `sess.run(training_step)`.  With towers=G the minibatch is split in G contiguous shards
    (data.py:174-175), per-tower gradients of -elbo are AVERAGED (helpers/tf_utils.py:52-87), log_z and
    x_samples are concatenated for the M-step (experiments.py:247-260).  Everything reads OLD values.
    Returns dict(elbo (sum over towers), details, grads (averaged), theta_star, lrcvi)."""
    names, params = st.trainables()
    ys, ns, zs = torch.chunk(y, towers), torch.chunk(noise, towers), torch.chunk(z_draws, towers)
    def one_tower(g):
        elbo, det, x_s, log_z = tower_forward(st, ys[g], ns[g], zs[g])
        gr = torch.autograd.grad(-elbo, params, allow_unused=True)
        gr = [torch.zeros_like(p) if g_ is None else g_ for g_, p in zip(gr, params)]
        return gr, elbo.detach(), torch.stack([d.detach() for d in det]), x_s.detach(), log_z.detach()

    if workers and workers > 1 and towers > 1:
        # test infrastructure only: the towers are independent graphs - several at a time on a thread pool (ATen releases the GIL),
        # combined in tower order below, so the sums do not depend on the pool
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(workers)) as ex:
            outs = list(ex.map(one_tower, range(towers)))
    else:
        outs = [one_tower(g) for g in range(towers)]
    grads_sum, elbos, details, xs_all, lz_all = None, [], [], [], []
    for gr, elbo_g, det_g, x_s, log_z in outs:
        grads_sum = gr if grads_sum is None else [a + b for a, b in zip(grads_sum, gr)]
        elbos.append(elbo_g)
        details.append(det_g)
        xs_all.append(x_s)
        lz_all.append(log_z)
    grads = [g_ / towers for g_ in grads_sum]
    lrcvi = lrcvi0 * decay_rate ** (st.global_step / 1000.0)
    r_nk = torch.exp(torch.cat(lz_all))
    if st.smm:
        theta_star = [svae_ref.m_step_smm(st.gmm_prior, r_nk)]
        new_theta0 = svae_ref.update_gmm_params([st.theta[0]], theta_star, lrcvi)
    else:
        theta_star = svae_ref.m_step(st.gmm_prior, torch.cat(xs_all), r_nk)
        new_theta = svae_ref.update_gmm_params(st.theta, theta_star, lrcvi)
    # Adam (TF form)
    if st.adam_m is None:
        st.adam_m = [torch.zeros_like(p) for p in params]
        st.adam_v = [torch.zeros_like(p) for p in params]
    t = st.global_step + 1
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    with torch.no_grad():
        for p, g_, m_, v_ in zip(params, grads, st.adam_m, st.adam_v):
            m_.mul_(b1).add_(g_, alpha=1 - b1)
            v_.mul_(b2).addcmul_(g_, g_, value=1 - b2)
            p.sub_(lr_t * m_ / (v_.sqrt() + eps))
    if st.smm:
        st.theta[0] = new_theta0[0].detach()
    else:
        st.theta = [t_.detach() for t_ in new_theta]
    st.global_step += 1
    return dict(elbo=torch.stack(elbos).sum(), details=torch.stack(details).sum(0), grads=dict(zip(names, grads)),
                theta_star=theta_star, lrcvi=lrcvi, x_samples=torch.cat(xs_all), log_z=torch.cat(lz_all))


def vmp_step_t2(phi_gmm, theta, gmm_prior, eta1, eta2d, noise, z_draws, Gx, Glz, lrcvi, smm=False, chunk=None, workers=None):
    """The T2 unit of SURVEY 8d on the reference graph: svae.e_step (svae.py:14-119), the regulariser part of compute_elbo(_smm)
    (svae.py:229-254 / 265-322), autodiff of  -elbo_reg + <x_k, Gx> + <log_z, Glz>  w.r.t. the encoder outputs and phi_gmm (+ the
    trainable theta/mu_k, theta/L_k of the Student-t model) - Gx, Glz standing for what the decoder and the reconstruction term send
    back (experiments.py:232) -, svae.subsample_x (s = 0 kept, svae.py:514), svae.m_step(_smm) and update_gmm_params
    (svae.py:154-196, 376-403).  `chunk`: rows per pass (the literal graph materialises (N,K,L,L) and (N,K,S,L) temporaries); the
    K-sized gradients and the M-step inputs are summed / concatenated over the passes.
    Returns dict(reg, g_eta1, g_eta2d, g_phi (3), g_theta (0 or 2), theta_new, log_z, x_samples)."""
    N = eta1.shape[0]
    chunk = N if chunk is None else chunk
    phi = [p.detach().clone().requires_grad_(True) for p in phi_gmm]
    th = [t.detach().clone() for t in theta]
    th_tr = []
    if smm:
        th[1].requires_grad_(True)
        th[2].requires_grad_(True)
        th_tr = [th[1], th[2]]
    def one_chunk(lo):
        sl = slice(lo, min(N, lo + chunk))
        e1 = eta1[sl].detach().clone().requires_grad_(True)
        e2 = eta2d[sl].detach().clone().requires_grad_(True)
        x_k, log_z, phi_tilde, _ = svae_ref.e_step((e1, e2), phi, noise[sl])
        elbo, det = (svae_ref.compute_elbo_smm if smm else svae_ref.compute_elbo)(None, None, th, phi_tilde, x_k, log_z)
        loss = -elbo + (x_k * Gx[sl]).sum() + (log_z * Glz[sl]).sum()
        gr = torch.autograd.grad(loss, [e1, e2] + phi + th_tr)
        return gr, det[3].detach(), svae_ref.subsample_x(x_k.detach(), z_draws[sl])[:, 0, :], log_z.detach()

    starts = list(range(0, N, chunk))
    if workers and workers > 1 and len(starts) > 1:
        # test infrastructure only: the row chunks are independent graphs - several at a time on a thread pool (ATen releases the
        # GIL), results combined in chunk order, so the sums do not depend on the pool
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(workers)) as ex:
            outs = list(ex.map(one_chunk, starts))
    else:
        outs = [one_chunk(lo) for lo in starts]
    g_k, g1, g2, regs, xs_all, lz_all = None, [], [], [], [], []
    for gr, reg_c, xs_c, lz_c in outs:
        g1.append(gr[0])
        g2.append(gr[1])
        g_k = list(gr[2:]) if g_k is None else [a + b for a, b in zip(g_k, gr[2:])]
        regs.append(reg_c)
        xs_all.append(xs_c)
        lz_all.append(lz_c)
    r_nk = torch.exp(torch.cat(lz_all))
    if smm:
        theta_new = svae_ref.update_gmm_params([th[0]], [svae_ref.m_step_smm(gmm_prior, r_nk)], lrcvi)
    else:
        theta_new = svae_ref.update_gmm_params(th, svae_ref.m_step(gmm_prior, torch.cat(xs_all), r_nk), lrcvi)
    return dict(reg=torch.stack(regs).sum(), g_eta1=torch.cat(g1), g_eta2d=torch.cat(g2), g_phi=g_k[:3], g_theta=g_k[3:],
                theta_new=[t.detach() for t in theta_new], log_z=torch.cat(lz_all), x_samples=torch.cat(xs_all))
