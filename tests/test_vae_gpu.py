"""SURVEY 8f rank 4 on the GPU: Bernoulli decoder, expected_bernoulli_loglike, bernoulli_logprob, the plain-VAE ELBO
(KL + reconstruction, both decoder types) and its gradients through the reference-shaped surface, against the golden
produced by the reference's own models/vae.py and losses.py (tests/golden/vae_bernoulli.npz)."""
import numpy as np
import pytest
import torch

import parity_log

pytestmark = pytest.mark.gpu


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a)).to('cuda', dtype)


def rel(got, want):
    want = np.asarray(want, dtype=np.float64)
    got = got.detach().double().cpu().numpy()
    return parity_log.record('rel', np.abs(got - want).max() / max(np.abs(want).max(), 1e-300))


def load_net(g, scope, prefix):
    from vmp_for_svae_amd.models import vae
    for k in g.files:
        if k.startswith('in_w_' + prefix):
            vae.VARIABLES[scope + '/' + k[len('in_w_' + prefix):]] = torch.nn.Parameter(dev(g[k]))


def test_bernoulli_decoder_and_losses(golden):
    from vmp_for_svae_amd import losses
    from vmp_for_svae_amd.models import vae
    g = golden('vae_bernoulli')
    N, K, S, Ld, D, U = [int(v) for v in g['in_dims']]
    yb, x4, lw, lws = [dev(g['in_' + k]) for k in ('y_bin', 'x4', 'lw', 'lws')]
    mask = dev(g['in_mask'], torch.bool)
    vae.reset_variables()
    load_net(g, 'decoder_net', 'dec_bernoulli/')
    tanh = torch.tanh
    probas, logits = vae.make_decoder(x4, [(U, tanh), (U, tanh), (D, 'bernoulli')])
    assert rel(probas, g['probas']) < 1e-5 and rel(logits, g['logits']) < 1e-5
    assert rel(vae.expected_bernoulli_loglike(yb, logits, torch.exp(lw)), g['ebl_weighted']) < 1e-5
    assert rel(vae.expected_bernoulli_loglike(yb, logits[:, 0].contiguous()), g['ebl_plain']) < 1e-5
    assert rel(losses.bernoulli_logprob(yb, logits[:, 0].contiguous()), g['blp_plain']) < 1e-5
    assert rel(losses.bernoulli_logprob(yb, logits, lw), g['blp_w']) < 1e-5
    assert rel(losses.bernoulli_logprob(yb, logits, lws, mask), g['blp_ws_mask']) < 1e-5
    coin = torch.where(dev(g['in_unif_pert']) < 0.5, 1.0, -1.0)
    assert rel(losses.perturb_data(yb, mask, 0, decoder_type='bernoulli', noise=coin), g['perturbed_bern']) == 0
    p2 = losses.perturb_data(yb, mask, 3, decoder_type='bernoulli')
    assert set(torch.unique(p2).tolist()) <= {-1.0, 1.0} and torch.equal(p2[~mask], yb[~mask])
    vae.reset_variables()


@pytest.mark.parametrize('head', ['bernoulli', 'standard'])
def test_plain_vae_elbo_and_gradients(golden, head):
    from vmp_for_svae_amd.models import vae
    g = golden('vae_bernoulli')
    N, K, S, Ld, D, U = [int(v) for v in g['in_dims']]
    y = dev(g['in_y_bin'] if head == 'bernoulli' else g['in_y_real'])
    vae.reset_variables()
    load_net(g, 'encoder_net', 'enc/')
    load_net(g, 'decoder_net', 'dec_%s/' % head)
    tanh = torch.tanh
    mu, var = vae.make_encoder(y, [(U, tanh), (U, tanh), (Ld, 'standard')])
    xs = vae.reparam_trick_sampling(mu, var, S, 0, noise=dev(g['in_noise_rep']))
    dec = vae.make_decoder(xs, [(U, tanh), (U, tanh), (D, head)])
    elbo = vae.compute_elbo(y, mu, var, dec, decoder_type=head)
    assert rel(mu, g['vae_%s_enc_mu' % head]) < 1e-5 and rel(var, g['vae_%s_enc_var' % head]) < 1e-5
    assert rel(xs, g['vae_%s_x' % head]) < 1e-5
    assert rel(vae.build_kl_divergence(mu, var), g['vae_%s_kl' % head]) < 1e-5
    assert rel(elbo, g['vae_%s_elbo' % head]) < 1e-5
    names = [k[len('vae_%s_grad_' % head):] for k in g.files if k.startswith('vae_%s_grad_' % head) and not k.endswith('__f32')]
    grads = torch.autograd.grad(-elbo, [vae.VARIABLES[n] for n in names])
    for n_, gr in zip(names, grads):
        a, b = g['vae_%s_grad_%s' % (head, n_)], g['vae_%s_grad_%s__f32' % (head, n_)].astype(np.float64)
        bar = max(3e-5, 3 * np.abs(a - b).max() / max(np.abs(a).max(), 1e-300))
        assert rel(gr, a) <= bar, (n_, rel(gr, a), bar)
    vae.reset_variables()


@pytest.mark.parametrize('head', ['bernoulli', 'standard'])
def test_plain_vae_trainer_improves_elbo(head):
    from vmp_for_svae_amd.models import vae
    from vmp_for_svae_amd.training import VAETrainer
    vae.reset_variables()
    g = torch.Generator(device='cuda').manual_seed(1)
    z = torch.randn(256, 2, device='cuda', generator=g)
    proj = torch.randn(2, 6, device='cuda', generator=g)
    y = z @ proj + 0.1 * torch.randn(256, 6, device='cuda', generator=g)
    if head == 'bernoulli':
        y = torch.where(y > 0, 1.0, -1.0)
    tr = VAETrainer(2, 16, 6, nb_samples=5, lr=1e-2, stddev_init_nn=0.1, decoder_type=head)
    first = float(tr.step(y)['elbo'])
    for _ in range(150):
        out = tr.step(y)
    assert np.isfinite(float(out['elbo'])) and float(out['elbo']) > first + 0.5
    vae.reset_variables()
