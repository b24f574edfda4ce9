"""Fused decoder MLP + reconstruction-term kernels (csrc/vmp_decoder.hip, C ABI vmp_decoder_loglike_fwd/bwd) against
the oracle's fp64 restatement of reference models/vae.py:75-128 (make_nnet) and :233-248 (weights branch of
expected_diagonal_gaussian_loglike) with torch autograd as the gradient truth.  Tolerance: 1e-5 relative to the
largest magnitude of each tensor (fp32 MFMA accumulates exactly like an fp32 fma chain)."""
import numpy as np
import pytest
import torch

import parity_log

pytestmark = pytest.mark.gpu
NET_VARS = ('layer_0/kernel', 'layer_0/bias', 'layer_1/kernel', 'layer_1/bias', 'gaussian_output/kernel',
            'gaussian_output/bias', 'shortcut/W', 'shortcut/b1', 'shortcut/b2')


def make_case(N, K, S, Ld, Dy, U, seed, wscale=0.3):
    rng = np.random.Generator(np.random.PCG64(seed))
    shapes = ((Ld, U), (U,), (U, U), (U,), (U, 2 * Dy), (2 * Dy,), (Ld, Dy), (Dy,), (Dy,))
    w = [rng.standard_normal(s) * wscale for s in shapes]
    x = rng.standard_normal((N, K, S, Ld)) * 1.5
    y = rng.standard_normal((N, Dy))
    r = rng.random((N, K)) + 0.05
    return x, y, r, w


def truth(x, y, r, w):
    from oracle import nets
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in w]
    mean, var = nets.decoder(xt, dict(zip(NET_VARS, wt)))
    yy = torch.tensor(y).unsqueeze(1).unsqueeze(1)
    ll = ((yy - mean) ** 2 / var + torch.log(var + 1e-8)).sum(-1)              # (N,K,S)
    A = ll.sum(-1)
    loss = (A * torch.tensor(r)).sum()
    grads = torch.autograd.grad(loss, [xt] + wt)
    return mean.detach(), var.detach(), A.detach(), grads


def relerr(got, want):
    want = want.detach().double().cpu()
    return parity_log.record('rel', ((got.detach().double().cpu() - want).abs().max() / want.abs().max().clamp_min(1e-300)).item())


CASES = [  # N, K, S, L, Dy, U
    (7, 4, 5, 3, 2, 8),
    (5, 3, 10, 2, 2, 20),          # C1 sizes: L = Dy = 2, U = 20
    (9, 10, 10, 6, 6, 50),         # the paper's Auto configuration
    (33, 16, 10, 8, 8, 50),        # C3 sizes
    (3, 2, 7, 8, 5, 64),
    (4, 5, 3, 5, 8, 40),
    (1, 1, 1, 1, 1, 1),
    (2, 3, 16, 7, 3, 33),
]


@pytest.mark.parametrize('dims', CASES)
def test_fused_decoder_fwd_bwd_vs_oracle(dims):
    from vmp_for_svae_amd.models import _svae_ops
    N, K, S, Ld, Dy, U = dims
    x, y, r, w = make_case(N, K, S, Ld, Dy, U, seed=sum(dims))
    mean_t, var_t, A_t, g_t = truth(x, y, r, w)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device='cuda')
    xg = f32(x).requires_grad_(True)
    wg = [f32(a).requires_grad_(True) for a in w]
    A = _svae_ops.DecoderLoglikeFn.apply(f32(y), xg, *wg)
    assert relerr(A, A_t) < 1e-5
    grads = torch.autograd.grad((A * f32(r)).sum(), [xg] + wg)
    for n_, g, gt in zip(('x',) + NET_VARS, grads, g_t):
        assert tuple(g.shape) == tuple(gt.shape)
        assert relerr(g, gt) < 1e-5, (n_, relerr(g, gt))
    mean, var = _svae_ops.decoder_outputs(f32(x), wg)
    assert relerr(mean, mean_t) < 1e-5 and relerr(var, var_t) < 1e-5


def test_fused_decoder_matches_unfused_path_and_is_deterministic():
    """Same numbers as the torch-MLP + loglike-kernel path the reference-shaped surface uses otherwise; bitwise
    run-to-run determinism of the gradient reduction (fixed order, no atomics)."""
    from vmp_for_svae_amd.models import _svae_ops, vae
    N, K, S, Ld, Dy, U = 257, 16, 10, 8, 8, 50
    x, y, r, w = make_case(N, K, S, Ld, Dy, U, seed=5, wscale=0.2)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device='cuda')
    yg, rg = f32(y), f32(r)
    outs = []
    for rep in range(2):
        xg = f32(x).requires_grad_(True)
        wg = [f32(a).requires_grad_(True) for a in w]
        A = _svae_ops.DecoderLoglikeFn.apply(yg, xg, *wg)
        outs.append([A.detach()] + list(torch.autograd.grad((A * rg).sum(), [xg] + wg)))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    vae.reset_variables()
    for n_, a in zip(NET_VARS, w):
        vae.VARIABLES['decoder_net/' + n_] = torch.nn.Parameter(f32(a))
    xg = f32(x).requires_grad_(True)
    mean, var = vae.make_nnet(xg, [(U, torch.tanh), (U, torch.tanh), (Dy, 'standard')], 1., 'decoder_net')
    A2 = _svae_ops.DiagGaussLoglikeFn.apply(yg, mean, var)
    ps = [vae.VARIABLES['decoder_net/' + n_] for n_ in NET_VARS]
    g2 = torch.autograd.grad((A2 * rg).sum(), [xg] + ps)
    assert relerr(outs[0][0], A2) < 2e-6
    for n_, g, gt in zip(('x',) + NET_VARS, outs[0][1:], g2):
        assert relerr(g, gt) < 2e-5, (n_, relerr(g, gt))
    vae.reset_variables()


def test_fused_decoder_full_size_properties():
    """C3-sized chunk (N*K*S = 2.6e6 rows): A is additive over row blocks and the parameter gradient of the whole
    equals the sum of the gradients of two halves (linearity of the reduction); dx of each half is unchanged."""
    from vmp_for_svae_amd.models import _svae_ops
    N, K, S, Ld, Dy, U = 16384, 16, 10, 8, 8, 50
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(N, K, S, Ld, device='cuda', generator=g)
    y = torch.randn(N, Dy, device='cuda', generator=g)
    r = torch.rand(N, K, device='cuda', generator=g)
    shapes = ((Ld, U), (U,), (U, U), (U,), (U, 2 * Dy), (2 * Dy,), (Ld, Dy), (Dy,), (Dy,))
    w = [(torch.randn(s, device='cuda', generator=g) * 0.2) for s in shapes]

    def run(sl):
        xg = x[sl].clone().requires_grad_(True)
        wg = [a.clone().requires_grad_(True) for a in w]
        A = _svae_ops.DecoderLoglikeFn.apply(y[sl].contiguous(), xg, *wg)
        gr = torch.autograd.grad((A * r[sl]).sum(), [xg] + wg)
        return A.detach(), gr
    A, gr = run(slice(0, N))
    A1, g1 = run(slice(0, N // 2 + 3))
    A2, g2 = run(slice(N // 2 + 3, N))
    assert torch.isfinite(A).all()
    assert torch.equal(torch.cat([A1, A2]), A)
    assert torch.equal(torch.cat([g1[0], g2[0]]), gr[0])
    for a, b, c in zip(gr[1:], g1[1:], g2[1:]):
        assert relerr(b + c, a) < 2e-5


@pytest.mark.parametrize('scale', [1.0, -0.37])
def test_weighted_loglike_single_launch(scale):
    """DecoderWeightedLoglikeFn (value + all gradients from one launch of the backward kernel) == the two-launch
    DecoderLoglikeFn contracted with the weights, including the gradient w.r.t. the weights and a non-unit upstream."""
    from vmp_for_svae_amd.models import _svae_ops
    N, K, S, Ld, Dy, U = 37, 10, 10, 6, 6, 50
    x, y, r, w = make_case(N, K, S, Ld, Dy, U, seed=21)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device='cuda')
    res = []
    for fused in (True, False):
        xg, rg = f32(x).requires_grad_(True), f32(r).requires_grad_(True)
        wg = [f32(a).requires_grad_(True) for a in w]
        if fused:
            out = _svae_ops.DecoderWeightedLoglikeFn.apply(f32(y), xg, rg, *wg)
        else:
            out = (_svae_ops.DecoderLoglikeFn.apply(f32(y), xg, *wg) * rg).sum()
        res.append([out.detach()] + list(torch.autograd.grad(out * scale, [xg, rg] + wg)))
    for n_, a, b in zip(('value', 'x', 'weights') + NET_VARS, *res):
        assert relerr(a, b) < 3e-6, (n_, relerr(a, b))
    with torch.no_grad():
        out = _svae_ops.DecoderWeightedLoglikeFn.apply(f32(y), f32(x), f32(r), *[f32(a) for a in w])
    assert relerr(out, res[0][0]) < 1e-6


@pytest.mark.parametrize('dims', [(64, 6, 8, 50), (700, 2, 2, 20), (33, 8, 8, 64), (5, 3, 7, 33)])
@pytest.mark.parametrize('head', ['natparam', 'standard'])
def test_fused_encoder_vs_oracle(dims, head):
    """vae.make_encoder through the fused MLP kernels (GaussMLPFn: forward + gradient-input backward) against the
    oracle's fp64 make_nnet (vae.py:75-128) with torch autograd, for both Gaussian heads, incl. the input gradient."""
    from oracle import nets
    from vmp_for_svae_amd.models import vae
    R, Din, Dout, U = dims
    rng = np.random.Generator(np.random.PCG64(R + U))
    shapes = ((Din, U), (U,), (U, U), (U,), (U, 2 * Dout), (2 * Dout,), (Din, Dout), (Dout,), (Dout,))
    w = [rng.standard_normal(s) * 0.3 for s in shapes]
    x = rng.standard_normal((R, Din)) * 1.5
    g1, g2 = rng.standard_normal((R, Dout)), rng.standard_normal((R, Dout))
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in w]
    o1, o2 = nets.mlp(xt, dict(zip(NET_VARS, wt)), head)
    gt = torch.autograd.grad((o1 * torch.tensor(g1)).sum() + (o2 * torch.tensor(g2)).sum(), [xt] + wt)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device='cuda')
    vae.reset_variables()
    for n_, a in zip(NET_VARS, w):
        vae.VARIABLES['encoder_net/' + n_] = torch.nn.Parameter(f32(a))
    xg = f32(x).requires_grad_(True)
    e1, e2 = vae.make_encoder(xg, [(U, torch.tanh), (U, torch.tanh), (Dout, head)])
    assert relerr(e1, o1) < 1e-5 and relerr(e2, o2) < 1e-5
    ps = [vae.VARIABLES['encoder_net/' + n_] for n_ in NET_VARS]
    gd = torch.autograd.grad((e1 * f32(g1)).sum() + (e2 * f32(g2)).sum(), [xg] + ps)
    for n_, a, b in zip(('x',) + NET_VARS, gd, gt):
        assert relerr(a, b) < 1e-5, (n_, relerr(a, b))
    vae.reset_variables()
