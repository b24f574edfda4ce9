"""Single-launch glue of the SVAE training step (csrc/vmp_step.hip; C ABI vmp_svae_elbo_tail, vmp_adam_step,
vmp_decoder_loglike_bwd_logw) and the autograd function built on it (_svae_ops.FusedElboFn): against the formulas of
reference svae.py:216-254 / vae.py:232-250 written with torch in fp64, and against the un-fused composition of the same
kernels."""
import ctypes
import math

import numpy as np
import pytest
import torch

import parity_log

pytestmark = pytest.mark.gpu


def rel(got, want):
    want = want.detach().double().cpu()
    return parity_log.record('rel', ((got.detach().double().cpu() - want).abs().max() / want.abs().max().clamp_min(1e-300)).item())


@pytest.mark.parametrize('N,K,S,Dy,sigma', [(64, 10, 10, 6, -1.0), (1, 1, 1, 1, 1.0), (5000, 16, 3, 8, -1.0),
                                            (40000, 16, 10, 8, 2.5), (0, 4, 2, 3, 1.0)])
def test_elbo_tail_kernel(N, K, S, Dy, sigma):
    """elbo / rec / reg, r and the two gradient seeds of ONE launch vs torch fp64; several grid shapes (one block, many
    blocks, grid-stride), launched twice on the same workspace (the ticket must come back to zero)."""
    import vmp_for_svae_amd as V
    L = V._lib
    g = torch.Generator(device='cuda').manual_seed(N + K)
    lz = torch.log_softmax(torch.randn(N, K, device='cuda', generator=g) * 2, -1)
    Tp = torch.randn(N, K, device='cuda', generator=g) * 3 - 5
    ll = torch.randn(N, K, S, device='cuda', generator=g) * 4 + 10
    ws = torch.zeros(L.lib().vmp_svae_elbo_tail_workspace_bytes(), dtype=torch.uint8, device='cuda')
    for _ in range(2):
        scal = torch.full((3,), float('nan'), device='cuda')
        g_lz, g_Tp, r = (torch.full((N, K), float('nan'), device='cuda') for _ in range(3))
        L.check(L.lib().vmp_svae_elbo_tail(L.ptr(lz), L.ptr(Tp), L.ptr(ll), N, K, S, Dy, sigma, L.ptr(scal), L.ptr(g_lz),
                                           L.ptr(g_Tp), L.ptr(r), L.ptr(ws), ws.numel(), L.stream()), 'vmp_svae_elbo_tail')
        lz64 = lz.double().requires_grad_(True)
        Tp64 = Tp.double().requires_grad_(True)
        r64 = torch.exp(lz64)
        rec = -0.5 / S * (r64 * ll.double().sum(-1)).sum() - N * Dy / 2.0 * math.log(2 * math.pi)
        reg = (r64 * (Tp64 + lz64)).sum()
        elbo = rec - reg
        want = torch.stack([elbo, rec, reg]).detach()
        scale = max(abs(rec.item()), abs(reg.item()), 1e-30)
        assert ((scal.double().cpu() - want.cpu()).abs().max() / scale).item() < 2e-7
        if N:
            glz, gTp = torch.autograd.grad(sigma * elbo, [lz64, Tp64])
            assert rel(r, r64) < 3e-7
            assert rel(g_lz, glz) < 1e-6 and rel(g_Tp, gTp) < 3e-7
        assert int(ws[:4].view(torch.int32).item()) == 0


def test_adam_step_kernel():
    """All tensors in one launch (40 tensors: two batches of <= 32; sizes around the 1024-element block boundary) vs the
    TF-1.3 Adam formulas in torch; step size by value and through the device word."""
    import vmp_for_svae_amd as V
    L = V._lib
    g = torch.Generator(device='cuda').manual_seed(11)
    sizes = [1, 7, 1024, 1025, 2500, 50, 400, 3000, 12, 5000] * 4
    p = [torch.randn(n, device='cuda', generator=g) for n in sizes]
    m = [torch.randn(n, device='cuda', generator=g) * 0.1 for n in sizes]
    v = [torch.rand(n, device='cuda', generator=g) * 0.01 for n in sizes]
    gr = [torch.randn(n, device='cuda', generator=g) for n in sizes]
    b1, b2, eps, lr_t = 0.9, 0.999, 1e-8, 3.7e-3
    n = len(sizes)
    arr = ctypes.c_void_p * n
    for dev_lr in (False, True):
        p0, m0, v0 = [t.clone() for t in p], [t.clone() for t in m], [t.clone() for t in v]
        lr_dev = torch.full((), lr_t, device='cuda') if dev_lr else None
        L.check(L.lib().vmp_adam_step(n, arr(*[t.data_ptr() for t in p0]), arr(*[t.data_ptr() for t in gr]),
                                      arr(*[t.data_ptr() for t in m0]), arr(*[t.data_ptr() for t in v0]),
                                      (ctypes.c_int64 * n)(*sizes), b1, b2, eps, 0.0 if dev_lr else lr_t, L.ptr(lr_dev),
                                      L.stream()), 'vmp_adam_step')
        cat = lambda ts: torch.cat([t.double() for t in ts])
        mw = cat(m) * b1 + (1 - b1) * cat(gr)
        vw = cat(v) * b2 + (1 - b2) * cat(gr) ** 2
        pw = cat(p) - lr_t * mw / (vw.sqrt() + eps)
        # element-wise: fp32 rounding of the operands (the sums cancel in places: errors are relative to the operands)
        assert ((cat(m0) - mw).abs() <= 3e-7 * (cat(m).abs() + cat(gr).abs())).all()
        assert ((cat(v0) - vw).abs() <= 3e-7 * (cat(v).abs() + cat(gr) ** 2)).all()
        assert ((cat(p0) - pw).abs() <= 3e-7 * (cat(p).abs() + lr_t * (mw / (vw.sqrt() + eps)).abs()) + 1e-6 * lr_t * (mw / (vw.sqrt() + eps)).abs()).all()


@pytest.mark.parametrize('dims', [(9, 10, 10, 6, 6, 50), (33, 16, 10, 8, 8, 50), (70, 3, 4, 2, 2, 20)])
def test_fused_elbo_fn_matches_unfused_composition(dims):
    """compute_elbo through FusedElboFn (value, details, every gradient) == the same ELBO composed from the weighted
    decoder function + torch ops; with the announced upstream gradient (-1, recognised by address) and with another one
    (generic rescaling branch)."""
    from test_decoder_gpu import make_case
    from vmp_for_svae_amd.models import _svae_ops, svae, vae
    N, K, S, Ld, Dy, U = dims
    x, y, r, w = make_case(N, K, S, Ld, Dy, U, seed=sum(dims))
    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device='cuda')
    g = torch.Generator(device='cuda').manual_seed(3)

    def leaves():
        xs = T(x).requires_grad_(True)
        lz = torch.log_softmax(torch.randn(N, K, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5)), -1).requires_grad_(True)
        Tp = (torch.randn(N, K, device='cuda', generator=torch.Generator(device='cuda').manual_seed(6)) - 3).requires_grad_(True)
        ws = [T(a).requires_grad_(True) for a in w]
        return xs, lz, Tp, ws
    yt = T(y)
    # un-fused composition (the formulation compute_elbo used before the fused tail existed)
    xs, lz, Tp, ws = leaves()
    r_nk = torch.exp(lz)
    rec = vae.expected_diagonal_gaussian_loglike(yt, vae.LazyReconstruction(xs, ws), None, weights=r_nk)
    reg = (r_nk * (Tp + lz)).sum()
    want = (rec - reg, rec, reg)
    want_g = torch.autograd.grad(-(rec - reg), [xs, lz, Tp] + ws)
    for seed, upstream in ((_svae_ops.GradSeed(-1.0, 'cuda'), None), (None, -1.0), (_svae_ops.GradSeed(-1.0, 'cuda'), 0.37)):
        xs, lz, Tp, ws = leaves()
        elbo, rec2, reg2, r2 = _svae_ops.FusedElboFn.apply(yt, xs, lz, Tp, None if seed is None else seed.tensor,
                                                           1.0 if seed is None else seed.value, *ws)
        assert abs(elbo.item() - want[0].item()) <= 2e-6 * abs(want[0].item())
        assert abs(rec2.item() - want[1].item()) <= 2e-6 * abs(want[1].item())
        assert abs(reg2.item() - want[2].item()) <= 2e-6 * abs(want[2].item())
        assert rel(r2, torch.exp(lz)) < 3e-7 and not r2.requires_grad and not rec2.requires_grad
        go = seed.tensor if upstream is None else torch.full((), upstream, device='cuda')
        got_g = torch.autograd.grad(elbo, [xs, lz, Tp] + ws, grad_outputs=go)
        f = 1.0 if upstream is None or upstream == -1.0 else -upstream
        for a, b in zip(got_g, want_g):
            assert rel(a, b * f) < 2e-5
