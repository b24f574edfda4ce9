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
    blocks, grid-stride), launched twice on the same workspace - which holds OTHER data's partials from the launch before."""
    import vmp_for_svae_amd as V
    L = V._lib
    g = torch.Generator(device='cuda').manual_seed(N + K)
    lz = torch.log_softmax(torch.randn(N, K, device='cuda', generator=g) * 2, -1)
    Tp = torch.randn(N, K, device='cuda', generator=g) * 3 - 5
    ll = torch.randn(N, K, S, device='cuda', generator=g) * 4 + 10
    ws = torch.empty(L.lib().vmp_svae_elbo_tail_workspace_bytes(), dtype=torch.uint8, device='cuda')
    ws.view(torch.float64).fill_(1e30)                       # whatever a previous launch left there must not matter
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


def test_pack_f64_and_packed_adam_kernels():
    """The data-parallel step's exchange (experiments.py:247-265, tf_utils.py:52-87): vmp_pack_f64 writes [moments (fp64) | all
    gradients (fp32 -> fp64) | scalars] with one launch; vmp_adam_step_packed applies Adam with gradient = gscale * buffer slice
    (the mean over the ranks, formed in fp64) and stores the averaged fp32 gradient.  Against torch; 40 tensors = two batches."""
    import vmp_for_svae_amd as V
    from vmp_for_svae_amd import training
    L = V._lib
    g = torch.Generator(device='cuda').manual_seed(12)
    sizes = [1, 7, 1024, 1025, 2500, 50, 400, 3000, 12, 5000] * 4
    stats = torch.randn(16, 74, device='cuda', generator=g, dtype=torch.float64)
    grads = [torch.randn(n, device='cuda', generator=g) for n in sizes]
    scal = [torch.randn((), device='cuda', generator=g, dtype=torch.float64), torch.randn((), device='cuda', generator=g),
            torch.randn((), device='cuda', generator=g, dtype=torch.float64)]
    buf, goffs = training.pack_exchange_buffer(stats, grads, scal)
    want = training.pack_for_allreduce(stats, grads, scal)
    assert buf.dtype == torch.float64 and torch.equal(buf, want)
    assert goffs[0] == stats.numel() and goffs[1] == stats.numel() + sizes[0]
    # two "ranks": the all-reduced buffer is the sum of two packs; Adam from it with gscale = 1/2
    grads_b = [torch.randn(n, device='cuda', generator=g) for n in sizes]
    buf2, _ = training.pack_exchange_buffer(stats, grads_b, scal)
    tot = buf + buf2
    params = [torch.nn.Parameter(torch.randn(n, device='cuda', generator=g)) for n in sizes]
    opt_a, opt_b = training.TFAdam(params, 3e-3), None
    ref_params = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt_b = training.TFAdam(ref_params, 3e-3)
    for step in range(2):
        gout = [torch.empty_like(p) for p in params]
        opt_a.apply_packed(tot, goffs, 0.5, gout)
        mean = [((a.double() + b.double()) * 0.5).float() for a, b in zip(grads, grads_b)]
        opt_b.apply_gradients(mean)
        for go, mg in zip(gout, mean):
            assert torch.equal(go, mg)
        for pa, pb in zip(params, ref_params):
            assert torch.equal(pa, pb)                      # same kernel arithmetic on the same fp32 gradient
    assert opt_a.t == opt_b.t == 2


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


@pytest.mark.parametrize('N,K,Ld', [(64, 10, 8), (100, 16, 6), (1, 3, 2), (512, 5, 8)])
def test_stats_cvi_one_launch_equals_the_two(N, K, Ld):
    """vmp_svae_stats_cvi (M-step moments + CVI update in one launch) == vmp_mix_stats followed by vmp_svae_cvi_update,
    bit for bit, with the step size by value and through the device word."""
    from vmp_for_svae_amd.models import _mix, _svae_ops, svae
    g = torch.Generator(device='cuda').manual_seed(N + K)
    x = torch.randn(N, Ld, device='cuda', generator=g) * 2
    r = torch.softmax(torch.randn(N, K, device='cuda', generator=g), -1)
    prior, theta0 = svae.init_mm(K, Ld, seed=1)
    for rho_dev in (None, torch.full((), 0.17, device='cuda')):
        th_a = [t.clone() for t in theta0]
        th_b = [t.clone() for t in theta0]
        st_a = _mix.raw_stats(x, r)
        star_a = _svae_ops.cvi_update(prior, th_a, st_a, 0.17 if rho_dev is None else 0.0, rho_dev=rho_dev)
        st_b, star_b = _svae_ops.stats_cvi(x, r, prior, th_b, 0.17 if rho_dev is None else 0.0, rho_dev=rho_dev)
        assert torch.equal(st_a, st_b)
        for a, b in zip(th_a + star_a, th_b + star_b):
            assert torch.equal(a, b)
    import vmp_for_svae_amd as V
    with pytest.raises(V._lib.VmpError):                  # larger batches take the two-call form
        _svae_ops.stats_cvi(torch.randn(513, Ld, device='cuda'), torch.rand(513, K, device='cuda'), prior, [t.clone() for t in theta0], 0.1)


@pytest.mark.parametrize('K,Ld', [(10, 8), (16, 6), (3, 1), (64, 8)])
def test_prep_one_launch_equals_the_two(K, Ld):
    """PhiPrepFn with theta (vmp_svae_prep_fwd: recognition unpacking + theta packing in one launch) == PhiPrepFn without
    + theta_pack_gmm, bit for bit; gradients w.r.t. the recognition parameters unchanged."""
    from vmp_for_svae_amd.models import _svae_ops, svae
    prior, theta = svae.init_mm(K, Ld, seed=2)
    g = torch.Generator(device='cuda').manual_seed(K)
    with torch.no_grad():
        theta[1].add_(0.3 * torch.eye(Ld, device='cuda'))
        theta[2].add_(torch.randn(K, Ld, device='cuda', generator=g) * 0.1)
    phi = [p.detach().clone().requires_grad_(True) for p in svae.init_recognition_params(theta, K, seed=2)]
    a = _svae_ops.PhiPrepFn.apply(*phi)
    m, W, kap = _svae_ops.theta_pack_gmm(theta)
    phi2 = [p.detach().clone().requires_grad_(True) for p in phi]
    b = _svae_ops.PhiPrepFn.apply(*phi2, *theta)
    for u, v in zip(list(a) + [m, W, kap], b):
        assert torch.equal(u, v)
    assert not b[3].requires_grad and not b[5].requires_grad
    G = [torch.randn_like(t) for t in a]
    ga = torch.autograd.grad(list(a), phi, G)
    gb = torch.autograd.grad(list(b[:3]), phi2, G)
    for u, v in zip(ga, gb):
        assert torch.equal(u, v)


def test_gauss_head_scale_inside_the_kernels():
    """GaussMLPFn with the 'natparam' head scale (-1/2, applied inside the forward / backward kernels) == the 'standard'
    head followed by a torch multiplication: outputs and every gradient."""
    from test_decoder_gpu import make_case
    from vmp_for_svae_amd.models import _svae_ops
    N, K, S, Ld, Dy, U = 70, 1, 1, 6, 8, 50
    x, y, r, w = make_case(N, K, S, Ld, Dy, U, seed=9)
    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device='cuda')
    xs = T(x).reshape(N, Ld)
    g = torch.Generator(device='cuda').manual_seed(4)
    G1, G2 = torch.randn(N, Dy, device='cuda', generator=g), torch.randn(N, Dy, device='cuda', generator=g)
    res = []
    for fused in (False, True):
        xi = xs.clone().requires_grad_(True)
        ws = [T(a).requires_grad_(True) for a in w]
        if fused:
            o1, o2 = _svae_ops.GaussMLPFn.apply(xi, -0.5, *ws)
        else:
            o1, v = _svae_ops.GaussMLPFn.apply(xi, 1.0, *ws)
            o2 = -0.5 * v
        gr = torch.autograd.grad([o1, o2], [xi] + ws, [G1, G2])
        res.append([o1, o2] + list(gr))
    for a, b in zip(*res):
        assert rel(b, a) < 3e-6


@pytest.mark.parametrize('N,K,Ld,S', [(64, 10, 8, 10), (37, 7, 4, 8), (5, 3, 2, 4), (100, 16, 8, 16), (512, 8, 6, 10), (1, 1, 1, 1)])
def test_bwd_tail_equals_tail_plus_backward(N, K, Ld, S):
    """Round 6: vmp_svae_estep_bwd_tail (the ELBO's scalar tail on a wave of the minibatch-form E-step backward; S = 16: no spare wave, wave
    0 runs it) against vmp_svae_elbo_tail followed by vmp_svae_estep_bwd_n on the same inputs: gradients, partial rows and r
    BIT-identical (dLoss/dlog_z, dLoss/dT' never reach memory in the fused form); the three scalars from the per-tile fp64 sums."""
    import vmp_for_svae_amd as V
    L = V._lib
    lib = L.lib()
    Dy = 6
    f32 = dict(dtype=torch.float32, device='cuda')
    g = torch.Generator(device='cuda').manual_seed(N * 7 + K)
    rn = lambda *s: torch.randn(*s, generator=g, **f32)
    x, Gx = rn(N, K, S, Ld), rn(N, K, S, Ld) * 0.3
    lz = torch.log_softmax(rn(N, K) * 2, -1)
    Tp, ll = rn(N, K) * 3 - 5, rn(N, K, S) * 3 + 8
    eta1, eta2d = rn(N, Ld), -torch.rand(N, Ld, generator=g, **f32) - 0.5
    hk, A_ = rn(K, Ld), rn(K, Ld, Ld) * 0.4
    Pk = (A_ @ A_.transpose(1, 2) + torch.eye(Ld, **f32)).contiguous()
    bias, mk, Wk = rn(K), rn(K, Ld), torch.tril(rn(K, Ld, Ld)).contiguous()
    assert lib.vmp_svae_bwd_tail_applies(N, K, Ld, S) == 1
    scal = torch.empty(3, **f32)
    g_lz, g_Tp, r = torch.empty(N, K, **f32), torch.empty(N, K, **f32), torch.empty(N, K, **f32)
    ws = torch.empty(lib.vmp_svae_elbo_tail_workspace_bytes(), dtype=torch.uint8, device='cuda')
    L.check(lib.vmp_svae_elbo_tail(L.ptr(lz), L.ptr(Tp), L.ptr(ll), N, K, S, Dy, -1.0, L.ptr(scal), L.ptr(g_lz), L.ptr(g_Tp), L.ptr(r), L.ptr(ws),
                                   ws.numel(), L.stream()), 'tail')
    nt, PW = lib.vmp_svae_bwd_blocks_for(N, K, Ld, S, 0), lib.vmp_svae_bwd_partial_words(Ld)
    assert nt == (N + 64 // K - 1) // (64 // K)
    a1, a2, ap = torch.empty(N, Ld, **f32), torch.empty(N, Ld, **f32), torch.full((nt, K, PW), float('nan'), **f32)
    L.check(lib.vmp_svae_estep_bwd_n(L.ptr(eta1), L.ptr(eta2d), L.ptr(hk), L.ptr(Pk), L.ptr(bias), L.ptr(mk), L.ptr(Wk), None, L.ptr(x), L.ptr(lz),
                                     L.ptr(Gx), L.ptr(g_lz), L.ptr(g_Tp), N, K, Ld, S, L.ptr(a1), L.ptr(a2), L.ptr(ap), ap.numel() * 4, nt, L.stream()),
            'bwd_n')
    b1, b2, bp = torch.empty(N, Ld, **f32), torch.empty(N, Ld, **f32), torch.full((nt, K, PW), float('nan'), **f32)
    r2, tp = torch.full((N, K), float('nan'), **f32), torch.full((nt, 2), float('nan'), dtype=torch.float64, device='cuda')
    L.check(lib.vmp_svae_estep_bwd_tail(L.ptr(eta1), L.ptr(eta2d), L.ptr(hk), L.ptr(Pk), L.ptr(bias), L.ptr(mk), L.ptr(Wk), L.ptr(x), L.ptr(lz),
                                        L.ptr(Tp), L.ptr(ll), -1.0, L.ptr(Gx), N, K, Ld, S, L.ptr(b1), L.ptr(b2), L.ptr(bp), bp.numel() * 4, L.ptr(r2),
                                        L.ptr(tp), tp.numel() * 8, L.stream()), 'bwd_tail')
    assert torch.equal(a1, b1) and torch.equal(a2, b2) and torch.equal(ap, bp) and torch.equal(r, r2)
    tot = tp.sum(0)
    rec = -tot[0].item() - N * Dy * 0.5 * math.log(2 * math.pi)
    want = scal.double().cpu()
    got = torch.tensor([rec - tot[1].item(), rec, tot[1].item()], dtype=torch.float64)
    assert ((got - want).abs().max() / max(abs(want[1].item()), abs(want[2].item()))).item() < 2e-7


@pytest.mark.parametrize('K,Ld,nblk', [(10, 8, 11), (16, 8, 1), (7, 4, 70), (3, 2, 300), (5, 5, 17)])
def test_bwd_reduce_prep_equals_reduce_then_prep_backward(K, Ld, nblk):
    """Round 6: vmp_svae_bwd_reduce_prep (block k sums component k's partial rows and differentiates the recognition unpacking on them;
    log softmax(pi) from the forward's vmp_svae_prep_fwd2) against vmp_svae_bwd_reduce + vmp_svae_phi_prep_bwd: bit-identical."""
    import vmp_for_svae_amd as V
    L = V._lib
    lib = L.lib()
    f32 = dict(dtype=torch.float32, device='cuda')
    g = torch.Generator(device='cuda').manual_seed(K * 31 + nblk)
    rn = lambda *s: torch.randn(*s, generator=g, **f32)
    PW = lib.vmp_svae_bwd_partial_words(Ld)
    partials = rn(nblk, K, PW)
    mu, Lraw, pi = rn(K, Ld), rn(K, Ld, Ld), rn(K)
    th = [torch.rand(K, generator=g, **f32) + 0.5, None, rn(K, Ld), torch.rand(K, generator=g, **f32) + 1.0, torch.rand(K, generator=g, **f32) + Ld + 3.0]
    A_ = rn(K, Ld, Ld)
    th[1] = (A_ @ A_.transpose(1, 2) + Ld * torch.eye(Ld, **f32) + th[2].unsqueeze(2) * th[2].unsqueeze(1) / th[3].view(K, 1, 1)).contiguous()
    Lk, P, bias = torch.empty(K, Ld, Ld, **f32), torch.empty(K, Ld, Ld, **f32), torch.empty(K, **f32)
    m, W, kap = torch.empty(K, Ld, **f32), torch.empty(K, Ld, Ld, **f32), torch.empty(K, **f32)
    logpi = torch.empty(K, dtype=torch.float64, device='cuda')
    L.check(lib.vmp_svae_prep_fwd2(L.ptr(mu), L.ptr(Lraw), L.ptr(pi), *[L.ptr(t) for t in th], K, Ld, L.ptr(Lk), L.ptr(P), L.ptr(bias), L.ptr(m),
                                   L.ptr(W), L.ptr(kap), L.ptr(logpi), L.stream()), 'prep_fwd2')
    assert torch.allclose(logpi, torch.log_softmax(pi.double(), 0), rtol=0, atol=1e-14)
    # the forward itself equals the launch without the extra output
    P0, b0 = torch.empty_like(P), torch.empty_like(bias)
    L.check(lib.vmp_svae_prep_fwd(L.ptr(mu), L.ptr(Lraw), L.ptr(pi), *[L.ptr(t) for t in th], K, Ld, L.ptr(torch.empty_like(Lk)), L.ptr(P0), L.ptr(b0),
                                  L.ptr(torch.empty_like(m)), L.ptr(torch.empty_like(W)), L.ptr(torch.empty_like(kap)), L.stream()), 'prep_fwd')
    assert torch.equal(P, P0) and torch.equal(bias, b0)
    g_hk, g_P, g_b = torch.empty(K, Ld, **f32), torch.empty(K, Ld, Ld, **f32), torch.empty(K, **f32)
    L.check(lib.vmp_svae_bwd_reduce(L.ptr(partials), nblk, K, Ld, L.ptr(g_hk), L.ptr(g_P), L.ptr(g_b), None, None, None, L.stream()), 'reduce')
    w = [torch.empty_like(mu), torch.empty_like(Lraw), torch.empty_like(pi)]
    L.check(lib.vmp_svae_phi_prep_bwd(L.ptr(mu), L.ptr(Lraw), L.ptr(pi), L.ptr(g_hk), L.ptr(g_P), L.ptr(g_b), K, Ld, *[L.ptr(t) for t in w],
                                      L.stream()), 'prep_bwd')
    o = [torch.full_like(mu, float('nan')), torch.full_like(Lraw, float('nan')), torch.full_like(pi, float('nan'))]
    L.check(lib.vmp_svae_bwd_reduce_prep(L.ptr(partials), nblk, L.ptr(mu), L.ptr(Lraw), L.ptr(pi), L.ptr(logpi), K, Ld, *[L.ptr(t) for t in o],
                                         L.stream()), 'reduce_prep')
    for a, b in zip(o, w):
        assert torch.equal(a, b)


def test_step_inputs_and_the_first_launch_with_a_scalar_table():
    """vmp_svae_step_inputs (scalars + minibatch copy in one launch) and vmp_mlp_gauss_head_fwd_prep (encoder forward + recognition /
    theta prep + row `counter` of the scalar table -> the step's 16 bytes, counter advanced): against the stand-alone launches."""
    import vmp_for_svae_amd as V
    L = V._lib
    lib = L.lib()
    f32 = dict(dtype=torch.float32, device='cuda')
    g = torch.Generator(device='cuda').manual_seed(9)
    rn = lambda *s: torch.randn(*s, generator=g, **f32)
    dst = torch.zeros(16, dtype=torch.uint8, device='cuda')
    src, y = rn(1000, 7), torch.zeros(1000, 7, **f32)
    L.check(lib.vmp_svae_step_inputs(L.ptr(dst), 0xDEADBEEF12345678, 0.125, 3e-4, L.ptr(src), L.ptr(y), src.numel(), L.stream()), 'step_inputs')
    assert torch.equal(y, src)
    assert dst[:8].view(torch.int64).item() == 0xDEADBEEF12345678 - (1 << 64)
    assert dst[8:12].view(torch.float32).item() == 0.125 and dst[12:].view(torch.float32).item() == np.float32(3e-4)
    for N, Dy, Ld, U, K in ((64, 6, 8, 50, 10), (5, 3, 2, 16, 3), (300, 8, 4, 64, 16)):
        enc = [rn(Dy, U) * 0.3, rn(U) * 0.1, rn(U, U) * 0.2, rn(U) * 0.1, rn(U, 2 * Ld) * 0.2, rn(2 * Ld) * 0.1, rn(Dy, Ld) * 0.3, rn(Ld) * 0.1, rn(Ld) * 0.1]
        yy = rn(N, Dy)
        mu, Lraw, pi = rn(K, Ld), rn(K, Ld, Ld), rn(K)
        th = [torch.rand(K, generator=g, **f32) + 0.5, None, rn(K, Ld), torch.rand(K, generator=g, **f32) + 1.0, torch.rand(K, generator=g, **f32) + Ld + 3.0]
        A_ = rn(K, Ld, Ld)
        th[1] = (A_ @ A_.transpose(1, 2) + Ld * torch.eye(Ld, **f32) + th[2].unsqueeze(2) * th[2].unsqueeze(1) / th[3].view(K, 1, 1)).contiguous()
        mk = lambda: (torch.empty(N, Ld, **f32), torch.empty(N, Ld, **f32), torch.empty(K, Ld, Ld, **f32), torch.empty(K, Ld, Ld, **f32),
                      torch.empty(K, **f32), torch.empty(K, Ld, **f32), torch.empty(K, Ld, Ld, **f32), torch.empty(K, **f32),
                      torch.empty(K, dtype=torch.float64, device='cuda'))
        a = mk()
        L.check(lib.vmp_mlp_gauss_head_fwd(L.ptr(yy), *[L.ptr(t) for t in enc], N, Dy, Ld, U, -0.5, L.ptr(a[0]), L.ptr(a[1]), L.stream()), 'enc')
        L.check(lib.vmp_svae_prep_fwd2(L.ptr(mu), L.ptr(Lraw), L.ptr(pi), *[L.ptr(t) for t in th], K, Ld, *[L.ptr(t) for t in a[2:]], L.stream()), 'prep')
        b = mk()
        rows = 5
        table = torch.zeros(rows, 2, dtype=torch.int64, device='cuda')
        table[:, 0] = torch.arange(100, 100 + rows, device='cuda')
        table[:, 1] = torch.arange(7, 7 + rows, device='cuda') << 32
        counter = torch.full((1,), 3, dtype=torch.int64, device='cuda')
        d16 = torch.zeros(2, dtype=torch.int64, device='cuda')
        for rep in range(3):                                   # rows 3, 4 and - the counter past the table - the last row again
            L.check(lib.vmp_mlp_gauss_head_fwd_prep(L.ptr(yy), *[L.ptr(t) for t in enc], N, Dy, Ld, U, -0.5, L.ptr(b[0]), L.ptr(b[1]), L.ptr(mu), L.ptr(Lraw),
                                                    L.ptr(pi), *[L.ptr(t) for t in th], K, *[L.ptr(t) for t in b[2:]], L.ptr(table), rows, L.ptr(counter),
                                                    L.ptr(d16), L.stream()), 'enc_prep')
            assert counter.item() == 4 + rep and d16[0].item() == 100 + min(3 + rep, rows - 1) and d16[1].item() == (7 + min(3 + rep, rows - 1)) << 32
        for u, v in zip(a, b):
            assert torch.equal(u, v)
