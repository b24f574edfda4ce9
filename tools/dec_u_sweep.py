"""Fused decoder forward/backward against a plain torch fp32/fp64 MLP for a sweep of hidden widths U (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import _svae_ops


def ref(y, x, w, dt):
    W0, b0, W1, b1, W2, b2, Ws, bs1, bs2 = [t.to(dt) for t in w]
    x = x.to(dt)
    h0 = torch.tanh(x @ W0 + b0)
    h1 = torch.tanh(h0 @ W1 + b1)
    o = h1 @ W2 + b2
    Dy = Ws.shape[1]
    mean = o[..., :Dy] + x @ Ws + bs1
    var = torch.nn.functional.softplus(o[..., Dy:]) + torch.log1p(torch.exp(bs2))
    ll = ((y.to(dt)[:, None, None, :] - mean) ** 2 / var + torch.log(var + 1e-8)).sum(-1)
    return ll.sum(-1)


N, K, S, Ld, Dy = 37, 5, 3, 5, 8
for U in [int(a) for a in sys.argv[1:]] or [1, 8, 16, 17, 20, 32, 33, 40, 47, 48, 49, 50, 63, 64]:
    g = torch.Generator(device='cuda').manual_seed(U)
    x = (torch.randn(N, K, S, Ld, device='cuda', generator=g) * 1.5).requires_grad_(True)
    y = torch.randn(N, Dy, device='cuda', generator=g)
    r = torch.rand(N, K, device='cuda', generator=g) + 0.05
    shapes = ((Ld, U), (U,), (U, U), (U,), (U, 2 * Dy), (2 * Dy,), (Ld, Dy), (Dy,), (Dy,))
    w = [(torch.randn(s, device='cuda', generator=g) * 0.3).requires_grad_(True) for s in shapes]
    A = _svae_ops.DecoderLoglikeFn.apply(y, x, *w)
    gr = torch.autograd.grad(A, [x] + w, r)
    A64 = ref(y, x, w, torch.float64)
    g64 = torch.autograd.grad(A64, [x] + w, r.double())
    A32 = ref(y, x, w, torch.float32)
    g32 = torch.autograd.grad(A32, [x] + w, r)
    rel = lambda a, b: ((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()
    print('U %2d  A: hip %.2e torch32 %.2e | grads hip max %.2e (W1 %.2e b1 %.2e) torch32 max %.2e'
          % (U, rel(A, A64), rel(A32, A64), max(rel(a, b) for a, b in zip(gr, g64)), rel(gr[3], g64[3]), rel(gr[4], g64[4]),
             max(rel(a, b) for a, b in zip(g32, g64))))
