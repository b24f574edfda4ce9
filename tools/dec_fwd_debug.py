import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import _svae_ops
Ld, Dy = 5, 8
for U in (40, 48, 33):
    g = torch.Generator(device='cuda').manual_seed(U)
    x = torch.randn(64, Ld, device='cuda', generator=g) * 1.5
    shapes = ((Ld, U), (U,), (U, U), (U,), (U, 2 * Dy), (2 * Dy,), (Ld, Dy), (Dy,), (Dy,))
    w = [(torch.randn(s, device='cuda', generator=g) * 0.3) for s in shapes]
    for trial in range(2):
        mean, var = _svae_ops.decoder_outputs(x, w)
        W0, b0, W1, b1, W2, b2, Ws, bs1, bs2 = [t.double() for t in w]
        xd = x.double()
        h0 = torch.tanh(xd @ W0 + b0); h1 = torch.tanh(h0 @ W1 + b1); o = h1 @ W2 + b2
        m_ref = o[:, :Dy] + xd @ Ws + bs1
        err = (mean.double() - m_ref).abs()
        print('U', U, 'trial', trial, 'max err', err.max().item())
        print('  per row (first 32):', ['%.0e' % e for e in err.max(1).values[:32].tolist()])
        print('  per dim:', ['%.0e' % e for e in err.max(0).values.tolist()])
    # which hidden units matter: zero W2 rows one tile at a time
    for t in range((U + 15) // 16):
        w2 = [a.clone() for a in w]
        w2[4][:16 * t] = 0; w2[4][16 * (t + 1):] = 0
        mean, var = _svae_ops.decoder_outputs(x, w2)
        W0, b0, W1, b1, W2, b2, Ws, bs1, bs2 = [a.double() for a in w2]
        h0 = torch.tanh(xd @ W0 + b0); h1 = torch.tanh(h0 @ W1 + b1); o = h1 @ W2 + b2
        m_ref = o[:, :Dy] + xd @ Ws + bs1
        print('  only h1 tile', t, 'feeding the output: max err', (mean.double() - m_ref).abs().max().item())
    for t in range((U + 15) // 16):
        w2 = [a.clone() for a in w]
        w2[2][:16 * t] = 0; w2[2][16 * (t + 1):] = 0
        mean, var = _svae_ops.decoder_outputs(x, w2)
        W0, b0, W1, b1, W2, b2, Ws, bs1, bs2 = [a.double() for a in w2]
        h0 = torch.tanh(xd @ W0 + b0); h1 = torch.tanh(h0 @ W1 + b1); o = h1 @ W2 + b2
        m_ref = o[:, :Dy] + xd @ Ws + bs1
        print('  only h0 tile', t, 'feeding layer 1: max err', (mean.double() - m_ref).abs().max().item())
