"""Times the fused decoder kernels (fwd, bwd) at a C3-sized chunk; prints ms and MFMA-rate fractions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import _svae_ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K, S, Ld, Dy, U = 16, 10, 8, 8, int(sys.argv[2]) if len(sys.argv) > 2 else 50
g = torch.Generator(device='cuda').manual_seed(3)
x = torch.randn(N, K, S, Ld, device='cuda', generator=g).requires_grad_(True)
y = torch.randn(N, Dy, device='cuda', generator=g)
r = torch.rand(N, K, device='cuda', generator=g)
shapes = ((Ld, U), (U,), (U, U), (U,), (U, 2 * Dy), (2 * Dy,), (Ld, Dy), (Dy,), (Dy,))
w = [(torch.randn(s, device='cuda', generator=g) * 0.2).requires_grad_(True) for s in shapes]
ev = lambda: torch.cuda.Event(enable_timing=True)
tf, tb = [], []
for it in range(6):
    e0, e1, e2 = ev(), ev(), ev()
    e0.record()
    A = _svae_ops.DecoderLoglikeFn.apply(y, x, *w)
    e1.record()
    gr = torch.autograd.grad(A, [x] + w, r)
    e2.record()
    torch.cuda.synchronize()
    tf.append(e0.elapsed_time(e1)); tb.append(e1.elapsed_time(e2))
rows = N * K * S
flop_f = 2.0 * rows * (Ld * U + U * U + U * 2 * Dy + Ld * Dy)
print('rows %d U %d  fwd %.3f ms (%.1f TF useful)  bwd %.3f ms (%.1f TF useful, 3x fwd flops incl. recompute)'
      % (rows, U, min(tf), flop_f / min(tf) / 1e9, min(tb), 3 * flop_f / min(tb) / 1e9))
