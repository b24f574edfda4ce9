"""Stage stamps (clock64) of ONE streaming tile (16 rows) of the fused decoder backward kernel - the 9th tile of block 0, wave 0 - in a
-DVMP_DEBUG_TS build:  tools/build_variant.sh ts "-DVMP_DEBUG_TS" vmp_decoder.hip ...;  VMP_LIB_PATH=.../libvmp_hip_ts.so python tools/dec_tile_ts.py"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import _svae_ops
L = V._lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
K, S, Ld, Dy, U = 16, 10, 8, 8, int(sys.argv[2]) if len(sys.argv) > 2 else 50
g = torch.Generator(device='cuda').manual_seed(3)
x = torch.randn(N, K, S, Ld, device='cuda', generator=g).requires_grad_(True)
y = torch.randn(N, Dy, device='cuda', generator=g)
r = torch.rand(N, K, device='cuda', generator=g)
shapes = ((Ld, U), (U,), (U, U), (U,), (U, 2 * Dy), (2 * Dy,), (Ld, Dy), (Dy,), (Dy,))
w = [(torch.randn(s, device='cuda', generator=g) * 0.2).requires_grad_(True) for s in shapes]
ts = torch.zeros(128, dtype=torch.int64, device='cuda')
h = ctypes.CDLL(L.LIB_PATH); h.vmp_debug_set_decoder_timestamps(ctypes.c_void_p(ts.data_ptr()))
names = ['inputs of the tile, x split + transposed', 'forward recompute (3 layers) + transposes', 'reconstruction term, dO, split',
         'dh1 = W2 . dO', 'dW2, dWs', 'tanh of layer 1', 'dh0 = W1 . dh1pre (+ split, transpose)', 'dW1', 'tanh of layer 0',
         'dx = W0 . dh0pre + Ws . dO, store', 'dW0']
for it in range(3):
    A = _svae_ops.DecoderWeightedLoglikeFn.apply(y, x, r, *w) if hasattr(_svae_ops, 'DecoderWeightedLoglikeFn') else None
    torch.cuda.synchronize()
    t = ts.cpu().tolist()
    if it == 0:
        continue
    print('U=%d rows=%d: tile total %d cycles' % (U, N * K * S, t[64 + 11] - t[64]))
    for i, n in enumerate(names):
        print('   %-46s %6d' % (n, t[64 + i + 1] - t[64 + i]))
