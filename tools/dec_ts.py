"""Stage time stamps of the fused decoder backward kernel (block 0, thread 0) in a -DVMP_DEBUG_TS build:
   tools/build_variant.sh ts "-DVMP_DEBUG_TS" vmp_svae.hip vmp_decoder.hip ; VMP_LIB_PATH=.../libvmp_hip_ts.so python tools/dec_ts.py [N]"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import vae
from vmp_for_svae_amd.training import SVAETrainer
L = V._lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
vae.reset_variables()
y = torch.randn(N, 6, device='cuda')
tr = SVAETrainer(10, 8, 50, 6, nb_samples=10)
ts = torch.zeros(64, dtype=torch.int64, device='cuda')
h = ctypes.CDLL(L.LIB_PATH); h.vmp_debug_set_decoder_timestamps(ctypes.c_void_p(ts.data_ptr()))
names = ['weight images filled', 'barrier + accumulators zeroed', 'tile loop', 'slabs + block sum', 'partials written']
for it in range(4):
    tr.step(y)          # the LAST decoder-kernel launch of a step is the encoder's backward (R = N rows)
    torch.cuda.synchronize()
    t = ts.cpu().tolist()
    if it == 0:
        continue
    print('encoder backward (R = %d): total %d cycles = %.2f us' % (N, t[5] - t[0], (t[32 + 5] - t[32]) / 100.0))
    for i, n in enumerate(names):
        print('   %-30s %7d cycles  %6.2f us' % (n, t[i + 1] - t[i], (t[32 + i + 1] - t[32 + i]) / 100.0))
    print('   epilogue: own slab stored %d, barrier %d, sum loop + barrier %d cycles' % (t[6] - t[3], t[7] - t[6], t[4] - t[7]))
