"""Where the fixed cost of the T1 pass kernel goes: clock64 stamps of every wave of blocks 0 and 100 (debug hook)."""
# needs a library built with: make -C vmp-for-svae_amd/csrc EXTRA=-DVMP_DEBUG_TS
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import _mix
L = V._lib
D, K = 8, 16
h = ctypes.CDLL(L.LIB_PATH)
for N in [int(a) for a in sys.argv[1:]] or (2048, 1000000):
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(N, D, device='cuda', generator=g) * 3
    r0 = torch.softmax(3 * torch.randn(N, K, device='cuda', generator=g), 1)
    loop = _mix.VMPLoop(x, r0, L.VMP_GMM)
    for _ in range(5): loop.step()
    ts = torch.zeros(128, dtype=torch.int64, device='cuda')
    h.vmp_debug_set_pass_timestamps(ctypes.c_void_p(ts.data_ptr()))
    loop.step(); torch.cuda.synchronize()
    t = ts.cpu().view(2, 8, 8)
    for b in range(2):
        t0 = int(t[b, :, 0].min())
        print('N=%d block %d: per wave [entry | params+rows | loop end | after block sync | slabs reduced | partials written] relative to the first entry' % (N, (0, 100)[b]))
        for w in range(8):
            hw = int(t[b, w, 6])
            print('   wave %d: ' % w + ' '.join('%7d' % (int(t[b, w, i]) - t0) for i in range(6))
                  + '   hw_id wave %d simd %d pipe %d cu %d sh %d se %d' % (hw & 15, (hw >> 4) & 3, (hw >> 6) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7))
    h.vmp_debug_set_pass_timestamps(None)
