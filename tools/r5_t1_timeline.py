"""Timeline of the ONE-LAUNCH T1 step from in-kernel stamps (-DVMP_DEBUG_TS build): wall clock (100 MHz) of block 0 - launch entry,
head (finalize_block) entry / end, last stamp of the launch - and shader-clock cycles of block 0 / wave 0 between the pass stamps.
usage: VMP_LIB_PATH=.../libvmp_hip_ts.so python tools/r5_t1_timeline.py [N ...]"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import _mix
L = V._lib
D, K = 8, 16
h = ctypes.CDLL(L.LIB_PATH)
for N in [int(a) for a in sys.argv[1:]] or (125000, 1000000):
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(N, D, device='cuda', generator=g) * 3
    r0 = torch.softmax(3 * torch.randn(N, K, device='cuda', generator=g), 1)
    for mode in (False, True):
        loop = _mix.VMPLoop(x, r0, L.VMP_GMM, one_launch=mode)
        for _ in range(20): loop.step()
        tp = torch.zeros(128, dtype=torch.int64, device='cuda')
        tf = torch.zeros(8, dtype=torch.int64, device='cuda')
        h.vmp_debug_set_pass_timestamps(ctypes.c_void_p(tp.data_ptr()))
        h.vmp_debug_set_finalize_timestamps(ctypes.c_void_p(tf.data_ptr()))
        print('N=%d one_launch=%s' % (N, mode))
        prev_end = None
        for it in range(5):
            loop.run(3)
            torch.cuda.synchronize()
            p, f = tp.cpu().tolist(), tf.cpu().tolist()
            # wall: f[6] head entry, f[7] head end, p[7] launch entry (block 0), p[15] end of block 0 (stamp 5); p[64+7] launch entry of block 100, p[64+15] its end
            e0 = p[7]
            print('   launch entry 0 | head entry %+.2f us, head end %+.2f us | block 0 end %+.2f us | block 100: entry %+.2f end %+.2f | cycles b0w0: rows+pack arrived %d, loop end %d, partials written %d | b100w0: pack arrived %d, loop end %d; head %d cyc'
                  % ((f[6] - e0) / 100., (f[7] - e0) / 100., (p[15] - e0) / 100., (p[64 + 7] - e0) / 100., (p[64 + 15] - e0) / 100.,
                     p[1] - p[0], p[2] - p[0], p[5] - p[0], p[64 + 1] - p[64], p[64 + 2] - p[64], f[5] - f[0]) + ' [head phases: partial sums %d, moments %d, S/C %d, Cholesky+digamma %d, pack %d]' % tuple(f[i + 1] - f[i] for i in range(5)))
        h.vmp_debug_set_pass_timestamps(None); h.vmp_debug_set_finalize_timestamps(None)
