"""T1 step timeline from in-kernel wall-clock stamps (100 MHz, comparable across kernels; -DVMP_DEBUG_TS build): finalize entry / end,
pass entry / prologue done / end of block 0 - i.e. where the ~20 us of a 53 us step that are not streaming go.
usage: VMP_LIB_PATH=.../libvmp_hip_ts.so python tools/t1_timeline.py [N ...]"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import _mix
L = V._lib
D, K = 8, 16
h = ctypes.CDLL(L.LIB_PATH)
for N in [int(a) for a in sys.argv[1:]] or (2048, 125000, 1000000):
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(N, D, device='cuda', generator=g) * 3
    r0 = torch.softmax(3 * torch.randn(N, K, device='cuda', generator=g), 1)
    loop = _mix.VMPLoop(x, r0, L.VMP_GMM)
    for _ in range(20): loop.step()
    tp = torch.zeros(128, dtype=torch.int64, device='cuda')
    tf = torch.zeros(8, dtype=torch.int64, device='cuda')
    h.vmp_debug_set_pass_timestamps(ctypes.c_void_p(tp.data_ptr()))
    h.vmp_debug_set_finalize_timestamps(ctypes.c_void_p(tf.data_ptr()))
    rows = []
    for it in range(6):
        loop.step()                      # finalize (reads the partials of the previous pass) -> pass
        torch.cuda.synchronize()
        p, f = tp.cpu().tolist(), tf.cpu().tolist()
        rows.append((f[6], f[7], p[7], p[15], f[5] - f[0], p[1] - p[0], p[2] - p[0], p[5] - p[0]))
    print('N=%d  (wall clock in us relative to the finalize entry; cycles of block 0 / wave 0)' % N)
    prev_end = None
    for (fe, fx, pe, px, fcyc, pro, loopend, pend) in rows[1:]:
        line = '  finalize %.2f us | gap %.2f us | pass entry -> end %.2f us (prologue until rows+pack arrived %d cyc, loop end %d cyc, partials written %d cyc; finalize %d cyc)' % (
            (fx - fe) / 100.0, (pe - fx) / 100.0, (px - pe) / 100.0, pro, loopend, pend, fcyc)
        print(line)
    h.vmp_debug_set_pass_timestamps(None); h.vmp_debug_set_finalize_timestamps(None)
