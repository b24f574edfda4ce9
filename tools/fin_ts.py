import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import _mix
L = V._lib
N, D, K = 1000000, 8, 16
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(N, D, device='cuda', generator=g) * 3
r0 = torch.softmax(3 * torch.randn(N, K, device='cuda', generator=g), 1)
loop = _mix.VMPLoop(x, r0, L.VMP_GMM)
for _ in range(5): loop.step()
ts = torch.zeros(8, dtype=torch.int64, device='cuda')
h = ctypes.CDLL(L.LIB_PATH); h.vmp_debug_set_finalize_timestamps(ctypes.c_void_p(ts.data_ptr()))
for _ in range(3):
    loop.step(); torch.cuda.synchronize()
    t = ts.cpu().tolist()
    print('cycles: loads+reduce', t[1]-t[0], '| combine..st', t[2]-t[1], '| phaseB', t[3]-t[2], '| phaseC', t[4]-t[3], '| phaseD+pack', t[5]-t[4], '| total', t[5]-t[0])
print('--- finalize called again on the same (not freshly written) partials')
for _ in range(3):
    loop.finalize(); torch.cuda.synchronize()
    t = ts.cpu().tolist()
    print('cycles: loads+reduce', t[1]-t[0], '| total', t[5]-t[0])
