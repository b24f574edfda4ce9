"""T1 loop: eager (2 launches per step from Python) vs one HIP graph holding S steps."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import _mix
x_h, r_h = bench.synth(1_000_000, 8, 16, 0)
x, r0 = torch.as_tensor(x_h).cuda(), torch.as_tensor(r_h).cuda()
loop = _mix.VMPLoop(x, r0, V._lib.VMP_GMM)
for _ in range(20): loop.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(400):
    loop.finalize_phase(); loop.estep()
torch.cuda.synchronize()
print('eager : %.2f us/step' % ((time.perf_counter() - t0) / 400 * 1e6))
S = 20
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    loop.finalize_phase(); loop.estep()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(S):
        loop.finalize_phase(); loop.estep()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
print('graph : %.2f us/step' % ((time.perf_counter() - t0) / (20 * S) * 1e6))
