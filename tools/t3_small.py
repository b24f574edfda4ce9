"""Small-minibatch training step (the reference's real operating point: size_minibatch 64-100): ms/step, kernel count."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vmp_for_svae_amd.models import vae
from vmp_for_svae_amd.training import SVAETrainer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K, Ld, S, U, Dy = 10, 8, 10, 50, 6
dev = torch.device('cuda', 0)
vae.reset_variables()
y = torch.randn(N, Dy, device=dev)
tr = SVAETrainer(K, Ld, U, Dy, nb_samples=S, device=dev)
for _ in range(5):
    tr.step(y)
torch.cuda.synchronize()
steps = 50
t0 = time.perf_counter()
for _ in range(steps):
    out = tr.step(y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print('T3 small N=%d K=%d L=%d U=%d: %.3f ms/step (%.0f steps/s)' % (N, K, Ld, U, dt * 1e3, 1 / dt))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        tr.step(y)
    torch.cuda.synchronize()
ev = prof.key_averages()
nk = sum(e.count for e in ev if e.device_type is not None and str(e.device_type).endswith('CUDA'))
tk = sum(e.device_time_total for e in ev if str(e.device_type).endswith('CUDA'))
print('GPU kernels per step: %.0f, GPU busy per step: %.3f ms' % (nk / 3.0, tk / 3.0 / 1e3))
print(prof.key_averages().table(sort_by='cpu_time_total', row_limit=25, max_name_column_width=60))

# ---- the same step captured as a HIP graph
from vmp_for_svae_amd.training import GraphedSVAEStep
gs = GraphedSVAEStep(tr, y)
for _ in range(5):
    gs(y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    out = gs(y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
print('T3 small GRAPHED N=%d: %.3f ms/step (%.0f steps/s)  elbo %.3f' % (N, dt * 1e3, 1 / dt, float(out['elbo'])))
# ---- what the per-call refreshes outside the graph cost: replay alone
t0 = time.perf_counter()
for _ in range(200):
    gs.graph.replay()
torch.cuda.synchronize()
dt2 = (time.perf_counter() - t0) / 200
print('T3 small GRAPHED replay only: %.3f ms/step' % (dt2 * 1e3))
