"""T3 (full SVAE training step) profiling target: N rows, no chunking; prints ms/step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vmp_for_svae_amd.models import vae
from vmp_for_svae_amd.training import SVAETrainer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
fused = (sys.argv[2] != '0') if len(sys.argv) > 2 else True
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else None
K, Ld, S, U = 16, 8, 10, 50
dev = torch.device('cuda', 0)
vae.reset_variables()
x_h, _ = bench.synth(N, Ld, K, seed=7)
y = torch.as_tensor(x_h).to(dev)
tr = SVAETrainer(K, Ld, U, Ld, nb_samples=S, device=dev, fused_decoder=fused)
for _ in range(2):
    tr.step(y, chunk=chunk)
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 5
for _ in range(steps):
    out = tr.step(y, chunk=chunk)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print('T3 N=%d fused=%s chunk=%s: %.3f ms/step  (%.3g datapoints/s)  elbo/N %.4f' % (N, fused, chunk, dt * 1e3, N / dt, float(out['elbo']) / N))
