"""Graph-replayed minibatch-64 training step (Auto-sized model: K=10, L=8, Dy=6, U=50, S=10): timing and, under
tools/kseq.sh <tag> step_scalars tools/r5_mb_graph.py, the kernel sequence of one replay."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmp_for_svae_amd.models import vae
from vmp_for_svae_amd.training import SVAETrainer, GraphedSVAEStep
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K, Ld, S, U, Dy = 10, 8, 10, 50, 6
dev = torch.device('cuda', 0)
vae.reset_variables()
y = torch.randn(N, Dy, device=dev) * 2
tr = SVAETrainer(K, Ld, U, Dy, nb_samples=S, device=dev)
gs = GraphedSVAEStep(tr, y)
for _ in range(10):
    gs(y)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(200):
        out = gs(y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print('graphed minibatch N=%d: %.1f us/step  elbo %.4f' % (N, dt * 1e6, float(out['elbo'])))
