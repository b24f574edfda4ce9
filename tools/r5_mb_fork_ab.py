"""Graph-replayed minibatch-64 step with and without the side-stream branches (GraphedSVAEStep(fork=...)): us per step, same process."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmp_for_svae_amd.models import vae
from vmp_for_svae_amd.training import SVAETrainer, GraphedSVAEStep
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K, Ld, S, U, Dy = 10, 8, 10, 50, 6
dev = torch.device('cuda', 0)
for fork in (False, True, False, True):
    vae.reset_variables()
    torch.manual_seed(0)
    y = torch.randn(N, Dy, device=dev) * 2
    tr = SVAETrainer(K, Ld, U, Dy, nb_samples=S, device=dev)
    gs = GraphedSVAEStep(tr, y, fork=fork)
    for _ in range(10):
        gs(y)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200):
            out = gs(y)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200)
    print('fork=%s: graphed minibatch N=%d: %.1f us/step  elbo %.6f' % (fork, N, best * 1e6, float(out['elbo'])))
