"""End-to-end sanity: the reference's pinwheel configuration (experiments.py:69-79: K=10, L=2, U=50, minibatch 100,
lr 0.01, lrcvi 0.1) for 3000 iterations through experiments.run (graph-replayed steps), with wall-clock."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmp_for_svae_amd import experiments
cfg = {'dataset': 'pinwheel', 'method': 'svae-cvi', 'lr': 0.01, 'lrcvi': 0.1, 'K': 10, 'L': 2, 'U': 50, 'seed': 0}
for graph in (True, False):
    torch.cuda.synchronize()
    t0 = time.time()
    tr, hist, log_id = experiments.run(cfg, nb_iters=3000, measurement_freq=1000, verbose=False, graph=graph)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print('graph=%s: 3000 iterations + 4 evaluations (S=100) in %.2f s  (%.0f it/s)' % (graph, dt, 3000 / dt))
    for h in hist:
        print('   ', {k: round(v, 4) for k, v in h.items()})
