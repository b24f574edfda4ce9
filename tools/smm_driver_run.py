import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmp_for_svae_amd import experiments
cfg = dict(dataset='pinwheel', method='svae-smm', K=8, L=2, U=20, lr=3e-3, lrcvi=0.2, decay_rate=0.95, seed=0, DoF=5)
h = experiments.run(cfg, nb_iters=600, size_minibatch=100, measurement_freq=200, verbose=True)
print('ok', len(h))
