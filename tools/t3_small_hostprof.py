"""Host-side profile of the eager minibatch-64 training step (cProfile, 300 steps)."""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmp_for_svae_amd.models import vae
from vmp_for_svae_amd.training import SVAETrainer
vae.reset_variables()
y = torch.randn(64, 6, device='cuda')
tr = SVAETrainer(10, 8, 50, 6, nb_samples=10)
for _ in range(20):
    tr.step(y)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    tr.step(y)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
