import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmp_for_svae_amd.models import vae
from vmp_for_svae_amd.training import SVAETrainer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
vae.reset_variables()
y = torch.randn(N, 6, device='cuda')
tr = SVAETrainer(10, 8, 50, 6, nb_samples=10)
for _ in range(20):
    tr.step(y)
torch.cuda.synchronize()
