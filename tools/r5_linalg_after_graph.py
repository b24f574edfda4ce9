"""Does torch's GPU linalg (the K-sized Cholesky of SVAETrainer.__init__ -> svae.make_loc_scale_variables) return the same numbers
after a HIP graph has been captured in the process?  single process"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmp_for_svae_amd.models import vae, svae
from vmp_for_svae_amd.training import SVAETrainer, GraphedSVAEStep
K, Ld, U, Dy, S, N = 10, 8, 50, 6, 10, 64
def chk():
    prior, theta = svae.init_mm(K, Ld, seed=3, param_device='cuda')
    phi = svae.init_recognition_params(theta, K, seed=3, param_device='cuda')
    sig = torch.eye(Ld, device='cuda').expand(K, Ld, Ld) * 2.0 + 0.1
    return [float(t.double().abs().sum()) for t in phi] + [float(torch.linalg.cholesky(sig).double().abs().sum()), float(torch.linalg.inv(sig).double().abs().sum())]
print('before any graph      :', chk())
vae.reset_variables()
tr = SVAETrainer(K, Ld, U, Dy, nb_samples=S, seed=3)
y = torch.randn(N, Dy, device='cuda')
gs = GraphedSVAEStep(tr, y, warmup=2)
print('after capture         :', chk())
for _ in range(3): gs(y)
torch.cuda.synchronize()
print('after replays         :', chk())
del gs, tr
import gc; gc.collect(); torch.cuda.synchronize()
print('after graph destroyed :', chk())
