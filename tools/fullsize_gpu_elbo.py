"""GPU half of tests/test_fullsize_gpu.py::test_t3_training_step_at_65536_vs_chunked_oracle[c3-65536]: prints the ELBO of the
step, several fresh repetitions (which of two disagreeing numbers is the GPU's usual one?)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import test_fullsize_gpu as T
from vmp_for_svae_amd.models import vae
from vmp_for_svae_amd.training import SVAETrainer
N, K, Ld, S, Dy, U = 65536, 16, 8, 10, 8, 50
for rep in range(4):
    y, w, m_unif, pi_norm, Lk_low = T._svae_problem(N, K, Ld, S, Dy, U, seed=3)
    g = torch.Generator(device='cuda').manual_seed(11)
    noise = torch.randn(N, K, Ld, S, device='cuda', generator=g)
    zd = torch.randint(0, K, (N, S), device='cuda', generator=g)
    vae.reset_variables()
    for n_, v in w.items():
        vae.VARIABLES[n_] = torch.nn.Parameter(torch.as_tensor(v).cuda())
    tr = SVAETrainer(K, Ld, U, Dy, nb_samples=S, m_uniform=torch.as_tensor(m_unif).cuda(), pi_normal=torch.as_tensor(pi_norm).cuda())
    with torch.no_grad():
        tr.phi_gmm[1].add_(torch.as_tensor(Lk_low).cuda())
    out = tr.step(torch.as_tensor(y).cuda(), noise=noise, z_draws=zd)
    print('elbo %.3f  rec %.3f  reg %.3f' % (out['elbo'].item(), out['neg_rec_err'].item(), out['regulariser'].item()))
