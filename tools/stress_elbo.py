"""Run-to-run determinism stress of the forward half of the T3 step at N=65536, K=16 (every kernel is deterministic by
construction: any difference between repetitions is a race).  Two different inputs alternate, so that a value left over
from the previous launch differs from the fresh one.  Prints which tensors differed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmp_for_svae_amd.models import svae, vae, _svae_ops
from vmp_for_svae_amd.training import SVAETrainer
N, K, Ld, S, Dy, U = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, int(os.environ.get('K', 16)), 8, 10, 8, 50
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
vae.reset_variables()
g = torch.Generator(device='cuda').manual_seed(0)
ys = [torch.randn(N, Dy, device='cuda', generator=g) * (1.0 + 0.5 * i) for i in range(2)]   # TWO inputs, alternating: a stale
                                                                                             # result of the other one is visible
tr = SVAETrainer(K, Ld, U, Dy, nb_samples=S, stddev_init_nn=0.3)
noises = [torch.randn(N, K, Ld, S, device='cuda', generator=g) for _ in range(2)]
zd = torch.randint(0, K, (N, S), device='cuda', generator=g)
refs = [None, None]
bad = {}
for it in range(reps):
    y, noise = ys[it & 1], noises[it & 1]
    ref = refs[it & 1]
    out = svae.inference(y, tr.phi_gmm, tr.encoder_layers, tr.decoder_layers, S, stddev_init_nn=tr.stddev_init_nn, seed=0,
                         noise=noise, z_draws=zd, theta=tr.theta, lazy_decoder=True)
    y_rec, phi_enc, x_k, x_s, log_z, _, phi_tilde = out
    elbo, details = svae.compute_elbo(y, y_rec, tr.theta, phi_tilde, x_k, log_z, 'standard')
    cur = dict(enc1=phi_enc[0], enc2=phi_enc[1], x=x_k, lz=log_z, Tp=phi_tilde.T_prime, xs=x_s, elbo=elbo, rec=details[0], reg=details[3],
               r=details.r_nk)
    names, params = tr.trainables()
    gr = torch.autograd.grad(elbo, params, allow_unused=True)
    for n_, g_ in zip(names, gr):
        if g_ is not None:
            cur['d/' + n_] = g_
    cur = {k: v.detach().clone() for k, v in cur.items()}
    if ref is None:
        refs[it & 1] = cur
        continue
    for k in cur:
        if not torch.equal(cur[k], ref[k]):
            bad.setdefault(k, []).append((it, (cur[k].double() - ref[k].double()).abs().max().item()))
print('repetitions', reps, 'elbo', refs[0]['elbo'].item(), refs[1]['elbo'].item())
print('differences:', {k: (len(v), v[:3]) for k, v in bad.items()} if bad else 'none')
