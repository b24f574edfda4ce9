"""How should the fp64 chunked oracles of the full-size tests use the GPU box's host?  (threads, concurrent chunks) sweep, CPU only."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import train_ref
K, Ld, S, Ns = 16, 8, 10, 131072
rng, _, prior, theta, phi, noise, zd = bench._cpu_model_inputs(K, Ld, Ld, 8, Ns, S)
D = lambda t: t.double()
prior = [D(t) for t in prior]; theta = [D(t) for t in theta]; phi = [D(t) for t in phi]; noise = D(noise)
e1 = torch.as_tensor(rng.standard_normal((Ns, Ld))); e2 = -0.5 * torch.nn.functional.softplus(torch.as_tensor(rng.standard_normal((Ns, Ld))))
Gx = torch.as_tensor(rng.standard_normal((Ns, K, S, Ld))) * 0.01; Glz = torch.as_tensor(rng.standard_normal((Ns, K))) * 0.1
print('host threads', os.cpu_count())
for th, w, ch in ((32, 1, 8192), (32, 4, 8192), (32, 8, 8192), (64, 8, 8192), (64, 16, 8192), (128, 16, 8192), (16, 8, 8192), (32, 8, 4096), (32, 16, 4096)):
    torch.set_num_threads(th)
    t0 = time.time(); r = train_ref.vmp_step_t2(phi, theta, prior, e1, e2, noise, zd, Gx, Glz, 0.2, chunk=ch, workers=w); dt = time.time() - t0
    print('threads %3d workers %2d chunk %5d: %.2fs (%.0f rows/s)  reg %.10f' % (th, w, ch, dt, Ns / dt, r['reg'].item()), flush=True)
