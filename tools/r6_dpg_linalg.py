"""Round-6 hunt for the round-5 finding "torch's GPU Cholesky returns a wrong factor after a data-parallel graphed step had run in a
process that shares its GPU with another rank".  Two ranks (gloo) on GPU 0; MODE selects what happens around the two-graph stepper:
  base          linalg checked before AND after the graphed steps
  nopre         no linalg call before the first capture (is it a lazily created solver handle / workspace?)
  sep_pools     the second graph gets its own memory pool
  sync_between  torch.cuda.synchronize() between the two replays and the collective
  eager         no graphs at all (control)
  onegraph      world-size-1 style capture in each process (no collective, one graph; control for "two processes on one GPU")
Prints, per rank, the largest deviation of GPU cholesky / inv / solve_triangular of a fixed batch from the host result."""
import os, sys, socket, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODE = os.environ.get('MODE', 'base')
if 'RANK' not in os.environ:
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))) for r in range(2)]
    sys.exit(max(p.wait() for p in ps))
import torch, torch.distributed as dist
rank, world = int(os.environ['RANK']), 2
torch.cuda.set_device(0)
if MODE != 'onegraph':
    dist.init_process_group('gloo', rank=rank, world_size=world)
from vmp_for_svae_amd import data as data_mod, training
from vmp_for_svae_amd.models import vae, svae
from vmp_for_svae_amd.training import SVAETrainer, GraphedSVAEStep
K, Ld, U, Dy, S, N = 10, 8, 50, 6, 10, 64
gcpu = torch.Generator().manual_seed(5)
A = torch.randn(K, Ld, Ld, generator=gcpu)
SPD = A @ A.transpose(-1, -2) + Ld * torch.eye(Ld)
want = (torch.linalg.cholesky(SPD), torch.linalg.inv(SPD))
def check(tag):
    d = SPD.cuda()
    c = torch.linalg.cholesky(d).cpu(); i = torch.linalg.inv(d).cpu()
    prior, theta = svae.init_mm(K, Ld, seed=3, param_device='cuda')
    std = [t.cpu() for t in theta]
    e = max((c - want[0]).abs().max().item(), (i - want[1]).abs().max().item())
    # the K-sized initialisation the round-5 symptom was seen in, evaluated on the device and on the host
    from vmp_for_svae_amd.distributions import niw
    mu_d, sig_d = niw.expected_values(niw.natural_to_standard(*theta[1:]))
    mu_h, sig_h = niw.expected_values(niw.natural_to_standard(*std[1:]))
    e2 = (torch.linalg.cholesky(sig_d).cpu() - torch.linalg.cholesky(sig_h)).abs().max().item()
    print('rank %d MODE=%s %-22s cholesky/inv dev %.3e   init chol dev %.3e' % (rank, MODE, tag, e, e2), flush=True)
    return max(e, e2)
bad = 0.0
if MODE != 'nopre':
    bad = max(bad, check('before any graph'))
gsl = data_mod.tower_slice(N, rank, world)
gg = torch.Generator(device='cuda').manual_seed(17)
ys = [(torch.randn(N, Dy, device='cuda', generator=gg) * 2)[gsl].contiguous() for _ in range(3)]
vae.reset_variables()
tr = SVAETrainer(K, Ld, U, Dy, nb_samples=S, lr=3e-3, lrcvi=0.2, decay_rate=0.95, stddev_init_nn=0.1, seed=3)
if MODE == 'eager':
    for i in range(4):
        tr.step(ys[i % 3])
else:
    if MODE == 'sep_pools':
        orig = torch.cuda.graph
        class G(orig):
            def __init__(self, g, pool=None, stream=None, **kw):
                super().__init__(g, pool=None, stream=stream, **kw)
        torch.cuda.graph = G
    gs = GraphedSVAEStep(tr, ys[0], warmup=2)
    if MODE == 'sep_pools':
        torch.cuda.graph = orig
    bad = max(bad, check('after capture'))
    for i in range(4):
        if MODE == 'sync_between':
            gs.y.copy_(ys[i % 3]); gs._refresh(); gs.graph.replay(); torch.cuda.synchronize()
            tr._step_exchange(gs._ctx); torch.cuda.synchronize(); gs.graph_back.replay(); torch.cuda.synchronize()
            tr.opt.t += 1; tr.global_step += 1
        else:
            gs(ys[i % 3])
torch.cuda.synchronize()
bad = max(bad, check('after 4 steps'))
vae.reset_variables()
tr2 = SVAETrainer(K, Ld, U, Dy, nb_samples=S, seed=3)          # the round-5 symptom: a trainer built AFTER the stepper
bad = max(bad, check('after a second trainer'))
print('rank %d MODE=%s RESULT %s (%.3e)' % (rank, MODE, 'CLEAN' if bad < 1e-4 else 'CORRUPT', bad), flush=True)
if MODE != 'onegraph':
    dist.barrier(); dist.destroy_process_group()
