"""Round-6: the round-5 symptom with the ORIGINAL device-side make_loc_scale_variables (monkeypatched in), same sequence as
tools/r5_dpg_debug.py: [pre-run of experiments.run with graphs | a pre-stepper] -> fresh trainer -> checksum of its phi_gmm
against the same construction with host linalg.  MODE: none | prerun | prestepper | prestepper_del | prerun_nograph"""
import os, sys, socket, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODE = os.environ.get('MODE', 'prerun')
if 'RANK' not in os.environ:
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))) for r in range(2)]
    sys.exit(max(p.wait() for p in ps))
import torch, torch.distributed as dist
rank, world = int(os.environ['RANK']), 2
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=rank, world_size=world)
from vmp_for_svae_amd import data as data_mod
from vmp_for_svae_amd.distributions import niw
from vmp_for_svae_amd.models import vae, svae
from vmp_for_svae_amd.training import SVAETrainer, GraphedSVAEStep
host_version = svae.make_loc_scale_variables
def device_version(theta, param_device='cuda', name='copy_m_v'):           # round 5's form before the host-side workaround
    std = niw.natural_to_standard(theta[1], theta[2], theta[3], theta[4])
    mu, sigma = niw.expected_values(std)
    return (torch.nn.Parameter(mu.clone(memory_format=torch.contiguous_format)),
            torch.nn.Parameter(torch.linalg.cholesky(sigma).clone(memory_format=torch.contiguous_format)))
Kg, Lg, Ug, Dg, Sg, Ng = 10, 8, 50, 6, 10, 64
gsl = data_mod.tower_slice(Ng, rank, world)
gg = torch.Generator(device='cuda').manual_seed(17)
ys = [(torch.randn(Ng, Dg, device='cuda', generator=gg) * 2)[gsl].contiguous() for _ in range(3)]
def fresh():
    vae.reset_variables()
    return SVAETrainer(Kg, Lg, Ug, Dg, nb_samples=Sg, lr=3e-3, lrcvi=0.2, decay_rate=0.95, stddev_init_nn=0.1, seed=3)
KEEP = []
if MODE.startswith('prerun'):
    from vmp_for_svae_amd import experiments, training
    vae.reset_variables()
    big = 'big' in MODE
    cfg = {'dataset': 'pinwheel', 'method': 'svae-cvi', 'lr': 0.003, 'lrcvi': 0.2, 'K': 10 if big else 5, 'L': 8 if big else 2, 'U': 50 if big else 20, 'seed': 0}
    if 'keep' in MODE:                                   # the stepper outlives experiments.run
        orig_init = training.GraphedSVAEStep.__init__
        def init(self, *a, **k):
            orig_init(self, *a, **k); KEEP.append(self)
        training.GraphedSVAEStep.__init__ = init
    if 'noeval' in MODE:
        experiments.evaluate = lambda *a, **k: {}
    experiments.run(cfg, nb_iters=6, size_minibatch=64, nb_samples=4, nb_samples_te=4, measurement_freq=100, verbose=False, graph=('nograph' not in MODE))
    if 'sync' in MODE:
        import gc; gc.collect(); torch.cuda.synchronize()
    if 'cur' in MODE:                                    # only the current stream
        torch.cuda.current_stream().synchronize()
if MODE.startswith('prestepper'):
    tr_p = fresh()
    gsp = GraphedSVAEStep(tr_p, ys[0], warmup=2)
    for i in range(4):
        gsp(ys[i % 3])
    torch.cuda.synchronize()
    if MODE == 'prestepper_del':
        del gsp, tr_p
        import gc; gc.collect(); torch.cuda.synchronize()
res = []
for rep in range(3):
    svae.make_loc_scale_variables = device_version
    tr_d = fresh()
    pd = [t.detach().cpu().clone() for t in tr_d.phi_gmm]
    svae.make_loc_scale_variables = host_version
    tr_h = fresh()
    ph = [t.detach().cpu().clone() for t in tr_h.phi_gmm]
    res.append(max((a - b).abs().max().item() for a, b in zip(pd, ph)))
    if res[-1] > 1e-5:
        print('rank %d   construction %d: per tensor (mu_k, L_k, pi_k) %s; theta / prior equal to host-built ones: %s' % (
            rank, rep, ['%.2e' % (a - b).abs().max().item() for a, b in zip(pd, ph)],
            [bool(torch.equal(a.cpu(), b.cpu())) for a, b in zip(list(tr_d.theta) + list(tr_d.gmm_prior), list(tr_h.theta) + list(tr_h.gmm_prior))]), flush=True)
print('rank %d MODE=%s  max |phi_gmm(device linalg) - phi_gmm(host linalg)| over 3 constructions: %s  -> %s'
      % (rank, MODE, ['%.2e' % r for r in res], 'CLEAN' if max(res) < 1e-5 else 'CORRUPT'), flush=True)
dist.barrier(); dist.destroy_process_group()
