"""debug: two ranks on one GPU (gloo), data-parallel eager step vs the two-graph replay; prints local / reduced ELBOs"""
import os, sys, socket, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if 'RANK' not in os.environ:
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))) for r in range(2)]
    sys.exit(max(p.wait() for p in ps))
import torch, torch.distributed as dist
rank, world = int(os.environ['RANK']), 2
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=rank, world_size=world)
from vmp_for_svae_amd import data as data_mod
from vmp_for_svae_amd.models import vae
from vmp_for_svae_amd.training import SVAETrainer, GraphedSVAEStep
Kg, Lg, Ug, Dg, Sg, Ng = 10, 8, 50, 6, 10, 64
gsl = data_mod.tower_slice(Ng, rank, world)
gg = torch.Generator(device='cuda').manual_seed(17)
ys = [(torch.randn(Ng, Dg, device='cuda', generator=gg) * 2)[gsl].contiguous() for _ in range(3)]
def fresh():
    vae.reset_variables()
    return SVAETrainer(Kg, Lg, Ug, Dg, nb_samples=Sg, lr=3e-3, lrcvi=0.2, decay_rate=0.95, stddev_init_nn=0.1, seed=3, rng=os.environ.get('RNG', 'philox'))
if os.environ.get('PRE_RUN'):
    from vmp_for_svae_amd import experiments
    vae.reset_variables()
    cfg = {'dataset': 'pinwheel', 'method': 'svae-cvi', 'lr': 0.003, 'lrcvi': 0.2, 'K': 5, 'L': 2, 'U': 20, 'seed': 0}
    tr2, hist, _ = experiments.run(cfg, nb_iters=6, size_minibatch=64, nb_samples=4, nb_samples_te=4, measurement_freq=100, verbose=False, graph=os.environ.get('PRE_GRAPH', '1') == '1')
    print('rank %d pre-run done' % rank, flush=True)
if os.environ.get('PRE_STEPPER'):
    tr_p = fresh()
    gsp = GraphedSVAEStep(tr_p, ys[0], warmup=2)
    for i in range(4):
        gsp(ys[i % 3])
    torch.cuda.synchronize()
    if os.environ.get('PRE_STEPPER') == 'del':
        del gsp, tr_p
        import gc; gc.collect(); torch.cuda.synchronize()
    print('rank %d pre-stepper done' % rank, flush=True)
tr_e = fresh()
f32 = dict(dtype=torch.float32, device='cuda')
vae.make_encoder(torch.zeros(1, Dg, **f32), tr_e.encoder_layers, tr_e.stddev_init_nn, seed=tr_e.seed)
vae.decoder_variables(tr_e.L, tr_e.decoder_layers, tr_e.stddev_init_nn, tr_e.seed, torch.device('cuda'))
print('rank %d init checksums: vars %.6f theta %.6f phi %.6f prior %.6f y %.6f' % (rank, sum(float(v.double().abs().sum()) for v in vae.VARIABLES.values()),
      sum(float(t.double().abs().sum()) for t in tr_e.theta), sum(float(t.double().abs().sum()) for t in tr_e.phi_gmm),
      sum(float(t.double().abs().sum()) for t in tr_e.gmm_prior), float(ys[0].double().abs().sum())), flush=True)
for i in range(3):
    ctx = tr_e._step_front(ys[i])
    loc = float(ctx['scal'][0])
    tr_e._step_exchange(ctx)
    out = tr_e._step_back(ctx)
    print('rank %d eager   step %d local elbo %.4f reduced %.4f  phi0 %.6f' % (rank, i, loc, float(out['elbo']), float(tr_e.phi_gmm[0].sum())), flush=True)
if os.environ.get('EAGER_ONLY'):
    dist.barrier(); dist.destroy_process_group(); sys.exit(0)
tr_g = fresh()
gs = GraphedSVAEStep(tr_g, ys[0], warmup=2)
print('rank %d after capture: phi0 %.6f step %d' % (rank, float(tr_g.phi_gmm[0].sum()), tr_g.global_step), flush=True)
for i in range(3):
    if os.environ.get('USE_CALL'):
        o = gs(ys[i]); loc = float('nan')
    else:
        gs.y.copy_(ys[i]); gs._refresh(); gs.graph.replay(); torch.cuda.synchronize()
        loc = float(gs._ctx['scal'][0])
        tr_g._step_exchange(gs._ctx); gs.graph_back.replay(); tr_g.opt.t += 1; tr_g.global_step += 1
    print('rank %d graphed step %d local elbo %.4f reduced %.4f  phi0 %.6f' % (rank, i, loc, float(gs.out['elbo']), float(tr_g.phi_gmm[0].sum())), flush=True)
dist.barrier(); dist.destroy_process_group()
