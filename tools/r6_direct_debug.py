"""direct step vs autograd step: which tensors differ (round 6 debugging aid); kernel level: vmp_svae_elbo_tail + vmp_svae_estep_bwd_n
against vmp_svae_estep_bwd_tail on the same inputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vmp_for_svae_amd as V
from vmp_for_svae_amd.models import vae
from vmp_for_svae_amd.training import SVAETrainer
L = V._lib
lib = L.lib()
N, K, Ld, U, Dy, S = 64, 10, 8, 50, 6, 10
g = torch.Generator(device='cuda').manual_seed(N + K)
y = torch.randn(N, Dy, device='cuda', generator=g) * 2
outs = {}
for direct in (False, True):
    vae.reset_variables()
    tr = SVAETrainer(K, Ld, U, Dy, nb_samples=S, lr=3e-3, lrcvi=0.2, decay_rate=0.95, stddev_init_nn=0.1, seed=3, direct_step=direct)
    outs[direct] = tr.step(y)
a, b = outs[True], outs[False]
for k in b['grads']:
    d = (a['grads'][k].double() - b['grads'][k].double()).abs().max().item()
    print('%-32s max|diff| %.3e  rel %.3e  equal %s' % (k, d, d / b['grads'][k].abs().max().item(), torch.equal(a['grads'][k], b['grads'][k])))
print('log_z', torch.equal(a['log_z'], b['log_z']), 'x_k', torch.equal(a['x_k'], b['x_k']), 'xs', torch.equal(a['x_samples'], b['x_samples']))
f32 = dict(dtype=torch.float32, device='cuda')
x = a['x_k'].contiguous(); lz = a['log_z'].contiguous()
Tp = torch.randn(N, K, **f32); ll = torch.randn(N, K, S, **f32) * 3 + 8; Gx = torch.randn(N, K, S, Ld, **f32)
eta1 = torch.randn(N, Ld, **f32); eta2d = -torch.rand(N, Ld, **f32) - 0.5
hk = torch.randn(K, Ld, **f32); A_ = torch.randn(K, Ld, Ld, **f32); Pk = (A_ @ A_.transpose(1, 2) + torch.eye(Ld, **f32)).contiguous()
bias = torch.randn(K, **f32); mk = torch.randn(K, Ld, **f32); Wk = torch.tril(torch.randn(K, Ld, Ld, **f32)).contiguous()
scal = torch.empty(3, **f32); g_lz, g_Tp, r = torch.empty(N, K, **f32), torch.empty(N, K, **f32), torch.empty(N, K, **f32)
ws = torch.empty(lib.vmp_svae_elbo_tail_workspace_bytes(), dtype=torch.uint8, device='cuda')
L.check(lib.vmp_svae_elbo_tail(L.ptr(lz), L.ptr(Tp), L.ptr(ll), N, K, S, Dy, -1.0, L.ptr(scal), L.ptr(g_lz), L.ptr(g_Tp), L.ptr(r), L.ptr(ws),
                               ws.numel(), L.stream()), 'tail')
nt = lib.vmp_svae_bwd_blocks_for(N, K, Ld, S, 0); PW = lib.vmp_svae_bwd_partial_words(Ld)
res = []
for mode in (0, 1):
    ge1, ge2 = torch.empty(N, Ld, **f32), torch.empty(N, Ld, **f32)
    part = torch.zeros(nt, K, PW, **f32)
    if mode == 0:
        L.check(lib.vmp_svae_estep_bwd_n(L.ptr(eta1), L.ptr(eta2d), L.ptr(hk), L.ptr(Pk), L.ptr(bias), L.ptr(mk), L.ptr(Wk), None, L.ptr(x),
                                         L.ptr(lz), L.ptr(Gx), L.ptr(g_lz), L.ptr(g_Tp), N, K, Ld, S, L.ptr(ge1), L.ptr(ge2), L.ptr(part),
                                         part.numel() * 4, nt, L.stream()), 'bwd_n')
        res.append((ge1, ge2, part, r, scal.clone()))
    else:
        r2 = torch.empty(N, K, **f32); tp = torch.empty(nt, 2, dtype=torch.float64, device='cuda')
        L.check(lib.vmp_svae_estep_bwd_tail(L.ptr(eta1), L.ptr(eta2d), L.ptr(hk), L.ptr(Pk), L.ptr(bias), L.ptr(mk), L.ptr(Wk), L.ptr(x),
                                            L.ptr(lz), L.ptr(Tp), L.ptr(ll), -1.0, L.ptr(Gx), N, K, Ld, S, L.ptr(ge1), L.ptr(ge2),
                                            L.ptr(part), part.numel() * 4, L.ptr(r2), L.ptr(tp), tp.numel() * 8, L.stream()), 'bwd_tail')
        res.append((ge1, ge2, part, r2, tp))
for i, nm in enumerate(('g_eta1', 'g_eta2d', 'partials', 'r')):
    d = (res[0][i].double() - res[1][i].double()).abs().max().item()
    print('kernel level %-10s max|diff| %.3e equal %s' % (nm, d, torch.equal(res[0][i], res[1][i])))
tp = res[1][4].sum(0)
rec = -tp[0].item() - N * Dy * 0.5 * 1.8378770664093453
print('scalars', res[0][4].tolist(), [rec - tp[1].item(), rec, tp[1].item()])

# ---- as the test: 4 steps each, the autograd path first; which step / tensor / element differs, and is each path repeatable
ys = []
g = torch.Generator(device='cuda').manual_seed(N + K)
ys = [torch.randn(N, Dy, device='cuda', generator=g) * 2 for _ in range(4)]
def run(direct):
    vae.reset_variables()
    tr = SVAETrainer(K, Ld, U, Dy, nb_samples=S, lr=3e-3, lrcvi=0.2, decay_rate=0.95, stddev_init_nn=0.1, seed=3, direct_step=direct)
    res = []
    for i in range(4):
        o = tr.step(ys[i])
        res.append({k: v.detach().clone() for k, v in o['grads'].items()})
    return res
runs = [('autograd', run(False)), ('direct', run(True)), ('autograd2', run(False)), ('direct2', run(True))]
for (na, ra), (nb_, rb) in ((runs[0], runs[1]), (runs[0], runs[2]), (runs[1], runs[3])):
    for i in range(4):
        bad = [k for k in ra[i] if not torch.equal(ra[i][k], rb[i][k])]
        print(na, 'vs', nb_, 'step', i, 'differing:', bad)
        for k in bad[:2]:
            d = (ra[i][k] != rb[i][k]).nonzero()
            print('   ', k, 'elements', d[:6].tolist(), 'of', tuple(ra[i][k].shape))

# ---- lockstep: state after every step
def mk(direct):
    vae.reset_variables()
    return SVAETrainer(K, Ld, U, Dy, nb_samples=S, lr=3e-3, lrcvi=0.2, decay_rate=0.95, stddev_init_nn=0.1, seed=3, direct_step=direct)
def state(tr):
    names, ps = tr.trainables()
    d = {n: p.detach().clone() for n, p in zip(names, ps)}
    d.update({'theta%d' % i: t.clone() for i, t in enumerate(tr.theta)})
    d.update({'m/' + n: t.clone() for n, t in zip(names, tr.opt.m)})
    d.update({'v/' + n: t.clone() for n, t in zip(names, tr.opt.v)})
    return d
states = {}
for direct in (False, True):
    tr = mk(direct)
    states[direct] = []
    for i in range(3):
        o = tr.step(ys[i])
        st_ = state(tr)
        st_['stats'] = o['stats'].clone() if 'stats' in o else None
        st_.update({'star%d' % j: t.clone() for j, t in enumerate(o['theta_star'])})
        states[direct].append(st_)
for i in range(3):
    bad = [k for k in states[False][i] if states[False][i][k] is not None and not torch.equal(states[False][i][k], states[True][i][k])]
    print('after step', i, 'differing state:', bad)
    for k in bad[:4]:
        x_, y_ = states[False][i][k], states[True][i][k]
        idx = (x_ != y_).nonzero()[:3].tolist()
        print('   ', k, idx, [(x_[tuple(j)].item(), y_[tuple(j)].item()) for j in idx])
