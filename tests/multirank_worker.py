"""One rank of the multi-rank GPU tests (tests/test_multirank_gpu.py starts WORLD_SIZE of these as child processes).
Every rank runs the product's data-parallel step functions on ITS shard of the rows - HIP kernels on the (shared) GPU,
the one packed all-reduce of a step through torch.distributed (gloo rendezvous: RCCL refuses two ranks on one device) -
and writes what it ends up with to <out>/rank<r>.npz.  Not a test module itself (no test_ prefix)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    payload, out_dir = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import vmp_for_svae_amd as V
    from vmp_for_svae_amd import data as data_mod, experiments
    from vmp_for_svae_amd.models import vae
    from vmp_for_svae_amd.models.parallel_mix import DistributedVMPLoop
    from vmp_for_svae_amd.training import SVAETrainer
    L = V._lib
    p = np.load(payload)
    dev = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a)).to('cuda', dt)
    res = {}

    # ---- T1: DistributedVMPLoop.step on a contiguous row shard (uneven split on purpose)
    x, r0 = p['t1_x'], p['t1_r0']
    N = x.shape[0]
    cut = [0, int(N * 0.37), N] if world == 2 else [N * i // world for i in range(world + 1)]
    sl = slice(cut[rank], cut[rank + 1])
    for name, flav in (('gmm', L.VMP_GMM), ('smm', L.VMP_SMM)):
        kap = torch.full((r0.shape[1],), 5.0, device='cuda') if flav == L.VMP_SMM else None
        loop = DistributedVMPLoop(dev(x[sl]), dev(r0[sl]), flav, kappa=kap)
        for it in range(3):
            r = loop.step()
            if it == 0:
                res['t1_%s_r1' % name] = r.cpu().numpy()
        res['t1_%s_r' % name] = r.cpu().numpy()
        for n_, t in zip(('alpha', 'beta', 'm', 'C', 'v'), loop.theta()):
            res['t1_%s_%s' % (name, n_)] = t.cpu().numpy()

    # ---- T3: SVAETrainer.step on this tower's rows of the minibatch (tf.split semantics, data.py:174-175)
    Nb, K, Ld, S, Dy, U = [int(v) for v in p['t3_dims']]
    tsl = data_mod.tower_slice(Nb, rank, world)
    vae.reset_variables()
    for k in p.files:
        if k.startswith('w_'):
            vae.VARIABLES[k[2:]] = torch.nn.Parameter(dev(p[k]))
    tr = SVAETrainer(K, Ld, U, Dy, nb_samples=S, m_uniform=dev(p['t3_m_unif']), pi_normal=dev(p['t3_pi_norm']))
    with torch.no_grad():
        tr.phi_gmm[1].add_(dev(p['t3_Lk_low']))
    for it in range(2):
        out = tr.step(dev(p['t3_y'][tsl]), noise=dev(p['t3_noise'][it][tsl]), z_draws=dev(p['t3_zd'][it][tsl], torch.int64))
        res['t3_elbo%d' % it] = np.float64(out['elbo'].item())
        for n_, g in out['grads'].items():
            res['t3_grad%d_%s' % (it, n_)] = g.cpu().numpy()
        for n_, t in zip(('alpha', 'A', 'b', 'beta', 'vhat'), out['theta_star']):
            res['t3_theta_star%d_%s' % (it, n_)] = t.cpu().numpy()
    names, params = tr.trainables()
    for n_, t in zip(names, params):
        res['t3_param_' + n_] = t.detach().cpu().numpy()
    for n_, t in zip(('alpha', 'A', 'b', 'beta', 'vhat'), tr.theta):
        res['t3_theta_' + n_] = t.cpu().numpy()

    # ---- the driver: experiments.run shards every minibatch by rank and ends with identical parameters everywhere
    vae.reset_variables()
    cfg = {'dataset': 'pinwheel', 'method': 'svae-cvi', 'lr': 0.003, 'lrcvi': 0.2, 'K': 5, 'L': 2, 'U': 20, 'seed': 0}
    tr2, hist, _ = experiments.run(cfg, nb_iters=6, size_minibatch=64, nb_samples=4, nb_samples_te=4, measurement_freq=100,
                                   verbose=False)
    _, params2 = tr2.trainables()
    res['run_params'] = np.concatenate([t.detach().cpu().numpy().reshape(-1) for t in params2])
    res['run_theta'] = np.concatenate([t.cpu().numpy().reshape(-1) for t in tr2.theta])
    res['run_elbo'] = np.float64(hist[-1]['neg_normed_elbo'])
    # ---- the data-parallel training step captured as TWO HIP graphs around its one collective (training.GraphedSVAEStep with
    # several ranks, round 5) against the same step run eagerly: call i of the graphed stepper == training step i, on both ranks
    try:
        from vmp_for_svae_amd.training import GraphedSVAEStep
        Kg, Lg, Ug, Dg, Sg, Ng = 10, 8, 50, 6, 10, 64
        gsl = data_mod.tower_slice(Ng, rank, world)
        gg = torch.Generator(device='cuda').manual_seed(17)
        ys = [(torch.randn(Ng, Dg, device='cuda', generator=gg) * 2)[gsl].contiguous() for _ in range(4)]

        def fresh():
            vae.reset_variables()
            return SVAETrainer(Kg, Lg, Ug, Dg, nb_samples=Sg, lr=3e-3, lrcvi=0.2, decay_rate=0.95, stddev_init_nn=0.1, seed=3)
        tr_e = fresh()
        phi_first = [t.detach().clone() for t in tr_e.phi_gmm]
        el_e = [float(tr_e.step(ys[i])['elbo']) for i in range(4)]
        want = [t.detach().clone() for t in tr_e.trainables()[1]] + [t.clone() for t in tr_e.theta]
        Xte_e = (torch.randn(96, Dg, device='cuda', generator=torch.Generator(device='cuda').manual_seed(23)) * 2)
        lab_e = torch.nn.functional.one_hot(torch.randint(0, 3, (96,), device='cuda', generator=torch.Generator(device='cuda').manual_seed(24)), 3).float()
        m_e = experiments.evaluate(tr_e, Xte_e, lab_e, 4, seed=0)
        res['dpg_eval_eager'] = np.array([m_e[k_] for k_ in sorted(m_e)], dtype=np.float64)
        # round 6: the eager data-parallel step above ran as the direct kernel sequence up to the packed exchange buffer
        # (SVAETrainer._step_direct(pack=True) -> vmp_svae_step_pack); the autograd step over the stand-alone launches must agree
        # (its moments come from another kernel with another summation order: fp64, equal to rounding)
        res['dpd_direct'] = np.int64(tr_e._direct_ok(ys[0], None, None, None, None))
        vae.reset_variables()
        tr_a = SVAETrainer(Kg, Lg, Ug, Dg, nb_samples=Sg, lr=3e-3, lrcvi=0.2, decay_rate=0.95, stddev_init_nn=0.1, seed=3, direct_step=False)
        el_a = [float(tr_a.step(ys[i])['elbo']) for i in range(4)]
        got_a = list(tr_a.trainables()[1]) + list(tr_a.theta)
        res['dpd_elbo_direct'], res['dpd_elbo_autograd'] = np.array(el_e), np.array(el_a)
        res['dpd_param_err'] = np.array([((a.detach() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() for a, b in zip(got_a, want)])
        tr_g = fresh()
        gs = GraphedSVAEStep(tr_g, ys[0], warmup=2)
        res['dpg_two_graphs'] = np.int64(gs.graph_back is not None)
        el_g = [float(gs(ys[i])['elbo']) for i in range(4)]
        got = list(tr_g.trainables()[1]) + list(tr_g.theta)
        res['dpg_elbo_eager'], res['dpg_elbo_graphed'] = np.array(el_e), np.array(el_g)
        res['dpg_param_err'] = np.array([((a.detach() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() for a, b in zip(got, want)])
        res['dpg_params'] = np.concatenate([t.detach().cpu().numpy().reshape(-1) for t in got])
        res['dpg_steps'] = np.array([tr_g.global_step, tr_g.opt.t])
        # round 6: what runs AFTER graphed data-parallel steps must not depend on torch's device solvers - evaluation
        # metrics of the graphed trainer against the eager one's, and the package's K-sized factorisations (host LAPACK through
        # _klinalg) on device tensors against the same call on host tensors; the raw device Cholesky is recorded, not asserted
        from vmp_for_svae_amd.distributions import gaussian, niw
        from vmp_for_svae_amd.models import svae as svae_mod
        Xte = (torch.randn(96, Dg, device='cuda', generator=torch.Generator(device='cuda').manual_seed(23)) * 2)
        lab = torch.nn.functional.one_hot(torch.randint(0, 3, (96,), device='cuda', generator=torch.Generator(device='cuda').manual_seed(24)), 3).float()
        m_g = experiments.evaluate(tr_g, Xte, lab, 4, seed=0)
        res['dpg_eval_graphed'] = np.array([m_g[k_] for k_ in sorted(m_g)], dtype=np.float64)
        res['dpg_eval_keys'] = np.array(sorted(m_g))
        gh = torch.Generator().manual_seed(5)
        A_ = torch.randn(Kg, Lg, Lg, generator=gh)
        spd = A_ @ A_.transpose(-1, -2) + Lg * torch.eye(Lg)
        mu_ = torch.randn(Kg, Lg, generator=gh)
        dev_out = list(gaussian.standard_to_natural(mu_.cuda(), spd.cuda())) + list(gaussian.natural_to_standard(*gaussian.standard_to_natural(mu_.cuda(), spd.cuda()))) \
            + [niw._spd_inverse(spd.cuda()), svae_mod._recognition_bias(mu_.cuda(), -0.5 * spd.cuda(), torch.softmax(mu_[:, 0], 0).cuda())[1]]
        host_out = list(gaussian.standard_to_natural(mu_, spd)) + list(gaussian.natural_to_standard(*gaussian.standard_to_natural(mu_, spd))) \
            + [niw._spd_inverse(spd), svae_mod._recognition_bias(mu_, -0.5 * spd, torch.softmax(mu_[:, 0], 0))[1]]
        res['dpg_klinalg_err'] = np.array([((a.cpu() - b).abs().max() / b.abs().max()).item() for a, b in zip(dev_out, host_out)])
        res['dpg_raw_device_cholesky_err'] = np.float64((torch.linalg.cholesky(spd.cuda()).cpu() - torch.linalg.cholesky(spd)).abs().max().item())
        tr_l = fresh()                                          # a trainer built after the graphed steps == one built before them
        res['dpg_late_trainer_phi_err'] = np.float64(max((a.detach() - b.detach()).abs().max().item() for a, b in zip(tr_l.phi_gmm, phi_first)))
    except Exception as e:
        import traceback
        res['dpg_error'] = np.array(traceback.format_exc())
    dist.barrier()

    # ---- T1 again with the ONE-LAUNCH exchange (vmp_mix_finalize_exchange): the finalize kernels of the two processes push
    # their moments into each other's IPC-mapped buffers and sum them in rank order; no host staging, no collective library
    try:
        from vmp_for_svae_amd.models.parallel_mix import PeerExchange
        import time
        for name, flav in (('gmm', L.VMP_GMM), ('smm', L.VMP_SMM)):
            kap = torch.full((r0.shape[1],), 5.0, device='cuda') if flav == L.VMP_SMM else None
            ex = PeerExchange(r0.shape[1], x.shape[1])
            loop = DistributedVMPLoop(dev(x[sl]), dev(r0[sl]), flav, kappa=kap, exchange=ex)
            for it in range(3):
                r = loop.step()
                if it == 0:
                    res['t1x_%s_r1' % name] = r.cpu().numpy()
            res['t1x_%s_r' % name] = r.cpu().numpy()
            for n_, t in zip(('alpha', 'beta', 'm', 'C', 'v'), loop.theta()):
                res['t1x_%s_%s' % (name, n_)] = t.cpu().numpy()
            # many iterations back to back: parity double-buffering and sequence words under skew between the two processes
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for it in range(200):
                loop.step()
            torch.cuda.synchronize()
            res['t1x_%s_us_per_step' % name] = np.float64((time.perf_counter() - t0) / 200 * 1e6)
            res['t1x_%s_r200' % name] = loop.r.cpu().numpy()
            res['t1x_%s_m200' % name] = loop.theta()[2].cpu().numpy()
            res['t1x_%s_status' % name] = ex.status.cpu().numpy()
            ex.close_collective()                              # barrier over the ranks, then unmap / free
    except Exception as e:                                   # reported by the test, the other sections stay usable
        res['t1x_error'] = np.array(repr(e))
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
