#!/usr/bin/env python3
"""bench.py - headline benchmark of the VMP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload gmm|smm|t2|t3] [--scaling weak|strong]
                    [--n ROWS --d D --k K] [--exchange rccl|peer]

A "step" is one pass of the hot path over one batch of synthetic Gaussian-mixture data resident in HBM:
  gmm / smm  one VMP iteration (M-step -> E-step -> assign; reference models/gmm.py:258-263, smm.py:232-238).  Default =
             BASELINE.json configs[2] (GMM N=1e6, D=8, K=16); --workload smm = configs[4]'s model (C5)
  t2         the SVAE VMP step without the MLPs (fused E-step fwd + bwd, sub-sampling, M-step moments, CVI update)
  t3         the full SVAE training step of experiments.py:196-267 (configs[3]'s step function at the C3 shape; C4)
Multi-GPU (one rank per GPU): rows are sharded, ONE exchange of the K-sized statistics (+ gradients for t2 / t3) per step.
  --scaling weak    every rank holds N rows (default; `value` = N x ranks / step time)
  --scaling strong  the N rows are split over the ranks (rows / rank = N / G) - the reference's tf.split of ONE
                    minibatch over its towers (data.py:174-175); `value` = N / step time
With more than one rank the T1 line carries BOTH: the other mode is measured right after the headline and reported under
extra.other_scaling.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
FP32_PEAK_FLOPS = 157.3e12     # fp32 vector = fp32 MFMA peak (same guide)


def synth(N, D, K, seed):
    """SURVEY 8d generator: centres ~ N(0, 25 I), uniform labels, unit covariance; r0 = softmax(3 N(0,1))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.standard_normal((K, D)) * 5.0
    z = rng.integers(0, K, size=N)
    x = (c[z] + rng.standard_normal((N, D), dtype=np.float32)).astype(np.float32)
    r0 = np.exp(3.0 * rng.standard_normal((N, K), dtype=np.float32))
    r0 = (r0 / r0.sum(1, keepdims=True)).astype(np.float32)
    return x, r0


def cpu_baseline(x, r0, workload, chunk=1 << 15, reps=3):
    """The reference CPU path = the oracle (op-for-op torch-CPU fp32 restatement of models/gmm.py / models/smm.py, pinned by
    tests/golden), timed on this host: ONE VMP step through oracle.mixtures.{gmm,smm}_inference_step_chunked - the very
    function the full-size parity tests check the HIP path against - over a bounded sample of the workload (the literal
    graph materialises (N,K,D,D) temporaries, hence the row chunks); best of `reps`."""
    from oracle import mixtures
    Ns = min(x.shape[0], 1 << 18)                      # bounded sample: 262144 rows
    xs, rs = torch.as_tensor(x[:Ns]), torch.as_tensor(r0[:Ns])
    us = torch.ones_like(rs)

    def one_step():
        if workload == 'smm':
            return mixtures.smm_inference_step_chunked(xs, rs, us, 5.0, chunk=chunk)[0]
        return mixtures.gmm_inference_step_chunked(xs, rs, chunk=chunk)[0]

    # The thread count is part of the baseline: on the 256-hardware-thread host of the GPU box torch's intra-op pool
    # collapses when every thread is used (measured: 1.5e4 datapoints/s at 256 threads vs 7.3e5 at 32), so a few
    # pool sizes are tried and the FASTEST is reported, with its thread count in `cores`.
    ncpu = os.cpu_count() or 1
    cands = sorted({t for t in (8, 16, 32, 64) if t <= ncpu} | ({ncpu} if ncpu <= 64 else set())) or [ncpu]
    best, best_t = float('inf'), cands[0]
    for th in cands:
        torch.set_num_threads(th)
        one_step()
        for _ in range(reps):
            t0 = time.perf_counter()
            one_step()
            dt = time.perf_counter() - t0
            if dt < best:
                best, best_t = dt, th
    torch.set_num_threads(best_t)
    return {'value': Ns / best, 'unit': 'datapoints/s', 'cores': best_t, 'kind': 'port',
            'steps_per_sec_at_sample': 1.0 / best, 'host_hw_threads': ncpu, 'thread_counts_tried': cands,
            'sample': 'one %s VMP step (oracle.mixtures.*_inference_step_chunked, fp32, torch-CPU, N-chunks of %d) on the first %d rows of the workload; '
                      'fastest of %d timed runs at each of %s threads (best: %d threads)'
                      % (workload, chunk, Ns, reps, cands, best_t)}



def _best_threads(fn, cands, reps):
    """fastest of `reps` timed calls of fn() at each thread count (one untimed call first); (seconds, threads)"""
    best, best_t = float('inf'), cands[0]
    for th in cands:
        torch.set_num_threads(th)
        fn()
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            if dt < best:
                best, best_t = dt, th
    torch.set_num_threads(best_t)
    return best, best_t


def _cpu_model_inputs(K, Ld, Dy, U, Ns, S, seed=1):
    """synthetic inputs of the oracle's SVAE step (weights ~ N(0, 0.01^2), y = GMM data, eps ~ N(0,1))"""
    from oracle import nets, svae_ref
    rng = np.random.Generator(np.random.PCG64(seed))
    w = {}
    for scope, din, dout in (('encoder_net', Dy, Ld), ('decoder_net', Ld, Dy)):
        shapes = {'layer_0/kernel': (din, U), 'layer_0/bias': (U,), 'layer_1/kernel': (U, U), 'layer_1/bias': (U,),
                  'gaussian_output/kernel': (U, 2 * dout), 'gaussian_output/bias': (2 * dout,), 'shortcut/b1': (dout,),
                  'shortcut/b2': (dout,)}
        for n_, shp in shapes.items():
            w[scope + '/' + n_] = torch.as_tensor((rng.standard_normal(shp) * 0.01).astype(np.float32))
        w[scope + '/shortcut/W'] = torch.as_tensor(nets.rand_partial_isometry(din, dout, 1., 0).astype(np.float32))
    prior, theta = svae_ref.init_mm(K, Ld, torch.as_tensor(rng.random((K, Ld)).astype(np.float32)), torch.float32)
    phi = svae_ref.init_recognition_params(theta, torch.as_tensor(rng.standard_normal(K).astype(np.float32)))
    noise = torch.as_tensor(rng.standard_normal((Ns, K, Ld, S)).astype(np.float32))
    zd = torch.as_tensor(rng.integers(0, K, size=(Ns, S)))
    return rng, w, prior, theta, phi, noise, zd


def _smm_theta(prior, theta, rng, K, Ld):
    """Student-t theta of the oracle (svae.py:265-322): (alpha_nat, mu_k, L_k raw, DoF), components spread out"""
    from oracle import svae_ref
    mu_t, L_t = svae_ref.make_loc_scale(prior)
    mu_t = mu_t + torch.as_tensor(rng.standard_normal((K, Ld)).astype(np.float32))
    return [theta[0].clone(), mu_t, L_t, torch.full((K,), 5.0)], prior[0]


def cpu_baseline_t2(K, Ld, S, smm=False, Ns=1 << 14, reps=3, chunk=1 << 14):
    """The reference CPU path of the T2 unit ("fwd+bwd+M-step for T2/T3"): oracle.train_ref.vmp_step_t2 - the
    literal restatement of svae.e_step, the regulariser of compute_elbo(_smm), autodiff, subsample_x, m_step and
    update_gmm_params (models/svae.py:14-262, 376-403) in fp32 torch-CPU - on a bounded sample of Ns rows of the same shape."""
    from oracle import train_ref
    rng, _, prior, theta, phi, noise, zd = _cpu_model_inputs(K, Ld, Ld, 8, Ns, S)
    if smm:
        theta, prior = _smm_theta(prior, theta, rng, K, Ld)
    e1 = torch.as_tensor(rng.standard_normal((Ns, Ld)).astype(np.float32))
    e2 = -0.5 * torch.nn.functional.softplus(torch.as_tensor(rng.standard_normal((Ns, Ld)).astype(np.float32)))
    Gx = torch.as_tensor(rng.standard_normal((Ns, K, S, Ld)).astype(np.float32)) * 0.01
    Glz = torch.as_tensor(rng.standard_normal((Ns, K)).astype(np.float32)) * 0.1
    ncpu = os.cpu_count() or 1
    cands = sorted({t for t in (8, 32) if t <= ncpu}) or [ncpu]
    best, best_t = _best_threads(lambda: train_ref.vmp_step_t2(phi, theta, prior, e1, e2, noise, zd, Gx, Glz, 0.2, smm=smm, chunk=chunk), cands, reps)
    return {'value': Ns / best, 'unit': 'datapoints/s', 'cores': best_t, 'kind': 'port', 'host_hw_threads': ncpu,
            'thread_counts_tried': cands, 'seconds_per_step_at_sample': best,
            'sample': 'one T2 %s-svae VMP step (oracle.train_ref.vmp_step_t2: e_step fwd + regulariser + autodiff + subsample + m_step + CVI, '
                      'fp32 torch-CPU, N-chunks of %d) on %d rows of the workload shape (K=%d, L=%d, S=%d); fastest of %d timed runs at each of %s threads'
                      % ('smm' if smm else 'gmm', chunk, Ns, K, Ld, S, reps, cands)}


def cpu_baseline_t3(K, Ld, S, U, smm=False, Ns=1 << 14, reps=3, tower=1 << 14):
    """The reference CPU path of the T3 unit: oracle.train_ref.train_step (experiments.py:196-267 op for op: encoder, E-step,
    decoder, ELBO, autodiff of all 21 / 23 variables, M-step, CVI, TF-Adam) in fp32 torch-CPU on a bounded sample of Ns rows."""
    from oracle import nets, train_ref
    rng, w, prior, theta, phi, noise, zd = _cpu_model_inputs(K, Ld, Ld, U, Ns, S)
    if smm:
        theta, prior = _smm_theta(prior, theta, rng, K, Ld)
    st = train_ref.State(phi, {n_: w['encoder_net/' + n_] for n_ in nets.NET_VARS},
                         {n_: w['decoder_net/' + n_] for n_ in nets.NET_VARS}, theta, prior, smm=smm)
    y = torch.as_tensor(synth(Ns, Ld, K, seed=7)[0])
    ncpu = os.cpu_count() or 1
    cands = sorted({t for t in (8, 32) if t <= ncpu}) or [ncpu]
    best, best_t = _best_threads(lambda: train_ref.train_step(st, y, noise, zd, 3e-4, 0.2, 0.95, towers=max(1, Ns // tower)), cands, reps)
    return {'value': Ns / best, 'unit': 'datapoints/s', 'cores': best_t, 'kind': 'port', 'host_hw_threads': ncpu,
            'thread_counts_tried': cands, 'seconds_per_step_at_sample': best,
            'sample': 'one T3 %s-svae training step (oracle.train_ref.train_step = experiments.py:196-267, fp32 torch-CPU, towers of %d rows) '
                      'on %d rows of the workload shape (K=%d, L=Dy=%d, S=%d, U=%d); fastest of %d timed runs at each of %s threads'
                      % ('smm' if smm else 'gmm', tower, Ns, K, Ld, S, U, reps, cands)}


def bench_t2(N, Ld, K, S, steps, warmup, dev, dist=None, world=1, smm=False, cpu=False, tensor_mode=True):
    """T2: SVAE VMP step without the MLPs - fused E-step forward (log_z, samples, regulariser terms),
    its backward (given decoder-side gradients), categorical sub-sampling, M-step moments and the CVI update.
    The step draws its OWN noise, as sample_x_per_comp does (svae.py:113-114).  Headline (`ms_per_step`): the form SVAETrainer runs
    by default - eps generated inside the forward kernel, fresh key every step; algorithmic bytes 4N(2KSL + 2K + 4L).  Side
    measurement (`noise_tensor`): the same step with a noise TENSOR, its per-step normal_() INSIDE the timed step; 4N(4KSL + 2K + 4L).
    smm=True: Student-t theta (svae.py:265-322; theta/mu_k, theta/L_k trainable, M-step = N_k only, experiments.py:154-176)."""
    import vmp_for_svae_amd as V
    from vmp_for_svae_amd.models import svae, _mix, _svae_ops
    g = torch.Generator(device=dev).manual_seed(1234)
    eta1 = torch.randn(N, Ld, device=dev, generator=g).requires_grad_(True)
    eta2d = (-0.5 * torch.nn.functional.softplus(torch.randn(N, Ld, device=dev, generator=g))).requires_grad_(True)
    prior, theta = svae.init_mm(K, Ld, seed=0, param_device=dev)
    phi = [p.detach().requires_grad_(True) for p in svae.init_recognition_params(theta, K, seed=0, param_device=dev)]
    th_params = []
    if smm:
        mu_t, L_t = svae.make_loc_scale_variables(prior, dev)
        with torch.no_grad():
            mu_t.add_(torch.randn(K, Ld, device=dev, generator=g))
        theta = [theta[0].clone(), mu_t, L_t, torch.full((K,), 5.0, device=dev)]
        prior = prior[0]
        th_params = [mu_t, L_t]
    Gx = torch.randn(N, K, S, Ld, device=dev, generator=g) * 0.01
    Glz = torch.randn(N, K, device=dev, generator=g) * 0.1
    E = lambda: [torch.cuda.Event(enable_timing=True) for _ in range(steps)]

    def run(mode):
        """`warmup` + `steps` self-contained steps; returns (seconds, mean ms of [noise draw, forward, backward])"""
        noise = torch.empty(N, K, Ld, S, device=dev) if mode == 'tensor' else None
        ev = [E() for _ in range(5)]

        def one(i, timed):
            if timed:
                ev[0][i].record()
            if mode == 'tensor':
                noise.normal_(generator=g)                    # the step's draw (tf.random_normal, svae.py:113-114)
                if timed:
                    ev[1][i].record()
                x, lz, pt, _ = svae.e_step((eta1, eta2d), phi, S, noise=noise, theta=theta)
            else:
                if timed:
                    ev[1][i].record()
                x, lz, pt, _ = svae.e_step((eta1, eta2d), phi, S, seed=1000 + i, noise='philox', theta=theta)
            if timed:
                ev[2][i].record()
            # what the forward kernel's epilogue left (in-kernel noise): r = exp(log_z), the sub-sample, moment partials (K = 16)
            r = pt.r_nk if pt.r_nk is not None else torch.exp(lz.detach())
            if timed:
                ev[3][i].record()
            grads = torch.autograd.grad([x, lz, pt.T_prime], [eta1, eta2d] + phi + th_params, [Gx, Glz, r])
            if timed:
                ev[4][i].record()
            xs = pt.x_samples if pt.x_samples is not None else svae.subsample_x(x, lz, seed=1000 + i, nb_out=1, u='philox' if mode != 'tensor' else None)[:, 0, :].contiguous()
            mom = None if (smm or dist is not None) else pt.mom
            if smm:                                               # svae.m_step_smm: N_k only (svae.py:179-196)
                from vmp_for_svae_amd.models import gmm as _gmm
                st = _gmm.update_Nk(r.contiguous()).double().reshape(-1, 1)
            elif mom is not None:
                st = None                                         # partials -> moments -> CVI in ONE launch below, as SVAETrainer.step does
            elif pt.mom is not None:
                st = _svae_ops.mom_cvi(pt.mom)[0]
            else:
                st = _mix.raw_stats(xs, r, pivot=False)           # as SVAETrainer.step: the natural-parameter M-step uses the raw moments
            if dist is not None:
                from vmp_for_svae_amd.models.parallel_mix import allreduce_sum_
                buf = torch.cat([st.reshape(-1)] + [gg.reshape(-1).double() for gg in grads[2:]])
                allreduce_sum_(buf)                               # RCCL all-reduce of the packed fp64 buffer (gloo: host-staged)
                st = buf[:st.numel()].reshape(st.shape)
            if smm:
                svae.update_gmm_params(theta[:1], [prior + st[:, 0].float()], 0.2)
            elif mom is not None:
                _svae_ops.mom_cvi(mom, prior, theta, 0.2, want_star=False, want_stats=False)
            else:                                                 # svae.m_step + update_gmm_params in one launch, as SVAETrainer.step does
                svae.cvi_update_from_stats(prior, theta, st.double(), 0.2, want_star=False)

        for i in range(warmup):
            one(i, False)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            one(i, True)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = max_over_ranks(dist, time.perf_counter() - t0, dev)
        ms = [float(np.mean([a.elapsed_time(b) for a, b in zip(ev[j], ev[j + 1])])) for j in (0, 1, 3)]
        del noise
        torch.cuda.empty_cache()
        return dt_, ms

    dt, (_, f_ms, b_ms) = run('philox')
    if not tensor_mode:                                       # shard-size side measurements: the headline form only
        return {'ms_per_step': dt / steps * 1e3, 'fwd_kernel_ms': f_ms, 'bwd_kernel_ms': b_ms, 'rows': N}
    dt_n, (rn_ms, fn_ms, bn_ms) = run('tensor')
    alg = 4.0 * N * (2.0 * K * S * Ld + 2 * K + 4 * Ld)          # SURVEY 8d T2 bytes per step, in-kernel generator
    alg_n = 4.0 * N * (4.0 * K * S * Ld + 2 * K + 4 * Ld)        # ... with an injected noise tensor
    fwd_bytes = 4.0 * N * (1.0 * K * S * Ld + 2 * Ld + 2 * K)    # reads eta, writes x + log_z + T'
    fwd_bytes_n = 4.0 * N * (2.0 * K * S * Ld + 2 * Ld + 2 * K)  # + reads eps
    bwd_bytes = 4.0 * N * (2.0 * K * S * Ld + 4 * Ld + 3 * K)    # reads x + dx + (lz, dlz, dT'), writes d eta
    gb = lambda by, ms_: by / (ms_ * 1e-3) / 1e9
    res = {'steps_per_sec': steps / dt, 'ms_per_step': dt / steps * 1e3, 'datapoints_per_sec': N * world * steps / dt,
           'noise': 'drawn inside the forward kernel, fresh key every step (SVAETrainer default)',
           'algorithmic_bytes_per_step': alg, 'algorithmic_GBps_whole_step': alg / (dt / steps) / 1e9,
           'frac_hbm_whole_step': alg / (dt / steps) / 1e9 / HBM_PEAK_GBS,
           'fwd_kernel_ms': f_ms, 'bwd_kernel_ms': b_ms, 'tail_ms': dt / steps * 1e3 - f_ms - b_ms,
           # what the two kernels MOVE: the contract's algorithmic bytes have the backward read dL/dx only (x re-derived from eps); the
           # ring backward reads x AND dL/dx instead of regenerating eps and redoing the forward's VALU work (1.3 ms at C3)
           'moved_bytes_per_step': fwd_bytes + bwd_bytes, 'moved_over_algorithmic': (fwd_bytes + bwd_bytes) / alg,
           'moved_GBps_whole_step': (fwd_bytes + bwd_bytes) / (dt / steps) / 1e9,
           'moved_frac_hbm_whole_step': (fwd_bytes + bwd_bytes) / (dt / steps) / 1e9 / HBM_PEAK_GBS,
           'moved_note': 'the step moves fwd %.2f GB (eta in, x + log_z + T\' + r + x_samples out) + bwd %.2f GB (x, dL/dx and the (N,K) terms in, d eta out) '
                         'against %.2f GB algorithmic (SURVEY 8d counts dL/dx only for the backward): re-reading x costs 5.1 GB = 0.8 ms of HBM time, '
                         're-deriving it from the noise stream would cost the forward\'s 1.3 ms of VALU work again' % (fwd_bytes / 1e9, bwd_bytes / 1e9, alg / 1e9),
           'epilogue': 'sub-sampling draw, r = exp(log_z) and (K = 16, L = 8) the M-step moment partials come out of the forward kernel; reduce + CVI is one launch',
           'fwd_GBps': gb(fwd_bytes, f_ms), 'bwd_GBps': gb(bwd_bytes, b_ms),
           'fwd_frac_hbm': gb(fwd_bytes, f_ms) / HBM_PEAK_GBS, 'bwd_frac_hbm': gb(bwd_bytes, b_ms) / HBM_PEAK_GBS,
           'noise_tensor': {'ms_per_step': dt_n / steps * 1e3, 'includes': 'the per-step normal_() of the (N,K,L,S) tensor',
                            'randn_ms': rn_ms, 'fwd_kernel_ms': fn_ms, 'bwd_kernel_ms': bn_ms,
                            'algorithmic_bytes_per_step': alg_n, 'frac_hbm_whole_step': alg_n / (dt_n / steps) / 1e9 / HBM_PEAK_GBS,
                            'fwd_frac_hbm': gb(fwd_bytes_n, fn_ms) / HBM_PEAK_GBS, 'bwd_frac_hbm': gb(bwd_bytes, bn_ms) / HBM_PEAK_GBS},
           'config': 'T2 %s-svae-vmp N=%d per GPU, L=%d, K=%d, S=%d (own noise draw, fwd+bwd of the fused E-step, sub-sampling, M-step, CVI)' % ('smm' if smm else 'gmm', N, Ld, K, S)}
    if cpu:
        res['cpu_baseline'] = cpu_baseline_t2(K, Ld, S, smm=smm)
        res['speedup_vs_cpu_baseline'] = res['datapoints_per_sec'] / res['cpu_baseline']['value']
    return res


def bench_t3(N, Ld, K, S, U, steps, warmup, dev, chunk, smm=False, cpu=False, randn_mode=True):
    """T3: the full training step of experiments.py:196-267 (encoder / decoder MLP + reconstruction term in the fused
    MFMA kernels; fused E-step kernels; all gradients, TF-Adam, CVI) on synthetic y = GMM data with Dy = L.  Timed with
    eps drawn inside the E-step kernel (the trainer's default, rng='philox') and, next to it, read from a torch.randn
    noise tensor (rng='torch').  smm=True: the Student-t mixture SVAE (BASELINE configs[4]'s model, experiments.py:154-176).
    Every step is also bracketed by HIP events on the launch stream: min / median / max and the FIRST timed step are
    reported, so that a one-off cost that lands in the timed region (allocator growth after empty_cache, a lazily loaded
    code object) shows as one step and not as the rate (round 3's driver run: 118.6 ms/step over 3 steps after 1 warm-up,
    against 34 ms from every run with more warm-up)."""
    from vmp_for_svae_amd.models import vae
    from vmp_for_svae_amd.training import SVAETrainer
    if warmup < 3 or steps < 5:
        print('bench_t3: warmup %d -> %d, steps %d -> %d (the caching allocator needs 3 steps to reach its steady state)'
              % (warmup, max(3, warmup), steps, max(5, steps)), file=sys.stderr)
    warmup, steps = max(3, warmup), max(5, steps)
    x_h, _ = synth(N, Ld, K, seed=7)
    y = torch.as_tensor(x_h).to(dev)

    def timed(rng):
        vae.reset_variables()
        tr = SVAETrainer(K, Ld, U, Ld, nb_samples=S, device=dev, rng=rng, smm=smm)
        for _ in range(warmup):
            tr.step(y, chunk=chunk)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            ev[i][0].record()
            out_ = tr.step(y, chunk=chunk)
            ev[i][1].record()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        per = [a.elapsed_time(b) for a, b in ev]
        elbo = float(out_['elbo'])
        del tr, out_
        torch.cuda.empty_cache()
        return dt_, per, elbo

    dt, per, elbo = timed('philox')
    if not randn_mode:                                        # shard-size side measurements: the headline form only
        return {'ms_per_step': dt / steps * 1e3, 'per_step_ms_median': float(np.median(per)), 'rows': N}
    dt_p, per_p, _ = timed('torch')
    rows = float(N) * K * S
    dec_flop = 3 * 2.0 * rows * (Ld * U + U * U + U * 2 * Ld + Ld * Ld)     # fwd + 2x bwd, useful flops of the decoder

    def stats(v):
        return {'first': v[0], 'min': min(v), 'median': float(np.median(v)), 'max': max(v)}
    # ms_per_step = MEDIAN of the per-step HIP-event times (as the headline's median over its timed regions): one step of a run now and
    # then takes 40-47 ms - the first after the warm-up, the caching allocator re-growing a 5 GB block - and a 5-step mean then reads
    # 31 ms for a 27.5 ms step (round-6 driver-command run: per-step 47.0, 27.3, 27.5, 27.6, 27.6).  The wall-clock mean stays beside it.
    med = float(np.median(per)) * 1e-3
    res = {'steps_per_sec': 1.0 / med, 'ms_per_step': med * 1e3, 'ms_per_step_wall_mean': dt / steps * 1e3, 'datapoints_per_sec': N / med,
           'timing': 'median of the per-step HIP-event times over `steps` steps; wall-clock mean beside it',
           'steps': steps, 'warmup': warmup, 'per_step_ms': stats(per),
           # the same step with a torch.randn noise tensor: the SAME two statistics as the headline (wall-clock mean over the timed
           # region, per-step HIP-event min / median / max) - compare like with like
           'noise_tensor_randn': {'ms_per_step': float(np.median(per_p)), 'ms_per_step_wall_mean': dt_p / steps * 1e3, 'per_step_ms': stats(per_p)},
           'elbo_per_datapoint': elbo / N, 'decoder_rows_per_step': rows,
           'decoder_useful_TFLOPs_over_whole_step': dec_flop / med / 1e12,
           'config': 'T3 %s-svae-train N=%d (%s), L=Dy=%d, K=%d, S=%d, U=%d' % (
               'smm' if smm else 'gmm', N, 'one pass' if chunk is None else 'chunks of %d' % chunk, Ld, K, S, U)}
    if cpu:
        res['cpu_baseline'] = cpu_baseline_t3(K, Ld, S, U, smm=smm)
        res['speedup_vs_cpu_baseline'] = res['datapoints_per_sec'] / res['cpu_baseline']['value']
    return res


def bench_minibatch(N, K, Ld, Dy, S, U, dev, steps=200, cpu=True):
    """The reference's own operating point (experiments.py:26,56-66: minibatches of 64 rows, Auto-sized model): the full
    training step eager, the same step replayed from one HIP graph, and the oracle's literal restatement of
    experiments.py:196-267 on the host cores (fp32)."""
    from vmp_for_svae_amd.models import vae
    from vmp_for_svae_amd.training import SVAETrainer, GraphedSVAEStep
    vae.reset_variables()
    g = torch.Generator(device=dev).manual_seed(99)
    y = torch.randn(N, Dy, device=dev, generator=g) * 2
    tr = SVAETrainer(K, Ld, U, Dy, nb_samples=S, device=dev)
    for _ in range(10):
        tr.step(y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(y)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps
    gs = GraphedSVAEStep(tr, y)
    for _ in range(10):
        gs(y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = gs(y)
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / steps
    assert torch.isfinite(out['elbo'])
    # the minibatch already in the step's static input (a loader that copies its minibatch straight into gs.y): replay only
    gs.y.copy_(y)
    for _ in range(10):
        gs(gs.y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = gs(gs.y)
    torch.cuda.synchronize()
    resident = (time.perf_counter() - t0) / steps
    assert torch.isfinite(out['elbo'])
    # four consecutive steps per replay (GraphedSVAEStep(steps_per_replay=4): step i reads minibatch i of the static input and its own
    # row of the scalar table - the same four steps the eager trainer takes); the four minibatches are another device tensor, copied in
    # with one launch per replay
    y4 = torch.randn(4, N, Dy, device=dev, generator=g) * 2
    gs4 = GraphedSVAEStep(tr, y, steps_per_replay=4)
    for _ in range(5):
        gs4(y4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps // 4):
        outs = gs4(y4)
    torch.cuda.synchronize()
    graphed4 = (time.perf_counter() - t0) / (4 * (steps // 4))
    assert all(torch.isfinite(o['elbo']) for o in outs)
    res = {'config': 'T3 svae-train minibatch N=%d, K=%d, L=%d, Dy=%d, S=%d, U=%d' % (N, K, Ld, Dy, S, U),
           'eager_steps_per_sec': 1.0 / eager, 'eager_ms_per_step': eager * 1e3,
           'graphed_steps_per_sec': 1.0 / graphed, 'graphed_ms_per_step': graphed * 1e3,
           'graphed_input_resident_ms_per_step': resident * 1e3,
           'graphed_4_steps_per_replay_ms_per_step': graphed4 * 1e3,
           'graphed_note': 'graphed: the minibatch is another device tensor, copied into the static input per call (one eager copy + '
                           'the replay); input_resident: it already sits there (replay only); 4_steps_per_replay: four consecutive '
                           'steps in one graph, their four minibatches copied in by one launch per replay'}
    if cpu:
        from oracle import nets, svae_ref, train_ref
        # tiny tensors: more than a few threads only adds synchronisation (256 threads: ~30 s per step); the fastest
        # of a few pool sizes is reported
        rng = np.random.Generator(np.random.PCG64(1))
        w = {}
        for scope, din, dout in (('encoder_net', Dy, Ld), ('decoder_net', Ld, Dy)):
            shapes = {'layer_0/kernel': (din, U), 'layer_0/bias': (U,), 'layer_1/kernel': (U, U), 'layer_1/bias': (U,),
                      'gaussian_output/kernel': (U, 2 * dout), 'gaussian_output/bias': (2 * dout,), 'shortcut/b1': (dout,),
                      'shortcut/b2': (dout,)}
            for n_, shp in shapes.items():
                w[scope + '/' + n_] = torch.as_tensor((rng.standard_normal(shp) * 0.01).astype(np.float32))
            w[scope + '/shortcut/W'] = torch.as_tensor(nets.rand_partial_isometry(din, dout, 1., 0).astype(np.float32))
        prior, theta = svae_ref.init_mm(K, Ld, torch.as_tensor(rng.random((K, Ld)).astype(np.float32)), torch.float32)
        phi = svae_ref.init_recognition_params(theta, torch.as_tensor(rng.standard_normal(K).astype(np.float32)))
        st = train_ref.State(phi, {n_: w['encoder_net/' + n_] for n_ in nets.NET_VARS},
                             {n_: w['decoder_net/' + n_] for n_ in nets.NET_VARS}, theta, prior)
        yc = y.cpu()
        noise = torch.as_tensor(rng.standard_normal((N, K, Ld, S)).astype(np.float32))
        zd = torch.as_tensor(rng.integers(0, K, size=(N, S)))
        ncpu = os.cpu_count() or 1
        cpu_t, cpu_th = float('inf'), 1
        for th in sorted({t for t in (1, 8, 32) if t <= ncpu}):
            torch.set_num_threads(th)
            train_ref.train_step(st, yc, noise, zd, 3e-4, 0.2, 0.95)
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                train_ref.train_step(st, yc, noise, zd, 3e-4, 0.2, 0.95)
            dt = (time.perf_counter() - t0) / reps
            if dt < cpu_t:
                cpu_t, cpu_th = dt, th
        res['cpu_oracle_steps_per_sec'] = 1.0 / cpu_t
        res['cpu_oracle_cores'] = cpu_th
    return res


def max_over_ranks(dist, value, dev):
    """MAX over ranks of a host scalar (the contract's timing rule); device tensor for RCCL, host tensor for gloo"""
    if dist is None:
        return value
    on_dev = dist.get_backend() == 'nccl'
    tt = torch.tensor([value], device=dev if on_dev else 'cpu', dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return tt.item()


def time_t1(loop, steps, warmup, reps, barrier, dist, dev):
    """`reps` repetitions of the contract's timed region (barrier + synchronize, EXACTLY `steps` steps, barrier +
    synchronize; MAX over ranks), each preceded by nothing but the previous region.  Returns the per-region wall times (s)
    and the mean duration (ms) of the dominant kernel, bracketed by HIP events on the launch stream on every EV_EVERY-th
    step (an event record between two dependent launches costs a few microseconds of gap: bracketing every step inflates
    the step)."""
    EV_EVERY = 4
    for _ in range(warmup):
        loop.step()
    walls, kern = [], []
    for _ in range(reps):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range((steps + EV_EVERY - 1) // EV_EVERY)]
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            loop.finalize_phase()
            if i % EV_EVERY == 0:
                ev[i // EV_EVERY][0].record()
                loop.stream_phase()                      # exactly one launch: the streaming pass (one-launch form: with the posterior in its head)
                ev[i // EV_EVERY][1].record()
            else:
                loop.stream_phase()
        barrier()
        dt = max_over_ranks(dist, time.perf_counter() - t0, dev)
        walls.append(dt)
        kern.append(float(np.mean([a.elapsed_time(b) for a, b in ev])))
    return walls, kern


def t1_flops(N, D, K):
    """F ~ N K [(2 D^2 + 3 D) + 2 (D + D^2) + 12] (Mahalanobis form + moments + softmax)."""
    return float(N) * K * ((2 * D * D + 3 * D) + 2 * (D + D * D) + 12)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh worker processes of this script (one rank per GPU,
    RCCL rendezvous on 127.0.0.1) BEFORE anything here has touched the GPU, wait for them and pass rank 0's JSON line
    through.  (Under torchrun the ranks already exist and this is not used.)"""
    import socket
    import subprocess
    s_ = socket.socket()
    s_.bind(('127.0.0.1', 0))
    port = s_.getsockname()[1]
    s_.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def measure_traffic(workload, N, D, K):
    """HBM bytes per launch of the fused pass kernel, measured NOW: two child processes run tools/t1_prof_target.py under
    `rocprofv3 --kernel-trace --pmc` (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) BEFORE
    this process touches the GPU; FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 bytes for wide coalesced reads),
    WRITE_SIZE x 1.  Returns (bytes, description) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    out = tempfile.mkdtemp(prefix='vmp_pmc_', dir='/tmp')
    env = dict(os.environ, N=str(N), D=str(D), K=str(K), FLAV=workload, REPS='3', TMPDIR='/tmp')
    vals = {}
    try:
        for tag, ctr in (('f', 'FETCH_SIZE'), ('w', 'WRITE_SIZE')):
            subprocess.run([exe, '--kernel-trace', '--pmc', ctr, '--output-format', 'csv', '-d', out, '-o', tag, '--',
                            sys.executable, os.path.join(ROOT, 'tools', 't1_prof_target.py')], cwd='/tmp', env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            v = []
            for f in glob.glob(os.path.join(out, '**', '*%s_counter_collection.csv' % tag), recursive=True):
                for row in csv.DictReader(open(f)):
                    n = row['Kernel_Name']
                    # pass_xdl_kernel<D, FLAV, STATS = true, MT> / pass_kernel<D, KT, FLAV, ESTEP = true, STATS = true, MASK = false>
                    fused = ('pass_xdl_kernel<' in n and ', true' in n.split('>')[0]) or \
                            ('pass_kernel<' in n and 'true, true, false>' in n)
                    if fused and row['Counter_Name'] == ctr:
                        v.append(float(row['Counter_Value']))
            if not v:
                return None, 'no fused pass kernel dispatch in the %s pass' % ctr
            vals[ctr] = sum(v) / len(v)
        return (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0, \
            ('measured by this run: rocprofv3 --kernel-trace --pmc, separate passes, %s launches averaged; FETCH_SIZE %.1f KB x 2 + '
             'WRITE_SIZE %.1f KB' % (len(v), vals['FETCH_SIZE'], vals['WRITE_SIZE']))
    except Exception as e:                                  # the bench line must not depend on the profiler
        return None, 'rocprofv3 pass failed: %r' % (e,)
    finally:
        shutil.rmtree(out, ignore_errors=True)


def shard_rows(n_total, rank, world):
    """contiguous near-equal row ranges (tf.split semantics, data.py:174-175; remainder to the first ranks)"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def dec_bwd_side_measurement(N, K, S, Ld, U, dev):
    """The dominant kernel of the T3 step (fused decoder backward: value + all gradients of the reconstruction term from
    one launch) timed on its own stream position with HIP events on a bounded row count, for the t3 roofline."""
    from vmp_for_svae_amd.models import vae, _svae_ops
    n = min(N, 1 << 20)                               # C3's N = 1e6 is timed whole (round 5; 262 144 rows scaled up over-stated it by 8 %)
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(n, K, S, Ld, device=dev, generator=g).requires_grad_(True)
    y = torch.randn(n, Ld, device=dev, generator=g)
    w = torch.softmax(torch.randn(n, K, device=dev, generator=g), 1)
    params = [p_ for _, p_ in vae.net_variables('decoder_net')]
    ts = []
    for i in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _svae_ops.DecoderWeightedLoglikeFn.apply(y, x, w, *params)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts[1:]))
    rows = float(n) * K * S
    useful = 3 * 2.0 * rows * (Ld * U + U * U + U * 2 * Ld + Ld * Ld)     # fwd recompute + 2 x bwd
    # issued v_mfma_f32_16x16x32_bf16 per 16-row tile (csrc/vmp_decoder.hip): UT 16-unit tiles, KB 32-unit k-blocks; the
    # backward data path multiplies 2-term splits (3 products) from 2^19 rows on, 3-term splits (6 products) below
    UT = (U + 15) // 16
    KB = (UT + 1) // 2
    p = 3 if float(N) * K * S >= 2 ** 19 else 6
    # (round 5: the forward recompute inside the backward kernel follows the data path's operand terms: p products per k-block)
    mf = (2 * UT + p * UT * KB + p * KB + 2) + ((3 if p == 6 else 2) * UT + 2 * UT + 2 + UT * KB * p + 2 * UT * UT + KB * p + (3 if p == 6 else 2) + 2 * UT)
    issued = mf * 16384.0 * rows / 16.0
    return ms * (float(N) / n), useful * (float(N) / n), issued * (float(N) / n), mf


def t3_roofline(N, K, S, Ld, U, dev):
    """roofline object of the T3 step's dominant kernel (fused decoder backward), from dec_bwd_side_measurement"""
    k_ms, useful, issued, mf = dec_bwd_side_measurement(N, K, S, Ld, U, dev)
    tf_s = useful / (k_ms * 1e-3) / 1e12
    is_s = issued / (k_ms * 1e-3) / 1e12
    return {'bound': 'mfma', 'achieved': is_s, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': is_s / 2500.0, 'traffic': None,
            'kernel': 'dec_bwd_kernel (bf16 split operands on the XDL pipe: %d v_mfma_f32_16x16x32_bf16 per 16-row tile)' % mf,
            'kernel_ms': k_ms, 'useful_fp32_TFLOPs': tf_s, 'useful_frac_of_fp32_vector_peak': tf_s * 1e12 / FP32_PEAK_FLOPS,
            'issued_over_useful': issued / useful,
            'note': 'timed on min(N, 2^20) data rows (scaled to N beyond that; the kernel is linear in rows); achieved = ISSUED bf16 MFMA '
                    'flops (operand splits, 50 -> 64 unit padding and k-slot padding included), useful = fp32-equivalent flops'}


def shard_steps(N, D, K, S, U, flav, kappa, dev, ms1, exch_us, exch_detail):
    """Strong scaling on driver-timed ground without a multi-GPU box: the step a rank of a G-GPU run executes on its N / G rows (the
    reference's tf.split of one minibatch over its towers, data.py:174-175; experiments.py:196-248), timed HERE for G = 2, 4, 8 and
    all three units, plus the one-rank cost of the step's single exchange (T1: the RCCL / peer forms of t1_forced_dist_1rank; T2 / T3:
    one in-place RCCL all-reduce of the packed fp64 buffer).  Implied factors: strong_G = t(N) / (t(N / G) + exchange), weak_G = G t(N) /
    (t(N) + exchange).  The inter-GPU hop itself (xGMI latency, ~2 us per store-and-poll, 15-25 us for an 8-rank ring of a 20-100 KB
    message) is NOT in these numbers: nothing here can measure it."""
    from vmp_for_svae_amd.models import _mix
    out = {'rows_per_rank': {}, 'note': 'one-GPU timings of the per-rank step at N/G rows; exchange = one-rank overhead measured by this run; no xGMI hop included'}
    t = {'t1': {}, 't2': {}, 't3': {}}
    for G in (2, 4, 8):
        n = N // G
        x_h, r0_h = synth(n, D, K, seed=50 + G)
        lp = _mix.VMPLoop(torch.as_tensor(x_h).to(dev), torch.as_tensor(r0_h).to(dev), flav, kappa=kappa)
        w, _ = time_t1(lp, 20, 5, 7, lambda: torch.cuda.synchronize(), None, dev)
        t['t1'][G] = float(np.median(w)) / 20 * 1e3
        del lp
        torch.cuda.empty_cache()
        t['t2'][G] = bench_t2(n, D, K, S, 6, 2, dev, None, 1, cpu=False, tensor_mode=False)['ms_per_step']
        torch.cuda.empty_cache()
        t['t3'][G] = bench_t3(n, D, K, S, U, 5, 3, dev, None, cpu=False, randn_mode=False)['per_step_ms_median']     # (median of the per-step HIP-event times:
        #                                                            one allocator-growth step in five would otherwise decide the factor)
        torch.cuda.empty_cache()
        out['rows_per_rank'][str(G)] = n
    ex = {'t1': exch_us['t1_peer'] * 1e-3, 't2': exch_us['t2'] * 1e-3, 't3': exch_us['t3'] * 1e-3}       # ms
    for u in ('t1', 't2', 't3'):
        out[u] = {'ms_per_step_full': ms1[u], 'ms_per_step_at_rows_per_rank': {str(G): t[u][G] for G in (2, 4, 8)},
                  'exchange_ms_1rank': ex[u],
                  'implied_strong_scaling': {str(G): ms1[u] / (t[u][G] + ex[u]) for G in (2, 4, 8)},
                  'implied_weak_scaling': {str(G): G * ms1[u] / (ms1[u] + ex[u]) for G in (2, 4, 8)}}
    out['t1']['exchange_ms_1rank_rccl_form'] = exch_us['t1_rccl'] * 1e-3
    out['t1']['implied_strong_scaling_rccl_form'] = {str(G): ms1['t1'] / (t['t1'][G] + exch_us['t1_rccl'] * 1e-3) for G in (2, 4, 8)}
    out['exchange_detail'] = exch_detail
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--reps', type=int, default=31, help='repetitions of the timed region (gmm / smm); the median is reported')
    ap.add_argument('--workload', default='gmm', choices=['gmm', 'smm', 't2', 't3'])
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--exchange', default='auto', choices=['auto', 'rccl', 'peer'],
                    help='T1 multi-rank exchange: rccl = local reduce + RCCL all-reduce + posterior (3 launches); '
                         'peer = ONE finalize launch pushing the moments into IPC-mapped peer buffers; auto (default) = peer where every '
                         'rank could map the buffers, three steps of it agree with the RCCL form, no wait timed out AND it is the faster '
                         'of the two on every rank - otherwise rccl')
    ap.add_argument('--n', type=int, default=1_000_000)
    ap.add_argument('--d', type=int, default=8)
    ap.add_argument('--k', type=int, default=16)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the side measurements (T2 / T3 / N=1e7 / forced-dist)')
    ap.add_argument('--no-traffic', action='store_true', help='do not run the two rocprofv3 counter passes that measure roofline.traffic')
    ap.add_argument('--s', type=int, default=10)
    ap.add_argument('--smm', action='store_true', help='t2 / t3: the Student-t mixture SVAE (BASELINE configs[4]: compute_elbo_smm, trainable theta/mu_k, theta/L_k)')
    ap.add_argument('--u', type=int, default=50)
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        ndev = torch.cuda.device_count()                 # does not initialise the GPU on this image
        if ndev < args.gpus:
            sys.exit('bench.py: --gpus %d but only %d GPU(s) visible' % (args.gpus, ndev))
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    # stdout carries exactly ONE line (the JSON): everything else this process or its libraries print (RCCL's version
    # banner, warnings) goes to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # roofline.traffic of the dominant kernel, measured by THIS run (single-GPU T1 runs; child processes, started before
    # this process initialises the GPU)
    traffic_live = (None, 'not measured (multi-rank run, other workload, or --no-traffic / --no-extra)')
    if (int(os.environ.get('WORLD_SIZE', '1')) == 1 and args.workload in ('gmm', 'smm') and not args.no_traffic
            and not args.no_extra):
        traffic_live = measure_traffic(args.workload, args.n, args.d, args.k)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d does not match WORLD_SIZE=%d' % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('VMP_BENCH_BACKEND', 'nccl')    # tests: 'gloo' lets two ranks share ONE GPU (RCCL refuses that)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    import vmp_for_svae_amd as V
    from vmp_for_svae_amd.models import _mix
    from vmp_for_svae_amd.models.parallel_mix import DistributedVMPLoop, PeerExchange
    L = V._lib
    L.lib()                                              # fail loudly if the HIP library is missing

    N, D, K = args.n, args.d, args.k

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def rows_for(mode):
        """(rows on this rank, rows of the whole job) under weak / strong scaling"""
        if mode == 'weak':
            return N, N * world
        lo, hi = shard_rows(N, rank, world)
        return hi - lo, N

    common = {'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'higher_is_better': True, 'scaling': args.scaling,
              'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'unit': 'datapoints/s'}
    out, extra = None, {}

    if args.workload in ('gmm', 'smm'):
        flav = L.VMP_SMM if args.workload == 'smm' else L.VMP_GMM
        kappa = torch.full((K,), 5.0, device=dev) if flav == L.VMP_SMM else None

        chosen = {'exchange': args.exchange}

        def pick_exchange(x_, r0_):
            """--exchange auto, decided once per launch by all ranks together (every branch below is taken by every rank): the peer
            form must (a) open on every rank, (b) reproduce the RCCL form's responsibilities after three steps on this rank's rows
            without a timed-out wait, (c) be faster than it over 30 timed steps on every rank."""
            ex = PeerExchange.try_open(K, D)
            if ex is None:
                chosen.update(exchange='rccl', why='peer buffers could not be opened: ' + PeerExchange.last_failure)
                return None
            ok, why, t = 1, '', {}
            try:
                loops = {'peer': DistributedVMPLoop(x_, r0_, flav, kappa=kappa, exchange=ex), 'rccl': DistributedVMPLoop(x_, r0_, flav, kappa=kappa)}
                for name, lp_ in loops.items():
                    barrier()
                    for _ in range(3):
                        lp_.step()
                    barrier()
                    t0 = time.perf_counter()
                    for _ in range(30):
                        lp_.step()
                    torch.cuda.synchronize()
                    t[name] = time.perf_counter() - t0
                err = (loops['peer'].r - loops['rccl'].r).abs().max().item()
                if int(ex.status.item()) != 0:
                    ok, why = 0, 'a wait of the peer form timed out'
                elif not err <= 1e-5:
                    ok, why = 0, 'peer and RCCL forms disagree (max |dr| = %.3g)' % err
                elif not t['peer'] < t['rccl']:
                    ok, why = 0, 'peer form slower on rank %d (%.1f vs %.1f us/step)' % (rank, t['peer'] / 30 * 1e6, t['rccl'] / 30 * 1e6)
                del loops
            except Exception as e:                           # noqa: BLE001
                ok, why = 0, 'rank %d: %r' % (rank, e)
            flag = torch.tensor([ok], dtype=torch.int32, device=dev if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                # a FRESH exchange for the measured loop: iteration counters and sequence words start from zero on every rank
                ex.close_collective()
                ex2 = PeerExchange.try_open(K, D)
                if ex2 is not None:
                    chosen.update(exchange='peer', why='verified against the RCCL form and faster: %.1f vs %.1f us/step on rank 0'
                                  % (t['peer'] / 30 * 1e6, t['rccl'] / 30 * 1e6) if t else '')
                    return ex2
                chosen.update(exchange='rccl', why='peer buffers could not be re-opened: ' + PeerExchange.last_failure)
                return None
            ex.close_collective()
            chosen.update(exchange='rccl', why=why or 'another rank rejected the peer form')
            return None

        def make_loop(n_rows, seed):
            x_h_, r0_h_ = synth(n_rows, D, K, seed=seed)
            x_, r0_ = torch.as_tensor(x_h_).to(dev), torch.as_tensor(r0_h_).to(dev)
            if world > 1:
                if args.exchange == 'auto' and chosen['exchange'] == 'auto':
                    ex = pick_exchange(x_, r0_)
                elif chosen['exchange'] == 'peer':
                    ex = PeerExchange(K, D)                 # (its constructor ends with a rendezvous of the ranks)
                else:
                    ex = None
                lp = DistributedVMPLoop(x_, r0_, flav, kappa=kappa, exchange=ex)
            else:
                lp = _mix.VMPLoop(x_, r0_, flav, kappa=kappa)
            return lp, x_h_, r0_h_, x_, r0_

        n_loc, n_job = rows_for(args.scaling)
        loop, x_h, r0_h, x, r0 = make_loop(n_loc, seed=rank)      # every rank: its own shard of the job's rows
        barrier()                                                 # every rank has its data and its loop before the first exchange
        walls, kerns = time_t1(loop, args.steps, args.warmup, max(1, args.reps), barrier, dist, dev)
        dt = float(np.median(walls))
        kern_ms = float(np.median(kerns))
        assert torch.isfinite(loop.r).all()
        if world > 1:
            loop.check()                                          # a timed-out in-kernel wait voids the line: fail, do not print it
        if world == 1 and args.workload == 'smm':
            # the opt-in accurate E-part (fp64 from an fp64 pack + a separate M-pass: VMPLoop(accurate=True)) beside the default
            aloop = _mix.VMPLoop(x, r0, flav, kappa=kappa, accurate=True)
            aw, _ = time_t1(aloop, args.steps, args.warmup, 7, barrier, None, dev)
            extra['accurate_mode'] = {'ms_per_step': float(np.median(aw)) / args.steps * 1e3, 'default_ms_per_step': dt / args.steps * 1e3,
                                      'what': 'VMPLoop(accurate=True): vmp_mix_finalize_ws64 + vmp_mix_estep_accurate (fp64 E-part) + vmp_mix_stats_ws_accurate (fp64 M-pass); '
                                              'meets the literal 1e-5 on r_nk at C5 (tests/test_fullsize_gpu.py smm-c5-accurate)'}
            del aloop
        if world > 1:
            # the OTHER scaling mode, same launch: fewer repetitions, same timed-region protocol
            other = 'strong' if args.scaling == 'weak' else 'weak'
            if loop.exchange is not None:
                loop.exchange.close_collective()                  # barrier, then unmap: no peer is polling or pushing any more
            barrier()
            del loop
            torch.cuda.empty_cache()
            n2, job2 = rows_for(other)
            loop2 = make_loop(n2, seed=100 + rank)[0]
            barrier()
            w2, k2 = time_t1(loop2, args.steps, args.warmup, 7, barrier, dist, dev)
            loop2.check()
            d2 = float(np.median(w2))
            extra['other_scaling'] = {'scaling': other, 'rows_per_rank': n2, 'rows_job': job2, 'ms_per_step': d2 / args.steps * 1e3,
                                      'value': job2 / (d2 / args.steps), 'kernel_ms': float(np.median(k2)),
                                      'exchange': chosen['exchange']}
            if loop2.exchange is not None:
                loop2.exchange.close_collective()
            barrier()
            del loop2
        # side measurements: single-GPU experiments (nothing after the timed region may take the JSON line down with it)
        if not args.no_extra and world == 1:
            del loop
            torch.cuda.empty_cache()
            # (a) what the data-parallel iteration costs on top of the plain one, with ONE rank: the 3-launch RCCL form
            #     (local reduction -> all-reduce -> posterior + pack) and the ONE-launch peer-exchange form
            import torch.distributed as dist1
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            dist1.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
            dloop = DistributedVMPLoop(x, r0, flav, kappa=kappa)
            dw, _ = time_t1(dloop, args.steps, args.warmup, max(1, args.reps), lambda: torch.cuda.synchronize(), None, dev)
            # the ONE collective of a T2 / T3 data-parallel step (packed fp64 buffer: moments + K-sized gradients [+ MLP gradients]),
            # in place on this stream, one rank: what the exchange adds to a step before any inter-GPU hop
            exch_us = {}
            for tag, nd in (('t2', K * (2 + D + D * D) + K * (D + D * D + 1) + 3), ('t3', K * (2 + D + D * D) + K * (D + D * D + 1) + 3 + 2 * (D * args.u + args.u + args.u * args.u + args.u + args.u * 2 * D + 2 * D + D * D + 2 * D))):
                bufx = torch.zeros(nd, dtype=torch.float64, device=dev)
                for _ in range(5):
                    dist1.all_reduce(bufx)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(50):
                    dist1.all_reduce(bufx)
                e1.record()
                torch.cuda.synchronize()
                exch_us[tag] = {'doubles': nd, 'allreduce_us_1rank': e0.elapsed_time(e1) / 50 * 1e3}
            dist1.destroy_process_group()
            d_us = float(np.median(dw)) / args.steps * 1e6
            del dloop
            ex = PeerExchange(K, D, rank=0, world=1, gather=lambda h: [h])
            ploop = DistributedVMPLoop(x, r0, flav, kappa=kappa, exchange=ex)
            pw, _ = time_t1(ploop, args.steps, args.warmup, max(1, args.reps), lambda: torch.cuda.synchronize(), None, dev)
            p_us = float(np.median(pw)) / args.steps * 1e6
            ex.check()
            del ploop
            ex.close()
            plain = dt / args.steps * 1e6
            extra['t1_forced_dist_1rank'] = {'plain_us_per_step': plain, 'rccl_us_per_step': d_us, 'rccl_overhead_us': d_us - plain,
                                             'peer_exchange_us_per_step': p_us, 'dist_overhead_us': p_us - plain}
            del x, r0
            torch.cuda.empty_cache()
            if (N, D, K) == (1_000_000, 8, 16):
                # (b) the cache-defeating size of SURVEY 8d: N=1e7 (x + r = 960 MB >> the 256 MiB Infinity Cache)
                Nb = 10_000_000
                g = torch.Generator(device=dev).manual_seed(5)
                cb = torch.randn(K, D, device=dev, generator=g) * 5
                xb = cb[torch.randint(0, K, (Nb,), device=dev, generator=g)] + torch.randn(Nb, D, device=dev, generator=g)
                rb = torch.softmax(3 * torch.randn(Nb, K, device=dev, generator=g), dim=1)
                bloop = _mix.VMPLoop(xb, rb, flav, kappa=kappa)
                bw, bk = time_t1(bloop, 20, 5, 7, lambda: torch.cuda.synchronize(), None, dev)
                b_ms, bk_ms = float(np.median(bw)) / 20 * 1e3, float(np.median(bk))
                wordsb = (2 * D + 2 * K) if flav == L.VMP_GMM else (2 * D + 4 * K)
                extra['t1_n1e7'] = {'ms_per_step': b_ms, 'datapoints_per_sec': Nb / (b_ms * 1e-3), 'kernel_ms': bk_ms,
                                    'algorithmic_GBps': 4.0 * Nb * wordsb / (bk_ms * 1e-3) / 1e9,
                                    'frac_hbm': 4.0 * Nb * wordsb / (bk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    'moved_GBps': 2.0 * Nb * wordsb / (bk_ms * 1e-3) / 1e9,
                                    'valu_frac': t1_flops(Nb, D, K) / (bk_ms * 1e-3) / FP32_PEAK_FLOPS}
                del bloop, xb, rb
                torch.cuda.empty_cache()
            extra['t2_svae_vmp'] = bench_t2(N, D, K, args.s, 10, 3, dev, None, 1, cpu=not args.no_cpu_baseline)
            torch.cuda.empty_cache()
            # (five warm-up steps: one driver run of round 6 caught an allocator-growth step - 39.6 ms - as the first timed step after three)
            extra['t3_svae_train'] = bench_t3(N, D, K, args.s, args.u, 5, 5, dev, None, cpu=not args.no_cpu_baseline)
            torch.cuda.empty_cache()
            extra['t3_svae_train']['roofline'] = t3_roofline(N, K, args.s, D, args.u, dev)     # dominant kernel of T3, timed by this run
            torch.cuda.empty_cache()
            if (N, D, K) == (1_000_000, 8, 16):
                # (c) the per-GPU steps of a STRONG-scaling run, timed on this one GPU: N / 2, N / 4, N / 8 rows per rank
                extra['shard_steps'] = shard_steps(N, D, K, args.s, args.u, flav, kappa, dev, ms1={
                    't1': dt / args.steps * 1e3, 't2': extra['t2_svae_vmp']['ms_per_step'], 't3': extra['t3_svae_train']['per_step_ms']['median']},
                    exch_us={'t1_rccl': extra['t1_forced_dist_1rank']['rccl_overhead_us'], 't1_peer': extra['t1_forced_dist_1rank']['dist_overhead_us'],
                             't2': exch_us['t2']['allreduce_us_1rank'], 't3': exch_us['t3']['allreduce_us_1rank']}, exch_detail=exch_us)
                torch.cuda.empty_cache()
            # BASELINE configs[3]-sized model at the reference's minibatch size (Auto: Dy=6, L=8, K=10, U=50)
            extra['t3_minibatch64'] = bench_minibatch(64, 10, 8, 6, args.s, args.u, dev, cpu=not args.no_cpu_baseline)
            torch.cuda.empty_cache()
        if rank == 0:
            ms = dt / args.steps * 1e3
            words = (2 * D + 2 * K) if flav == L.VMP_GMM else (2 * D + 4 * K)
            alg_bytes = 4.0 * n_loc * words                  # T1 algorithmic bytes per step (this GPU's rows)
            min_bytes = alg_bytes / 2                        # what the fused pass has to move: read x, write r (u)
            achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
            traffic, traffic_src = traffic_live
            tf = os.path.join(ROOT, 'profiles', 'traffic_%s.json' % args.workload)
            if traffic is None and os.path.exists(tf) and (n_loc, D, K) == (1_000_000, 8, 16):
                tj = json.load(open(tf))
                traffic = tj.get('hbm_bytes_per_launch')
                traffic_src = 'committed rocprofv3 PMC summary (%s), not re-measured by this run [%s]' % (tj.get('source', tf), traffic_live[1])
            out = dict(common)
            out.update({
                'metric': 'vmp_step_datapoints_per_sec', 'value': n_job / (dt / args.steps), 'reps': len(walls), 'ms_per_step': ms,
                'ms_per_step_min': min(walls) / args.steps * 1e3, 'ms_per_step_max': max(walls) / args.steps * 1e3,
                'steps_per_sec': args.steps / dt,
                'config': {'workload': 'T1 %s VMP step (M-step + E-step), synthetic GMM N=%d %s, D=%d, K=%d'
                                       % (args.workload, N, 'per GPU' if args.scaling == 'weak' else 'in total (rows split over the ranks)', D, K),
                           'N_per_gpu': n_loc, 'N_job': n_job, 'D': D, 'K': K,
                           'parallelism': 'dp%d (rows sharded, 1 exchange of K-sized stats per step: %s)' % (world, chosen['exchange'] if world > 1 else 'none'),
                           'timing': 'median over `reps` timed regions of `steps` steps each'},
                # `achieved`/`frac` follow the contract: ALGORITHMIC bytes (4N(2D+2K), the un-fused M-pass +
                # E-pass) / kernel time.  The fused pass MOVES half of that (x read once, r written once): `moved_*` is the
                # physical HBM rate, `valu_frac` the fp32 arithmetic rate.
                'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                             'kernel': 'pass_xdl_kernel / pass_kernel <E-step + fused moments>', 'kernel_ms': kern_ms,
                             'algorithmic_bytes_per_launch': alg_bytes,
                             'moved_bytes_min_per_launch': min_bytes,
                             'moved_GBps': min_bytes / (kern_ms * 1e-3) / 1e9,
                             'moved_frac': min_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             'flops_per_launch': t1_flops(n_loc, D, K),
                             'valu_frac': t1_flops(n_loc, D, K) / (kern_ms * 1e-3) / FP32_PEAK_FLOPS},
            })
            if world > 1:
                out['config']['exchange_choice'] = dict(chosen, requested=args.exchange)
            if 'other_scaling' in extra:
                # both scaling modes in the line's top-level config, so that a SCALE record cannot be read as the wrong mode (SURVEY 8e
                # reads the >= 6x target at N / G rows per GPU = strong scaling)
                o = extra['other_scaling']
                vals = {args.scaling: out['value'], o['scaling']: o['value']}
                out['config']['datapoints_per_sec_by_scaling'] = vals
                out['config']['scaling_note'] = ('`value` is the %s-scaling number (%d rows per GPU); %s scaling (%d rows per GPU) measured right after it in the same '
                                                 'launch: weak %.4g, strong %.4g datapoints/s' % (args.scaling, n_loc, o['scaling'], o['rows_per_rank'], vals['weak'], vals['strong']))
            if world == 1 and not args.no_cpu_baseline:
                out['cpu_baseline'] = cpu_baseline(x_h, r0_h, args.workload)
                out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
    else:
        # ---- t2 / t3: millisecond-scale steps; `steps` capped so that the default flags finish in minutes
        n_loc, n_job = rows_for(args.scaling)
        steps = min(args.steps, 20)
        warm = min(args.warmup, 3)
        S, U = args.s, args.u
        if args.workload == 't2':
            res = bench_t2(n_loc, D, K, S, steps, warm, dev, dist, world, smm=args.smm, cpu=(world == 1 and not args.no_cpu_baseline))
            ms = res['ms_per_step']
            bwd_bytes = 4.0 * n_loc * (2.0 * K * S * D + 4 * D + 3 * K)
            roof = {'bound': 'hbm', 'achieved': res['bwd_GBps'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': res['bwd_frac_hbm'],
                    'traffic': None, 'kernel': 'svae_estep_bwd_ring_kernel (+ partial reduce)', 'kernel_ms': res['bwd_kernel_ms'],
                    'algorithmic_bytes_per_launch': bwd_bytes, 'fwd_kernel_ms': res['fwd_kernel_ms'], 'fwd_frac': res['fwd_frac_hbm']}
            metric, wl = 'svae_vmp_step_datapoints_per_sec', 'T2 %ssvae-vmp step (fused E-step fwd + bwd, sub-sampling, M-step, CVI), L=%d, K=%d, S=%d' % ('Student-t (smm) ' if args.smm else '', D, K, S)
            extra['t2'] = res
        else:
            res = bench_t3(n_loc, D, K, S, U, steps, warm, dev, None, smm=args.smm, cpu=(world == 1 and not args.no_cpu_baseline))
            ms = res['ms_per_step_wall_mean']                 # the line's own value: wall clock over the timed steps (the contract's timing)
            roof = t3_roofline(n_loc, K, S, D, U, dev)
            metric, wl = 'svae_train_step_datapoints_per_sec', 'T3 %ssvae training step (experiments.py:196-267), L=Dy=%d, K=%d, S=%d, U=%d' % ('Student-t (smm) ' if args.smm else '', D, K, S, U)
            extra['t3'] = res
        ms = max_over_ranks(dist, ms, dev)
        if rank == 0:
            out = dict(common)
            out.update({'metric': metric, 'value': n_job / (ms * 1e-3), 'steps': steps, 'warmup': warm, 'ms_per_step': ms,
                        'config': {'workload': wl + ', N=%d %s' % (N, 'per GPU' if args.scaling == 'weak' else 'in total'),
                                   'N_per_gpu': n_loc, 'N_job': n_job, 'D': D, 'K': K,
                                   'parallelism': 'dp%d (rows sharded, 1 packed all-reduce of moments + gradients per step)' % world},
                        'roofline': roof})
            if 'cpu_baseline' in res:
                out['cpu_baseline'] = res['cpu_baseline']
                out['speedup_vs_cpu_baseline'] = out['value'] / res['cpu_baseline']['value']

    if rank == 0:
        if extra:
            out['extra'] = extra
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
