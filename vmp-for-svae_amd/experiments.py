"""Driver in the shape of reference experiments.py (config -> init -> training loop with periodic evaluation) on top
of training.SVAETrainer.  SURVEY 8f rank 3; thin by design (no TensorBoard, no plotting).

    python -m vmp_for_svae_amd.experiments            # pinwheel, svae-cvi, as the reference's commented config
"""
import time

import numpy as np
import torch

from . import data as data_mod
from . import losses
from .helpers.logging_utils import generate_log_id
from .helpers.scheduling import create_schedule
from .models import svae, vae
from .training import SVAETrainer


def evaluate(tr, y, labels, nb_samples, seed=0):
    """experiments.py:270-304: inference on the evaluation set with nb_samples_te samples, then the metrics."""
    with torch.no_grad():
        y_rec, _, x_k, x_s, log_z, _, _ = svae.inference(y, tr.phi_gmm, tr.encoder_layers, tr.decoder_layers, nb_samples,
                                                         stddev_init_nn=tr.stddev_init_nn, seed=seed)
        out = {'mse': float(losses.weighted_mse(y, y_rec[0], torch.exp(log_z))),
               'loli': float(losses.diagonal_gaussian_logprob(y, y_rec[0], y_rec[1], log_z))}
        if labels is not None:
            e, p = losses.purity(torch.exp(log_z), labels)
            out['entropy'], out['purity'] = float(e), float(p)
    return out


def evaluate_imputation(tr, y, missing_data_mask, nb_samples_pert=20, nb_samples_te=100, seed=0):
    """experiments.py:361-377: missing entries are replaced by noise nb_samples_pert times, each perturbed set is pushed
    through svae.inference with nb_samples_te samples, and the imputations are scored on the missing entries."""
    def impute(y_perturbed):                                          # experiments.py:365-372
        (y_k_mean, out2), _, _, _, log_r_nk, _, _ = svae.inference(
            y_perturbed.contiguous(), tr.phi_gmm, tr.encoder_layers, tr.decoder_layers, nb_samples_te,
            stddev_init_nn=tr.stddev_init_nn, seed=seed)
        return y_k_mean, out2, log_r_nk
    with torch.no_grad():
        mse, lopr = losses.imputation_losses(y, missing_data_mask, impute, nb_samples_pert, nb_samples_te, seed=seed)
    return {'imp_mse': float(mse), 'imp_logprob': float(lopr)}


THETA_NAMES = ('alpha_k', 'A', 'b', 'beta_k', 'v_hat')               # natural parameters, svae.py:433-458


def checkpoint_state(tr):
    """Everything a run needs to resume (the reference's tf.train.Saver covers the same variables, experiments.py:357,
    433-440): MLP weights, phi_gmm, theta, Adam slots, step counters - keyed by the reference's variable names."""
    st = {}
    names, params = tr.trainables()
    for n, p in zip(names, params):
        st[n] = p.detach().cpu().numpy()
    if tr.smm:
        st['theta/alpha_k'] = tr.theta[0].detach().cpu().numpy()
        st['theta/DoF'] = tr.theta[3].detach().cpu().numpy()
    else:
        for n, t in zip(THETA_NAMES, tr.theta):
            st['theta/' + n] = t.detach().cpu().numpy()
    st['global_step'] = np.array(tr.global_step)
    if tr.opt is not None:
        st['adam/t'] = np.array(tr.opt.t)
        for n, m, v in zip(names, tr.opt.m, tr.opt.v):
            st['adam/m/' + n] = m.cpu().numpy()
            st['adam/v/' + n] = v.cpu().numpy()
    return st


def save_checkpoint(tr, path):
    np.savez(path, **checkpoint_state(tr))
    return path


def load_checkpoint(tr, path):
    """Restore a trainer created with the same configuration from save_checkpoint's .npz."""
    from .training import TFAdam
    z = np.load(path)
    # make sure the MLP variables exist (they are created lazily by the first forward pass)
    dev = tr.device
    if not vae.net_variables('decoder_net'):
        Dy = tr.decoder_layers[-1][0]
        vae.make_encoder(torch.zeros(1, Dy, device=dev), tr.encoder_layers, tr.stddev_init_nn, seed=tr.seed)
        vae.decoder_variables(tr.L, tr.decoder_layers, tr.stddev_init_nn, tr.seed, dev)
    names, params = tr.trainables()
    with torch.no_grad():
        for n, p in zip(names, params):
            p.copy_(torch.as_tensor(z[n]).to(dev))
        if tr.smm:
            tr.theta[0].copy_(torch.as_tensor(z['theta/alpha_k']).to(dev))
            tr.theta[3].copy_(torch.as_tensor(z['theta/DoF']).to(dev))
        else:
            for n, t in zip(THETA_NAMES, tr.theta):
                t.copy_(torch.as_tensor(z['theta/' + n]).to(dev))
    tr.global_step = int(z['global_step'])
    if 'adam/t' in z.files:
        tr.opt = TFAdam(params, tr.lr)
        tr.opt.t = int(z['adam/t'])
        for n, m, v in zip(names, tr.opt.m, tr.opt.v):
            m.copy_(torch.as_tensor(z['adam/m/' + n]).to(dev))
            v.copy_(torch.as_tensor(z['adam/v/' + n]).to(dev))
    return tr


def run(config, nb_iters=2000, size_minibatch=100, nb_samples=10, nb_samples_te=100, measurement_freq=500,
        path_dataset=None, device='cuda', verbose=True, ratio_tr=0.7, imputation_freq=None, nb_samples_pert=20,
        ratio_missing_data=0.1, checkpoint_freq=None, checkpoint_dir=None, graph=True, group=None, steps_per_replay=1):
    """One run of the reference driver (experiments.py:86-457).  Under torch.distributed (one process per GPU) every
    rank draws the same shuffled minibatch stream and trains on its tower_slice of each minibatch - the reference's
    tf.split over towers (data.py:174-175, experiments.py:196-244); SVAETrainer.step sums moments / ELBO and AVERAGES
    gradients over ranks (helpers/tf_utils.py:52-87), i.e. the run equals the reference's nb_gpu = world run; a
    single-process run on the whole minibatch SUMS the gradient over all rows instead (world x larger gradient).
    Rank 0 alone prints and writes checkpoints (every rank holds identical parameters); metrics are evaluated by all
    ranks, identically.
    graph: True = capture the fixed-size training step once as a HIP graph and replay it - single-process runs only; a
    data-parallel run (several ranks) steps eagerly unless graph='dp' asks for the two-graphs-around-the-collective form
    (training.GraphedSVAEStep, round 5).  It is opt-in because of a finding that is not ours to fix: after HIP-graph replays in a
    process that shares its GPU with another rank, torch's device Cholesky has returned a wrong factor on its first call
    (tools/r6_dpg_repro.py, profiles/r06_dpg_linalg.txt); this package keeps every K-sized factorisation on the host
    (_klinalg), but a caller's own GPU linalg after such a run is exposed.
    steps_per_replay = n > 1 (single-process graph runs whose step is the trainer's direct kernel sequence): up to n consecutive
    iterations that no measurement / imputation / checkpoint iteration interrupts run from ONE graph replay
    (GraphedSVAEStep(steps_per_replay=n)); the same steps on the same minibatches, bit for bit - only the launches are batched."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    torch.manual_seed(config.get('seed', 0))
    vae.reset_variables()
    X, lab = data_mod.load_dataset(config['dataset'], path_dataset)
    X_tr, y_tr, X_te, y_te = data_mod.split_and_scale(config['dataset'], X, lab, ratio_tr=ratio_tr, seed_split=0,
                                                      noise_level=config.get('noise_level', 0.1))
    dev = torch.device(device)
    Xtr, Xte = torch.as_tensor(X_tr).to(dev), torch.as_tensor(X_te).to(dev)
    Lte = None if y_te is None else torch.as_tensor(y_te, dtype=torch.float32).to(dev)
    smm = 'smm' in config['method']
    tr = SVAETrainer(config['K'], config['L'], config['U'], X_tr.shape[1], nb_samples=nb_samples, lr=config['lr'],
                     lrcvi=config['lrcvi'], decay_rate=config.get('decay_rate', 1), seed=config.get('seed', 0),
                     device=dev, smm=smm, dof=config.get('DoF', 5), group=group)
    batches = data_mod.minibatches_device(Xtr, size_minibatch, seed=config.get('seed', 0), rank=rank, world=world)
    log_id = generate_log_id(config)
    missing_data_mask = losses.generate_missing_data_mask(Xte, ratio_missing_data, seed=config.get('seed', 0))
    history = []
    t0 = time.time()
    # fixed-size minibatches: capture the training step once as a HIP graph and replay it (training.GraphedSVAEStep)
    stepper = multi = None
    nrep = max(1, int(steps_per_replay))

    def observed(j):                                       # iterations after which the loop looks at the trainer or the step's output
        return (j % measurement_freq == 0 or j == nb_iters - 1 or
                bool(checkpoint_freq and checkpoint_dir and (j % checkpoint_freq == 0)))
    pending = []                                           # outputs of the iterations a multi-step replay has already run
    for i in range(nb_iters):
        if pending:
            out = pending.pop(0)
        else:
            yb = next(batches)
            if graph and dev.type == 'cuda' and (world == 1 or graph == 'dp'):
                if stepper is None:
                    from .training import GraphedSVAEStep
                    stepper = GraphedSVAEStep(tr, yb)
                    if nrep > 1 and world == 1 and stepper.table_mode:
                        multi = GraphedSVAEStep(tr, yb, steps_per_replay=nrep)
                group_ok = (multi is not None and yb.shape[0] == stepper.y.shape[0] and i + nrep <= nb_iters and
                            not any(observed(j) for j in range(i, i + nrep - 1)))
                if group_ok:
                    ybs = [yb] + [next(batches) for _ in range(nrep - 1)]
                    if all(b.shape[0] == yb.shape[0] for b in ybs):
                        outs = multi(torch.stack(ybs))
                        # (the replay's output tensors are the graph's static outputs: what a later iteration reads is copied now)
                        pending = [dict(o, elbo=o['elbo'].clone()) for o in outs[1:]]
                        out = outs[0]
                    else:                                  # a ragged batch inside the group: these iterations one by one
                        res = [stepper(b) if b.shape[0] == stepper.y.shape[0] else tr.step(b) for b in ybs]
                        pending = [dict(o, elbo=o['elbo'].clone()) for o in res[1:]]
                        out = res[0]
                else:
                    out = stepper(yb) if yb.shape[0] == stepper.y.shape[0] else tr.step(yb)
            else:
                out = tr.step(yb)
        if i % measurement_freq == 0 or i == nb_iters - 1:
            m = evaluate(tr, Xte, Lte, nb_samples_te, seed=config.get('seed', 0))
            m['iter'], m['neg_normed_elbo'] = i, -float(out['elbo']) / size_minibatch      # experiments.py:318-320
            if imputation_freq and (i % imputation_freq == 0 or i == nb_iters - 1):       # experiments.py:446-452
                m.update(evaluate_imputation(tr, Xte, missing_data_mask, nb_samples_pert, nb_samples_te,
                                             seed=config.get('seed', 0)))
            history.append(m)
            if verbose and rank == 0:
                print('Iteration %5d\t\t%.4fsec\t\t%.4f   %s' % (i, time.time() - t0, m['neg_normed_elbo'],
                                                                 {k: round(v, 4) for k, v in m.items() if k not in ('iter', 'neg_normed_elbo')}))
        if checkpoint_freq and checkpoint_dir and rank == 0 and (i % checkpoint_freq == 0 or i == nb_iters - 1):
            import os
            os.makedirs(checkpoint_dir, exist_ok=True)
            save_checkpoint(tr, os.path.join(checkpoint_dir, '%s_iter%d.npz' % (log_id, i)))   # experiments.py:433-440
    return tr, history, log_id


if __name__ == '__main__':
    schedule = create_schedule({'dataset': 'pinwheel', 'method': 'svae-cvi', 'lr': [0.01], 'lrcvi': [0.1], 'K': 10,
                                'L': [2], 'U': 50, 'seed': 0})
    for cfg in schedule:
        run(cfg)
