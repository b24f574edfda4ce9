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


def run(config, nb_iters=2000, size_minibatch=100, nb_samples=10, nb_samples_te=100, measurement_freq=500,
        path_dataset=None, device='cuda', verbose=True, ratio_tr=0.7):
    torch.manual_seed(config.get('seed', 0))
    vae.reset_variables()
    X, lab = data_mod.load_dataset(config['dataset'], path_dataset)
    X_tr, y_tr, X_te, y_te = data_mod.split_and_scale(config['dataset'], X, lab, ratio_tr=ratio_tr, seed_split=0)
    dev = torch.device(device)
    Xtr, Xte = torch.as_tensor(X_tr).to(dev), torch.as_tensor(X_te).to(dev)
    Lte = None if y_te is None else torch.as_tensor(y_te, dtype=torch.float32).to(dev)
    smm = 'smm' in config['method']
    tr = SVAETrainer(config['K'], config['L'], config['U'], X_tr.shape[1], nb_samples=nb_samples, lr=config['lr'],
                     lrcvi=config['lrcvi'], decay_rate=config.get('decay_rate', 1), seed=config.get('seed', 0),
                     device=dev, smm=smm, dof=config.get('DoF', 5))
    batches = data_mod.minibatches(X_tr, size_minibatch, seed=config.get('seed', 0))
    log_id = generate_log_id(config)
    history = []
    t0 = time.time()
    for i in range(nb_iters):
        idx = torch.as_tensor(next(batches)).to(dev)
        out = tr.step(Xtr[idx].contiguous())
        if i % measurement_freq == 0 or i == nb_iters - 1:
            m = evaluate(tr, Xte, Lte, nb_samples_te, seed=config.get('seed', 0))
            m['iter'], m['neg_normed_elbo'] = i, -float(out['elbo']) / size_minibatch      # experiments.py:318-320
            history.append(m)
            if verbose:
                print('Iteration %5d\t\t%.4fsec\t\t%.4f   %s' % (i, time.time() - t0, m['neg_normed_elbo'],
                                                                 {k: round(v, 4) for k, v in m.items() if k not in ('iter', 'neg_normed_elbo')}))
    return tr, history, log_id


if __name__ == '__main__':
    schedule = create_schedule({'dataset': 'pinwheel', 'method': 'svae-cvi', 'lr': [0.01], 'lrcvi': [0.1], 'K': 10,
                                'L': [2], 'U': 50, 'seed': 0})
    for cfg in schedule:
        run(cfg)
