"""Small host-side data utilities - the part of reference data.py a driver needs:
the pinwheel generator (data.py:216-235, after Johnson et al. 2016), train/test split + scaling
(data.py:82-120) and shuffled minibatches (the TF shuffle queue of data.py:130-171 becomes a generator).
Host I/O only; nothing here is on the hot path."""
import numpy as np


def make_pinwheel_data(radial_std, tangential_std, num_classes, num_per_class, rate):
    """Same construction and RNG call order as the reference (np.random.seed(1); one randn; one permutation), so the
    generated points are identical."""
    np.random.seed(1)
    n = num_classes * num_per_class
    base = np.random.randn(n, 2) * np.array([radial_std, tangential_std])
    base[:, 0] += 1.0
    labels = np.repeat(np.arange(num_classes), num_per_class)
    spoke = np.linspace(0.0, 2.0 * np.pi, num_classes, endpoint=False)[labels]
    ang = spoke + rate * np.exp(base[:, 0])
    c, s = np.cos(ang), np.sin(ang)
    pts = 10.0 * np.stack([base[:, 0] * c + base[:, 1] * s, -base[:, 0] * s + base[:, 1] * c], axis=1)
    shuffled = np.random.permutation(np.hstack([pts, labels[:, None]]))
    return shuffled[:, 0:2], shuffled[:, 2].astype(int)


def load_dataset(dataset, path_datadir=None):
    """(data (N,Dy) float64, labels (N,) int or None).  'pinwheel' is generated; 'auto' / 'aggregation' / 'geyser' are
    read from `path_datadir` laid out like the reference's datasets/ directory (data.py:40-70)."""
    if dataset in ('pinwheel', 'noisy-pinwheel'):
        return make_pinwheel_data(0.3, 0.05, 5, 200, 0.25)
    if path_datadir is None:
        raise ValueError("dataset '%s' needs path_datadir" % dataset)
    import pandas as pd
    if dataset == 'auto':
        raw = pd.read_csv(path_datadir + '/Auto/auto-mpg.csv', sep=',', header=None).values
        raw = raw[raw[:, 3] != '?']
        cyl = raw[:, 1].astype(int)
        labels = np.searchsorted(np.array([3, 4, 5, 6, 8]), cyl)               # cylinders {3,4,5,6,8} -> 0..4
        return raw[:, [0, 2, 3, 4, 5, 6]].astype(np.float64), labels
    if dataset == 'aggregation':
        raw = pd.read_csv(path_datadir + '/Aggregation.txt', sep='\t', header=None).values
        return raw[:, 0:2], raw[:, 2].astype(int) - 1
    if dataset == 'geyser':
        raw = pd.read_csv(path_datadir + '/geyser', sep=' ', header=None).values[:, [1, 2]]
        return raw, (raw[:, 1] > 20).astype(int)
    raise Exception("Dataset '%s' does not exist." % dataset)


def split_and_scale(dataset, data, labels, ratio_tr=0.7, seed_split=0):
    """data.py:82-120: sklearn train_test_split; 'auto' is standardised and scaled by 5, pinwheel is left as is,
    everything else is standardised.  Returns (X_tr, y_tr, X_te, y_te) with one-hot labels (or None)."""
    from sklearn.model_selection import train_test_split
    from sklearn.preprocessing import StandardScaler
    onehot = None
    if labels is not None:
        onehot = np.zeros((data.shape[0], np.unique(labels).size))
        onehot[np.arange(data.shape[0]), labels] = 1
        X_tr, X_te, y_tr, y_te = train_test_split(data, onehot, test_size=1 - ratio_tr, random_state=seed_split)
    else:
        X_tr, X_te = train_test_split(data, test_size=1 - ratio_tr, random_state=seed_split)
        y_tr = y_te = None
    if dataset not in ('pinwheel', 'noisy-pinwheel'):
        sc = StandardScaler().fit(X_tr)
        mult = 5.0 if dataset == 'auto' else 1.0
        X_tr, X_te = sc.transform(X_tr) * mult, sc.transform(X_te) * mult
    return X_tr.astype(np.float32), y_tr, X_te.astype(np.float32), y_te


def minibatches(X, size_minibatch, seed=0):
    """Endless generator of shuffled minibatches (index arrays)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    N = X.shape[0]
    while True:
        perm = rng.permutation(N)
        for i in range(0, N - size_minibatch + 1, size_minibatch):
            yield perm[i:i + size_minibatch]


def minibatches_device(X_dev, size_minibatch, seed=0):
    """As `minibatches`, but yields the minibatch ROWS gathered on the device: one host-to-device copy of the epoch's
    permutation instead of one (synchronising) index copy per iteration.  Same permutation stream as `minibatches`."""
    import torch
    rng = np.random.Generator(np.random.PCG64(seed))
    N = X_dev.shape[0]
    while True:
        perm = torch.as_tensor(rng.permutation(N)).to(X_dev.device)
        for i in range(0, N - size_minibatch + 1, size_minibatch):
            yield X_dev.index_select(0, perm[i:i + size_minibatch])


def perturb_data(x, noise_ratio=0.1, noise_mean=0, noise_stddev=10, seed=0):
    """reference data.py:238-259: a random `noise_ratio` of the rows is REPLACED by N(mean, stddev) noise (in place,
    same numpy RandomState stream)."""
    np.random.seed(seed)
    N, D = x.shape
    N_noise = int(N * noise_ratio)
    noise_indices = np.random.permutation(np.arange(N))[:N_noise]
    x[noise_indices, :] = np.random.normal(loc=noise_mean, scale=noise_stddev, size=(N_noise, D))
    return x


def make_minibatch(dataset, ratio_tr=None, ratio_val=None, binarise=False, path_datadir='../datasets', size_minibatch=128,
                   size_testbatch=-1, nb_towers=1, nb_threads=2, seed_split=0, seed_minibatch=0, dtype=None,
                   name='data_prep', noise_level=0.1, device='cuda'):
    """reference data.py:9-176 for the table datasets (pinwheel, noisy-pinwheel, auto, aggregation, geyser):
    returns (y_tr, lbl_tr, y_te, lbl_te) like the reference, where y_tr is an endless generator of shuffled minibatches
    gathered on the device (the reference's tf.train.shuffle_batch queue; split over `nb_towers` = ranks is done by the
    launcher) and y_te / labels are device tensors.  The MNIST-style TFRecord inputs (data.py:13-33,179-213) are not
    built: there are no such files in this image."""
    import torch
    if dataset in ('mnist', 'mnist-small', 'fashion') or binarise:
        raise NotImplementedError('TFRecord image datasets are not built')
    data, labels = load_dataset(dataset, path_datadir)
    if dataset == 'noisy-pinwheel':
        data = perturb_data(np.array(data, dtype=np.float64), noise_ratio=noise_level)
    X_tr, l_tr, X_te, l_te = split_and_scale(dataset, data, labels, ratio_tr=0.7 if ratio_tr is None else ratio_tr,
                                             seed_split=seed_split)
    dev = torch.device(device)
    Xtr, Xte = torch.as_tensor(X_tr).to(dev), torch.as_tensor(X_te).to(dev)
    if size_testbatch and size_testbatch > 0:
        Xte, l_te = Xte[:size_testbatch], (None if l_te is None else l_te[:size_testbatch])
    to_t = lambda a: None if a is None else torch.as_tensor(a, dtype=torch.float32).to(dev)
    return minibatches_device(Xtr, size_minibatch, seed=seed_minibatch), to_t(l_tr), Xte, to_t(l_te)
