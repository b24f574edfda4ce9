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


def split_and_scale(dataset, data, labels, ratio_tr=0.7, seed_split=0, noise_level=0.1, ratio_val=None):
    """data.py:82-120: sklearn train_test_split; 'noisy-pinwheel' perturbs the TRAINING rows only, after the split, with
    seed=seed_split (data.py:108-109); 'auto' is standardised and scaled by 5, pinwheel is left as is, everything else is
    standardised.  ratio_val (data.py:91-105): the training part is split once more with test_size = ratio_val / (ratio_tr +
    ratio_val), and - as in the reference, which re-binds X_te / y_te there - the VALIDATION rows are what comes back as the
    "test" outputs (the first split's test rows are only counted).  Returns (X_tr, y_tr, X_te, y_te) with one-hot labels (or None)."""
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
    if ratio_val is not None:
        if not ratio_tr > ratio_val:
            raise AssertionError('ratio_tr > ratio_val (data.py:93)')
        ratio_val_tr = ratio_val / (ratio_tr + ratio_val)
        if labels is None:
            X_tr, X_te = train_test_split(X_tr, test_size=ratio_val_tr, random_state=seed_split)
        else:
            X_tr, X_te, y_tr, y_te = train_test_split(X_tr, y_tr, test_size=ratio_val_tr, random_state=seed_split)
    if dataset == 'noisy-pinwheel':
        X_tr = perturb_data(np.array(X_tr, dtype=np.float64), noise_ratio=noise_level, noise_mean=0, noise_stddev=10,
                            seed=seed_split)
    if dataset not in ('pinwheel', 'noisy-pinwheel'):
        sc = StandardScaler().fit(X_tr)
        mult = 5.0 if dataset == 'auto' else 1.0
        X_tr, X_te = sc.transform(X_tr) * mult, sc.transform(X_te) * mult
    return X_tr.astype(np.float32), y_tr, X_te.astype(np.float32), y_te


def _epochs(N, size_minibatch, seed):
    """Endless stream of index arrays: a fresh permutation per epoch, cut into whole minibatches."""
    if not 0 < size_minibatch <= N:
        raise ValueError('size_minibatch=%d must be in 1..N=%d' % (size_minibatch, N))
    rng = np.random.Generator(np.random.PCG64(seed))
    while True:
        perm = rng.permutation(N)
        for i in range(0, N - size_minibatch + 1, size_minibatch):
            yield perm[i:i + size_minibatch]


def tower_slice(size_minibatch, rank, world):
    """Rows of a minibatch that tower / rank `rank` of `world` owns: the contiguous equal split of
    tf.split(m_batch, nb_towers, axis=0) (data.py:174-175)."""
    if size_minibatch % world:
        raise ValueError('size_minibatch=%d is not divisible by the %d towers' % (size_minibatch, world))
    per = size_minibatch // world
    return slice(rank * per, (rank + 1) * per)


def minibatches(X, size_minibatch, seed=0):
    """Endless generator of shuffled minibatches (index arrays)."""
    return _epochs(X.shape[0], size_minibatch, seed)


def minibatches_device(X_dev, size_minibatch, seed=0, rank=0, world=1):
    """As `minibatches`, but yields the minibatch ROWS gathered on the device: one host-to-device copy of the epoch's
    permutation instead of one (synchronising) index copy per iteration.  Same permutation stream as `minibatches`.
    With world > 1 every rank draws the SAME stream and keeps its tower_slice of each minibatch."""
    import torch
    N = X_dev.shape[0]
    if not 0 < size_minibatch <= N:
        raise ValueError('size_minibatch=%d must be in 1..N=%d' % (size_minibatch, N))
    sl = tower_slice(size_minibatch, rank, world)
    rng = np.random.Generator(np.random.PCG64(seed))
    while True:
        perm = torch.as_tensor(rng.permutation(N)).to(X_dev.device)
        for i in range(0, N - size_minibatch + 1, size_minibatch):
            yield X_dev.index_select(0, perm[i:i + size_minibatch][sl])


def perturb_data(x, noise_ratio=0.1, noise_mean=0, noise_stddev=10, seed=0):
    """reference data.py:238-259: int(N * noise_ratio) randomly chosen rows of `x` are overwritten, in place, by
    N(noise_mean, noise_stddev) draws.  The draws come from numpy's legacy global stream seeded with `seed` (one
    permutation of the row indices, then one normal block), which is what makes the perturbed set identical to the
    reference's."""
    n_rows, n_cols = x.shape
    n_replaced = int(n_rows * noise_ratio)
    np.random.seed(seed)
    victims = np.random.permutation(np.arange(n_rows))[:n_replaced]
    x[victims] = np.random.normal(noise_mean, noise_stddev, (n_replaced, n_cols))
    return x


def make_minibatch(dataset, ratio_tr=None, ratio_val=None, binarise=False, path_datadir='../datasets', size_minibatch=128,
                   size_testbatch=-1, nb_towers=1, nb_threads=2, seed_split=0, seed_minibatch=0, dtype=None,
                   name='data_prep', noise_level=0.1, device='cuda', rank=None):
    """reference data.py:9-176 for the table datasets (pinwheel, noisy-pinwheel, auto, aggregation, geyser):
    returns (y_tr, lbl_tr, y_te, lbl_te) like the reference (with ratio_val: the validation rows as y_te / lbl_te, data.py:91-105).
      size_minibatch > 0: y_tr and lbl_tr are endless generators over the SAME shuffled stream (the reference's
        tf.train.shuffle_batch([X_tr, y_tr]) queue, data.py:130-150): the i-th next(lbl_tr) holds the labels of the rows of
        the i-th next(y_tr);
      size_minibatch <= 0: the full training tensors (data.py:151-153), as gmm.py:316 / smm.py:285 / vae.py:363 use it.
    nb_towers > 1 (data.py:174-175): each minibatch is the list of its nb_towers contiguous splits; with `rank` given
    (one process per GPU) only that tower's split is produced.
    The MNIST-style TFRecord inputs (data.py:13-33,179-213) are not built: there are no such files in this image."""
    import torch
    if dataset in ('mnist', 'mnist-small', 'fashion') or binarise:
        raise NotImplementedError('TFRecord image datasets are not built')
    data, labels = load_dataset(dataset, path_datadir)
    X_tr, l_tr, X_te, l_te = split_and_scale(dataset, data, labels, ratio_tr=0.7 if ratio_tr is None else ratio_tr,
                                             seed_split=seed_split, noise_level=noise_level, ratio_val=ratio_val)
    dev = torch.device(device)
    to_t = lambda a: None if a is None else torch.as_tensor(a, dtype=torch.float32).to(dev)
    Xtr, Xte, Ltr, Lte = to_t(X_tr), to_t(X_te), to_t(l_tr), to_t(l_te)

    def stream(T, size, seed):
        def pick(rows):
            if nb_towers <= 1:
                return rows
            parts = [rows[tower_slice(size, t, nb_towers)] for t in range(nb_towers)]
            return parts if rank is None else parts[rank]
        for idx in _epochs(T.shape[0], size, seed):
            yield pick(T.index_select(0, torch.as_tensor(idx).to(dev)))

    if size_minibatch > 0:
        y_tr = stream(Xtr, size_minibatch, seed_minibatch)
        lbl_tr = None if Ltr is None else stream(Ltr, size_minibatch, seed_minibatch)
    else:
        y_tr, lbl_tr = Xtr, Ltr
    if size_testbatch and size_testbatch > 0:
        y_te = stream(Xte, size_testbatch, seed_minibatch)
        lbl_te = None if Lte is None else stream(Lte, size_testbatch, seed_minibatch)
    else:
        y_te, lbl_te = Xte, Lte
    return y_tr, lbl_tr, y_te, lbl_te
