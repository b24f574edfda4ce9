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
    The MNIST-style TFRecord inputs (data.py:13-33,179-213) are read by read_from_tfrec_file below when the files exist (there are
    none in this image: FileNotFoundError)."""
    import torch
    if dataset in ('mnist', 'mnist-small', 'fashion'):
        # data.py:13-33: <datadir>/<dataset>_new/{train, test | validation}.tfrecords, 784 pixels, one-hot labels of 10 classes;
        # the split is the files' own (ratio_tr is ignored, as in the reference)
        import os
        base = os.path.join(path_datadir, dataset + '_new')
        f_tr = os.path.join(base, 'train.tfrecords')
        f_te = os.path.join(base, 'test.tfrecords' if ratio_val is None else 'validation.tfrecords')
        for f in (f_tr, f_te):
            if not os.path.exists(f):
                raise FileNotFoundError("'%s' not found: the image data sets are not shipped with this repository" % f)
        X_tr, y_tr_i = read_from_tfrec_file(f_tr, 784, binarise=binarise, seed=seed_split)
        X_te, y_te_i = read_from_tfrec_file(f_te, 784, binarise=binarise, seed=seed_split)
        l_tr, l_te = np.eye(10, dtype=np.float32)[y_tr_i], np.eye(10, dtype=np.float32)[y_te_i]
    else:
        if binarise:
            raise NotImplementedError                            # data.py:36-37
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


# ---------------------------------------------------------------------------------------------------------
# TFRecord image files (reference data.py:179-213: mnist / mnist-small / fashion).  No TensorFlow here: the record framing
# (u64 length | u32 masked crc | payload | u32 masked crc) and the tf.train.Example wire format are read directly.
# ---------------------------------------------------------------------------------------------------------
def _pb_fields(buf):
    """(field number, wire type, value) of one protobuf message: varint -> int, length-delimited -> bytes"""
    i, n = 0, len(buf)
    while i < n:
        tag, sh = 0, 0
        while True:
            c = buf[i]; i += 1
            tag |= (c & 0x7F) << sh; sh += 7
            if c < 0x80:
                break
        fn, wt = tag >> 3, tag & 7
        if wt == 0:
            v, sh = 0, 0
            while True:
                c = buf[i]; i += 1
                v |= (c & 0x7F) << sh; sh += 7
                if c < 0x80:
                    break
        elif wt == 2:
            ln, sh = 0, 0
            while True:
                c = buf[i]; i += 1
                ln |= (c & 0x7F) << sh; sh += 7
                if c < 0x80:
                    break
            v = bytes(buf[i:i + ln]); i += ln
        elif wt == 1:
            v = bytes(buf[i:i + 8]); i += 8
        elif wt == 5:
            v = bytes(buf[i:i + 4]); i += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield fn, wt, v


def _parse_example(payload):
    """tf.train.Example -> {name: bytes | list of ints}: Example.features(1) . Features.feature(1: map entry key(1), value(2)) .
    Feature.{bytes_list(1), int64_list(3)} . value(1)"""
    out = {}
    for fn, _, feats in _pb_fields(payload):
        if fn != 1:
            continue
        for fn2, _, entry in _pb_fields(feats):
            if fn2 != 1:
                continue
            key, feat = None, None
            for fn3, _, v in _pb_fields(entry):
                if fn3 == 1:
                    key = v.decode()
                elif fn3 == 2:
                    feat = v
            if key is None or feat is None:
                continue
            for fn4, _, lst in _pb_fields(feat):
                if fn4 == 1:                                   # BytesList
                    out[key] = [v for f5, _, v in _pb_fields(lst) if f5 == 1][0]
                elif fn4 == 3:                                 # Int64List (packed or not)
                    vals = []
                    for f5, wt5, v in _pb_fields(lst):
                        if f5 != 1:
                            continue
                        if wt5 == 0:
                            vals.append(v)
                        else:
                            vals += _packed_varints(v)
                    out[key] = vals
    return out


def _packed_varints(b):
    vals, i = [], 0
    while i < len(b):
        v, sh = 0, 0
        while True:
            c = b[i]; i += 1
            v |= (c & 0x7F) << sh; sh += 7
            if c < 0x80:
                break
        vals.append(v)
    return vals


def read_from_tfrec_file(filename_q, D, binarise=False, seed=0):
    """reference data.py:179-213 without TensorFlow: every record of the TFRecord file(s) `filename_q` (a path or a list of paths -
    the reference passes a filename queue) is a tf.train.Example with 'image_raw' (D uint8 bytes) and 'label' (int64).
    Returns (images (M, D) float32, labels (M,) int64): pixel values / 255 (data.py:196), or - binarise=True - stochastically
    binarised to {-1, +1} with the intensity as the probability of +1 (data.py:199-206; numpy Generator(seed) replaces
    tf.multinomial's stream; seed defaults to 0 as the reference's does, whose make_minibatch passes none).
    DELIBERATE DIVERGENCE (out-of-scope pipeline): the reference binarises DYNAMICALLY - a fresh tf.multinomial
    draw every time a record is dequeued - whereas this reader binarises the whole set ONCE: every epoch sees the same binary
    image per example.  A caller that needs the reference's regulariser reads with binarise=False and draws
    `np.where(rng.random(batch.shape) < batch, 1, -1)` per minibatch."""
    import struct
    paths = [filename_q] if isinstance(filename_q, (str, bytes)) else list(filename_q)
    imgs, labels = [], []
    for path in paths:
        with open(path, 'rb') as f:
            while True:
                head = f.read(12)
                if len(head) < 12:
                    break
                ln, = struct.unpack('<Q', head[:8])
                payload = f.read(ln)
                f.read(4)                                      # crc of the payload (not verified)
                ex = _parse_example(payload)
                img = np.frombuffer(ex['image_raw'], dtype=np.uint8)
                if img.size != D:
                    raise ValueError('%s: record with %d bytes, expected D=%d' % (path, img.size, D))
                imgs.append(img)
                labels.append(int(ex['label'][0]))
    X = np.stack(imgs).astype(np.float32) * np.float32(1.0 / 255) if imgs else np.zeros((0, D), np.float32)
    if binarise:
        rng = np.random.Generator(np.random.PCG64(seed))
        X = np.where(rng.random(X.shape) < X, 1.0, -1.0).astype(np.float32)
    return X, np.asarray(labels, dtype=np.int64)
