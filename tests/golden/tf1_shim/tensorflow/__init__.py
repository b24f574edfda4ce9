"""Eager stand-in for the slice of the TensorFlow-1.3 API the reference's hot path touches.

TEST INFRASTRUCTURE ONLY.  TensorFlow 1.3 cannot be installed here (no py3.10 wheel, no network), so
``tests/golden/make_fixtures.py`` puts this directory first on ``sys.path`` and imports the reference's
own ``distributions/*`` and ``models/*`` from ``/root/reference`` on top of it, to produce the golden
vectors committed under ``tests/golden/*.npz``.  Nothing in the product, the oracle, ``bench.py`` or
the test-suite imports this module; it never travels to the GPU box as anything but dead files.

Semantics: every ``tf.*`` call executes immediately on torch-CPU tensors.  ``tf.float32`` maps to
torch's *default* dtype so the same reference code can be evaluated in fp32 and (for "truth") fp64.
Random ops pop pre-seeded tensors from ``INJECT`` so fixtures carry their own noise.
"""
import builtins as _b
import contextlib
import math

import numpy as _np
import torch as _t

# ----------------------------------------------------------------------------- dtypes
float32 = 'float'    # resolved to torch.get_default_dtype() at call time
float64 = 'float'
int32 = _t.int64
int64 = _t.int64
bool = _t.bool


def _dt(dtype):
    if dtype is None or dtype == 'float':
        return _t.get_default_dtype()
    return dtype


# ----------------------------------------------------------------------------- tensor type
class TensorShape(_b.tuple):
    def as_list(self):
        return list(self)

    def concatenate(self, other):
        other = (other,) if isinstance(other, int) else _b.tuple(other)
        return TensorShape(_b.tuple(self) + other)

    def __getitem__(self, i):
        r = _b.tuple.__getitem__(self, i)
        return TensorShape(r) if isinstance(i, slice) else r

    def __eq__(self, other):
        if isinstance(other, int):
            other = (other,)
        return _b.tuple(self) == _b.tuple(other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = _b.tuple.__hash__


class Tensor(_t.Tensor):
    """torch tensor with the handful of tf.Tensor attributes the reference uses.  In-place python
    operators rebind instead of mutating (tf tensors are immutable)."""

    def get_shape(self):
        return TensorShape(self.shape)

    @property
    def name(self):
        return getattr(self, '_tf_name', 'Tensor:0')

    def assign(self, value):
        return value

    def __iadd__(self, o):
        return self + o

    def __isub__(self, o):
        return self - o

    def __imul__(self, o):
        return self * o

    def __itruediv__(self, o):
        return self / o


def _wrap(x):
    return x.as_subclass(Tensor) if isinstance(x, _t.Tensor) and not isinstance(x, Tensor) else x


def _T(x, dtype=None):
    if isinstance(x, _t.Tensor):
        return _wrap(x if dtype is None else x.to(_dt(dtype)))
    if isinstance(x, _np.ndarray) and dtype is None and x.dtype.kind == 'f':
        dtype = 'float'
    if dtype is None and isinstance(x, (float, list, _b.tuple)):
        dtype = 'float' if _np.asarray(x).dtype.kind == 'f' else None
    return _wrap(_t.as_tensor(x, dtype=_dt(dtype) if dtype is not None else None))


def _named(x, name, scope=True):
    if name is not None and isinstance(x, _t.Tensor):
        x._tf_name = ('/'.join(_SCOPE + [name]) if scope else name) + ':0'
    return x


# ----------------------------------------------------------------------------- scopes / variables
_SCOPE = []
VARIABLES = {}        # full name -> Tensor (requires_grad = trainable)
INJECT = {'random_normal': [], 'random_uniform': [], 'multinomial': [], 'dirichlet': []}
GENERATOR = _t.Generator().manual_seed(0)


def reset():
    del _SCOPE[:]
    VARIABLES.clear()
    for v in INJECT.values():
        del v[:]


class _VarScope(object):
    def __init__(self, name):
        self.name = name

    def reuse_variables(self):
        pass


@contextlib.contextmanager
def name_scope(name=None, *a, **k):
    yield name


@contextlib.contextmanager
def variable_scope(name_or_scope, *a, **k):
    if isinstance(name_or_scope, _VarScope):
        yield name_or_scope
        return
    _SCOPE.append(name_or_scope)
    try:
        yield _VarScope('/'.join(_SCOPE))
    finally:
        _SCOPE.pop()


def get_variable_scope():
    return _VarScope('/'.join(_SCOPE))


@contextlib.contextmanager
def device(*a, **k):
    yield


def get_variable(name, shape=None, initializer=None, trainable=True, dtype=None, **k):
    full = '/'.join(_SCOPE + [name])
    if full in VARIABLES:
        return VARIABLES[full]
    if callable(initializer):
        val = initializer(shape)
    else:
        val = _T(initializer, dtype)
    val = _T(val).detach().clone().to(_dt(dtype)).as_subclass(Tensor)
    val.requires_grad_(builtins_bool(trainable))
    val._tf_name = full + ':0'
    VARIABLES[full] = val
    return val


def Variable(initial_value, dtype=None, name=None, trainable=False, **k):
    v = _T(initial_value, dtype).detach().clone().as_subclass(Tensor)
    return _named(v, name)


builtins_bool = _b.bool


# ----------------------------------------------------------------------------- creation
def constant(value, dtype=None, name=None, shape=None):
    if isinstance(value, _t.Tensor):
        return _T(value, dtype)
    if dtype is None:
        dtype = 'float' if _np.asarray(value).dtype.kind == 'f' else int64
    return _T(value, dtype)


def constant_initializer(v):
    return lambda shape: _t.full(_b.tuple(shape or ()), float(v))


def _shape(s):
    if isinstance(s, _t.Tensor):
        return _b.tuple(int(v) for v in s)
    if isinstance(s, int):
        return (s,)
    return _b.tuple(int(v) for v in s)


def ones(shape, dtype=None, name=None):
    return _wrap(_t.ones(_shape(shape), dtype=_dt(dtype)))


def zeros(shape, dtype=None, name=None):
    return _wrap(_t.zeros(_shape(shape), dtype=_dt(dtype)))


def ones_like(x, dtype=None, name=None):
    return _wrap(_t.ones_like(x, dtype=_dt(dtype) if dtype else None))


def zeros_like(x, dtype=None, name=None):
    return _wrap(_t.zeros_like(x, dtype=_dt(dtype) if dtype else None))


def eye(n, dtype=None, name=None):
    return _wrap(_t.eye(int(n), dtype=_dt(dtype)))


def range(*args, **k):
    dtype = k.get('dtype')
    args = [float(a) if (isinstance(a, _t.Tensor) or isinstance(a, float)) else a for a in args]
    if dtype is None:
        dtype = 'float' if any(isinstance(a, float) for a in args) else int64
    return _wrap(_t.arange(*args, dtype=_dt(dtype)))


# ----------------------------------------------------------------------------- random (injectable)
def _pop(kind, shape):
    q = INJECT[kind]
    if q:
        v = _T(q.pop(0), 'float' if kind != 'multinomial' else None)
        if shape is not None:
            assert _b.tuple(v.shape) == _b.tuple(shape), (kind, v.shape, shape)
        return v
    return None


def random_normal(shape, mean=0., stddev=1., dtype=None, seed=None, name=None):
    shape = _shape(shape)
    v = _pop('random_normal', shape)
    if v is None:
        v = _t.randn(shape, generator=GENERATOR, dtype=_t.float64).to(_dt(dtype))
    return _wrap(v * stddev + mean)


def random_uniform(shape, minval=0., maxval=1., dtype=None, seed=None, name=None):
    shape = _shape(shape)
    v = _pop('random_uniform', shape)
    if v is None:
        v = _t.rand(shape, generator=GENERATOR, dtype=_t.float64).to(_dt(dtype))
    return _wrap(v * (maxval - minval) + minval)


def random_normal_initializer(mean=0., stddev=1., dtype=None, seed=None):
    return lambda shape: random_normal(shape, mean, stddev, dtype)


def multinomial(logits, num_samples, seed=None, name=None):
    v = _pop('multinomial', (logits.shape[0], num_samples))
    if v is None:
        p = _t.softmax(logits.detach().double(), dim=-1)
        v = _t.multinomial(p, num_samples, replacement=True, generator=GENERATOR)
    return _wrap(v.to(_t.int64))


def set_random_seed(seed):
    GENERATOR.manual_seed(int(seed))


# ----------------------------------------------------------------------------- shape ops
def identity(x, name=None):
    return _named(_T(x) * 1 if False else _T(x), None)


def tuple(tensors, name=None, **k):
    return [_T(x) for x in tensors]


def group(*a, **k):
    return list(a)


def stop_gradient(x, name=None):
    return _wrap(x.detach())


def expand_dims(x, axis=None, name=None, dim=None):
    return _wrap(_T(x).unsqueeze(axis if axis is not None else dim))


def reshape(x, shape, name=None):
    return _wrap(_T(x).reshape(_shape(shape)))


def tile(x, multiples, name=None):
    return _wrap(_T(x).repeat(*[int(m) for m in multiples]))


def transpose(x, perm=None, name=None):
    return _wrap(x.permute(*perm) if perm is not None else x.t())


def concat(values, axis, name=None):
    return _wrap(_t.cat([_T(v) for v in values], dim=axis))


def split(x, num, axis=0, name=None):
    return [_wrap(c) for c in _t.chunk(x, num, dim=axis)]


def cast(x, dtype=None, name=None):
    return _T(x).to(_dt(dtype))


def to_float(x, name=None):
    return _T(x).to(_dt('float'))


def to_int32(x, name=None):
    return _T(x).to(_t.int64)


def gather_nd(params, indices, name=None):
    idx = indices.long()
    return _wrap(params[builtins_tuple(idx[..., i] for i in _b.range(idx.shape[-1]))])


builtins_tuple = _b.tuple


def argmax(x, axis=None, name=None, **k):
    return _wrap(_t.argmax(x, dim=axis))


# ----------------------------------------------------------------------------- math
def _r(fn):
    def f(x, axis=None, keep_dims=False, name=None, **k):
        x = _T(x)
        if axis is None:
            return _wrap(fn(x))
        return _wrap(fn(x, dim=axis, keepdim=keep_dims))
    return f


reduce_sum = _r(_t.sum)
reduce_mean = _r(_t.mean)


def reduce_max(x, axis=None, keep_dims=False, name=None):
    return _wrap(x.max() if axis is None else x.amax(dim=axis, keepdim=keep_dims))


def reduce_logsumexp(x, axis=None, keep_dims=False, name=None):
    return _wrap(_t.logsumexp(x, dim=axis, keepdim=keep_dims))


def _u(fn):
    return lambda x, name=None: _wrap(fn(_T(x)))


def _bin(fn):
    return lambda a, b, name=None: _named(_wrap(fn(_T(a), _T(b))), None)


log = _u(_t.log)
exp = _u(_t.exp)
log1p = _u(_t.log1p)
digamma = _u(_t.digamma)
lgamma = _u(_t.lgamma)
tanh = _u(_t.tanh)
sqrt = _u(_t.sqrt)
square = _u(_t.square)
is_nan = _u(_t.isnan)
logical_not = _u(_t.logical_not)
add = _bin(_t.add)
subtract = _bin(_t.sub)
multiply = _bin(_t.mul)
divide = _bin(_t.div)
equal = _bin(_t.eq)
pow = _bin(_t.pow)


def where(cond, x=None, y=None, name=None):
    return _wrap(_t.where(cond, _T(x), _T(y)))


def einsum(eq, *ops):
    return _wrap(_t.einsum(eq, *[_T(o) for o in ops]))


def matmul(a, b, name=None, **k):
    return _wrap(_t.matmul(_T(a), _T(b)))


def matrix_inverse(x, name=None):
    return _wrap(_t.linalg.inv(x))


def matrix_solve(a, b, name=None):
    return _wrap(_t.linalg.solve(a, b))       # partial-pivot LU, as Eigen's


def matrix_determinant(x, name=None):
    return _wrap(_t.linalg.det(x))


def matrix_transpose(x, name=None):
    return _wrap(x.transpose(-1, -2))


def matrix_diag(x, name=None):
    return _wrap(_t.diag_embed(x))


def matrix_diag_part(x, name=None):
    return _wrap(_t.diagonal(x, dim1=-2, dim2=-1))


def matrix_set_diag(x, d, name=None):
    return _wrap(x - _t.diag_embed(_t.diagonal(x, dim1=-2, dim2=-1)) + _t.diag_embed(d))


def cholesky(x, name=None):
    return _wrap(_t.linalg.cholesky(x))


def assign(ref, value, name=None):
    return _named(_T(value), None)


class Dimension(int):
    pass


# ----------------------------------------------------------------------------- sub-namespaces
class _NS(object):
    pass


nn = _NS()
nn.softplus = lambda x, name=None: _wrap(_t.logaddexp(x, _t.zeros_like(x)))
nn.softmax = lambda x, name=None, **k: _wrap(_t.softmax(x, dim=-1))
nn.sigmoid = lambda x, name=None: _wrap(_t.sigmoid(x))
sigmoid = nn.sigmoid


def _dense(inputs, units, activation=None, kernel_initializer=None, bias_initializer=None, name=None, **k):
    with variable_scope(name):
        w = get_variable('kernel', (int(inputs.shape[-1]), units), kernel_initializer)
        b = get_variable('bias', (units,), bias_initializer)
    out = _wrap(_t.matmul(inputs, w) + b)
    return activation(out) if activation is not None else out


layers = _NS()
layers.dense = _dense


class _TriL(object):
    def __init__(self, tril, name=None):
        self._m = tril

    def to_dense(self):
        return _wrap(_t.tril(self._m))


class _Dirichlet(object):
    def __init__(self, conc):
        self.conc = conc

    def sample(self, n, seed=None):
        v = _pop('dirichlet', (int(n), int(self.conc.shape[0])))
        if v is None:
            g = _t.distributions.Gamma(self.conc.double(), _t.ones_like(self.conc.double())).sample((int(n),))
            v = g / g.sum(-1, keepdim=True)
        return _wrap(v.to(_dt('float')))


class _Normal(object):
    """tf.contrib.distributions.Normal(loc, scale).sample(seed=...): one draw of the parameters' shape (vae.py:292-293)."""
    def __init__(self, loc, scale):
        self.loc, self.scale = loc, scale

    def sample(self, seed=None):
        return random_normal(_b.tuple(self.loc.shape)) * self.scale + self.loc


class _BernoulliDist(object):
    """tf.distributions.Bernoulli(probs=...).sample(seed=...) in {0,1} (losses.py:302-303); draws = injected uniforms."""
    def __init__(self, probs=None, logits=None):
        self.probs = probs

    def sample(self, seed=None):
        u = random_uniform(_b.tuple(self.probs.shape))
        return _wrap((u < self.probs).to(self.probs.dtype))


contrib = _NS()
contrib.linalg = _NS()
contrib.linalg.LinearOperatorTriL = _TriL
contrib.distributions = _NS()
contrib.distributions.Dirichlet = _Dirichlet
contrib.distributions.Normal = _Normal
distributions = _NS()
distributions.Bernoulli = _BernoulliDist

summary = _NS()
for _n in ('scalar', 'histogram', 'tensor_summary', 'image', 'merge_all', 'merge', 'FileWriter'):
    setattr(summary, _n, lambda *a, **k: None)

train = _NS()
