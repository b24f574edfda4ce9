"""vmp-for-svae_amd - MI355X-native VMP hot path of emtiyaz/vmp-for-svae behind the reference's
own Python function surface.

    from vmp_for_svae_amd.distributions import gaussian, niw, dirichlet, student_t
    from vmp_for_svae_amd.models import gmm, smm, svae, vae

Same module / function names, argument orders, shapes and return tuples as the reference's
``distributions/*.py`` and ``models/*.py``, executed eagerly on torch (ROCm) tensors.  Every N-sized
operation runs in hand-written HIP kernels for gfx950 through the C ABI of ``lib/libvmp_hip.so``
(``include/vmp_hip.h``); there is NO CPU fallback - calling an N-sized op without the library or with
non-GPU tensors raises.  K-sized parameter algebra (a handful of KxLxL matrices) is plain torch.

The directory name contains '-' so it is not importable by name; the top-level shim module
``vmp_for_svae_amd.py`` loads it under the importable name ``vmp_for_svae_amd``.
"""
from . import _lib  # noqa: F401
from . import distributions, helpers, models  # noqa: F401
from . import losses, training  # noqa: F401

__all__ = ['distributions', 'helpers', 'models', 'losses', 'training', '_lib']
