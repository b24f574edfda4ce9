"""Mirror of the hot-path part of reference helpers/tf_utils.py (logdet :25-49, average_gradients :52-87)."""
import torch

from .. import _klinalg


def logdet(A, name='logdet'):
    """log det of SPD matrices (..., D, D) = 2 * sum log diag chol(A)  (reference tf_utils.py:25-49).
    K-sized use only; the per-(n,k) log-determinants of the hot path live inside the HIP kernels."""
    return 2.0 * _klinalg.cholesky(A).diagonal(dim1=-2, dim2=-1).log().sum(-1)


def average_gradients(tower_grads):
    """reference tf_utils.py:52-87: list over towers of [(grad, var), ...] -> [(mean grad, var), ...].
    In the one-process-per-GPU design the towers are ranks and this mean is one packed RCCL all-reduce
    (vmp_for_svae_amd.training.SVAETrainer.step); this in-process form is kept for API parity and tests."""
    out = []
    for gv in zip(*tower_grads):
        g = torch.stack([g_ for g_, _ in gv], dim=0).mean(dim=0)
        out.append((g, gv[0][1]))
    return out


def variable_on_device(name, shape, initializer, trainable=True, dtype=torch.float32, device='cuda'):
    """reference tf_utils.py:8-22: a named variable living on `device` (get-or-create in the vae variable store)."""
    from ..models import vae

    def init():
        v = initializer(shape) if callable(initializer) else torch.as_tensor(initializer)
        return v.to(device=device, dtype=dtype)
    return vae._get_variable(name, init, trainable)
