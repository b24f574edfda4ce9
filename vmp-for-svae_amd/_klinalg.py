"""K-sized factorisations on the HOST.

The reference factorises K-sized (K, L, L) parameter matrices with TensorFlow ops (tf.cholesky / tf.matrix_inverse /
tf.matrix_determinant: distributions/niw.py:8-43, distributions/gaussian.py:8-27, models/gmm.py:117-131, models/svae.py:70-92,
474-485).  None of them is on the hot path here - every N-sized factorisation runs inside the HIP kernels, and the training step's
K-sized maps are the single-launch kernels of csrc/vmp_prep.hip - so the few that remain (model construction, the stand-alone API
functions, evaluation) run through LAPACK on the host and return to the tensor's device.

Why not torch's GPU solvers: round 5 found, and round 6 pinned down (tools/r6_dpg_repro.py, profiles/r06_dpg_linalg.txt), that the
FIRST torch.linalg.cholesky on the device after HIP-graph replays returns a wrong factor now and then when two processes share one
GPU - inputs bit-equal to a host-built copy, only the solver's output wrong, also after a full device synchronisation, with no
graph memory involved (separate pools, graphs kept alive or destroyed: no difference).  K-sized work gains nothing from the GPU.

Differentiable (the device <-> host copies are autograd-aware).  Batches larger than HOST_MAX_ELEMS (API-parity callers passing
N-sized batches) and calls made while a stream is being captured stay on the device."""
import torch

HOST_MAX_ELEMS = 1 << 16


def _on_host(t):
    if not t.is_cuda:
        return False
    if t.numel() > HOST_MAX_ELEMS:
        return False
    return not torch.cuda.is_current_stream_capturing()


def _apply(fn, t, *more):
    if not _on_host(t):
        return fn(t, *more)
    out = fn(t.cpu(), *[m.cpu() if torch.is_tensor(m) else m for m in more])
    if isinstance(out, tuple):
        return tuple(o.to(t.device) for o in out)
    return out.to(t.device)


def cholesky(A):
    return _apply(torch.linalg.cholesky, A)


def inv(A):
    return _apply(torch.linalg.inv, A)


def cholesky_inverse_spd(A):
    """inverse of a symmetric positive definite batch through its Cholesky factor"""
    return _apply(lambda a: torch.cholesky_inverse(torch.linalg.cholesky(a)), A)


def slogdet(A):
    return _apply(lambda a: tuple(torch.linalg.slogdet(a)), A)
