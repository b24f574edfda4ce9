"""ctypes binding of lib/libvmp_hip.so (C ABI: include/vmp_hip.h).  No fallback: if the library is
missing or a tensor is not a contiguous fp32 GPU tensor, the call raises."""
import ctypes
import threading
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VMP_LIB_PATH') or os.path.join(_HERE, 'lib', 'libvmp_hip.so')   # override: A/B builds

VMP_GMM, VMP_SMM = 0, 1
MAX_D, MAX_K = 8, 64

_lib = None

_c = ctypes
_P = _c.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    'vmp_abi_version': (_c.c_int, []),
    'vmp_last_error': (_c.c_char_p, []),
    'vmp_mix_pack_words': (_c.c_int, [_c.c_int]),
    'vmp_mix_stats_words': (_c.c_int, [_c.c_int]),
    'vmp_mix_workspace_bytes': (_c.c_size_t, [_c.c_int64, _c.c_int, _c.c_int]),
    'vmp_mix_stats': (_c.c_int, [_P, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _P, _P, _c.c_size_t, _P]),
    'vmp_mix_pivot': (_c.c_int, [_P, _c.c_int64, _c.c_int, _P, _P]),
    'vmp_mix_finalize': (_c.c_int, [_P, _c.c_int, _c.c_int, _c.c_int] + [_P] * 6 + [_P] * 9 + [_P]),
    'vmp_mix_pack_from_params': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int] + [_P] * 6 + [_P, _P, _P]),
    'vmp_mix_estep': (_c.c_int, [_P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P, _P, _P, _P, _P,
                                 _c.c_size_t, _P]),
    'vmp_mix_estep_fused': (_c.c_int, [_P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P, _P, _P, _c.c_size_t, _P]),
    'vmp_mix_finalize_ws64': (_c.c_int, [_P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int] + [_P] * 6 + [_P] * 10 + [_P, _P]),
    'vmp_mix_stats_ws_accurate': (_c.c_int, [_P, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _P, _c.c_size_t, _P]),
    'vmp_mix_estep_accurate': (_c.c_int, [_P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P, _P]),
    'vmp_mix_stats_ws': (_c.c_int, [_P, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _P, _c.c_size_t, _P]),
    'vmp_mix_iterate': (_c.c_int, [_P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int] + [_P] * 7 + [_P] * 11 + [_P, _c.c_size_t, _c.c_int, _P]),
    'vmp_svae_estep_fwd': (_c.c_int, [_P] * 10 + [_c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P]),
    'vmp_svae_rng_in_kernel': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int]),
    'vmp_svae_philox_noise': (_c.c_int, [_c.c_uint64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P]),
    'vmp_svae_estep_fwd_rng': (_c.c_int, [_P] * 5 + [_c.c_uint64] + [_P] * 4 + [_c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P, _P]),
    'vmp_svae_philox_noise_dev': (_c.c_int, [_P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P]),
    'vmp_svae_estep_fwd_rng_dev': (_c.c_int, [_P] * 10 + [_c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P]),
    'vmp_svae_fwd_mom_blocks': (_c.c_int, [_c.c_int64, _c.c_int, _c.c_int, _c.c_int]),
    'vmp_svae_estep_fwd_rng_epi': (_c.c_int, [_P] * 5 + [_c.c_uint64] + [_P] * 5 + [_c.c_int64, _c.c_int, _c.c_int, _c.c_int] + [_P] * 6 + [_c.c_size_t, _P]),
    'vmp_svae_mom_cvi': (_c.c_int, [_P, _c.c_int] + [_P] * 16 + [_c.c_float, _c.c_int, _c.c_int, _P, _P]),
    'vmp_svae_subsample_rng': (_c.c_int, [_P, _P, _c.c_uint64, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P]),
    'vmp_svae_bwd_partial_words': (_c.c_int, [_c.c_int]),
    'vmp_svae_bwd_blocks': (_c.c_int, [_c.c_int64, _c.c_int]),
    'vmp_svae_workspace_bytes': (_c.c_size_t, [_c.c_int64, _c.c_int, _c.c_int]),
    'vmp_svae_bwd_blocks_for': (_c.c_int, [_c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    'vmp_svae_estep_bwd_n': (_c.c_int, [_P] * 13 + [_c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _c.c_size_t, _c.c_int, _P]),
    'vmp_svae_estep_bwd': (_c.c_int, [_P] * 13 + [_c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _c.c_size_t, _P]),
    'vmp_svae_subsample': (_c.c_int, [_P, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P]),
    'vmp_gauss_logprob_nat_per_samp': (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P]),
    'vmp_gauss_logprob_nat': (_c.c_int, [_P, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _P, _P]),
    'vmp_mix_mahalanobis': (_c.c_int, [_P] * 6 + [_c.c_int64, _c.c_int, _c.c_int, _P, _P]),
    'vmp_student_t_logprob': (_c.c_int, [_P, _P, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P]),
    'vmp_gauss_logprob_nat_per_samp_bwd': (_c.c_int, [_P, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P]),
    'vmp_student_t_bwd_blocks': (_c.c_int, [_c.c_int64, _c.c_int]),
    'vmp_student_t_logprob_bwd': (_c.c_int, [_P, _P, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P]),
    'vmp_eval_cell_metrics': (_c.c_int, [_P, _P, _P, _P, _c.c_int, _P, _c.c_int, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P]),
    'vmp_diag_gauss_loglike_fwd': (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _P, _P]),
    'vmp_diag_gauss_loglike_bwd': (_c.c_int, [_P, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _P, _P, _P]),
    'vmp_bernoulli_rows_fwd': (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P]),
    'vmp_bernoulli_rows_bwd': (_c.c_int, [_P, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P]),
    'vmp_decoder_param_words': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int]),
    'vmp_decoder_workspace_bytes': (_c.c_size_t, [_c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    'vmp_decoder_loglike_fwd': (_c.c_int, [_P] * 11 + [_c.c_int64] + [_c.c_int] * 5 + [_P] * 4),
    'vmp_decoder_loglike_bwd': (_c.c_int, [_P] * 12 + [_c.c_int64] + [_c.c_int] * 5 + [_P, _P, _P, _P, _c.c_size_t, _P]),
    'vmp_decoder_loglike_bwd_logw': (_c.c_int, [_P] * 3 + [_c.c_float] + [_P] * 9 + [_c.c_int64] + [_c.c_int] * 5
                                     + [_P, _P, _P, _P, _c.c_size_t, _P]),
    'vmp_mlp_gauss_head_fwd': (_c.c_int, [_P] * 10 + [_c.c_int64] + [_c.c_int] * 3 + [_c.c_float, _P, _P, _P]),
    'vmp_mlp_gauss_head_bwd': (_c.c_int, [_P] * 3 + [_c.c_float] + [_P] * 9 + [_c.c_int64] + [_c.c_int] * 3 + [_P, _P, _P, _c.c_size_t, _P]),
    'vmp_decoder_elbo': (_c.c_int, [_P] * 4 + [_c.c_float] + [_P] * 9 + [_c.c_int64] + [_c.c_int] * 5 + [_P] * 8
                         + [_c.c_size_t, _P, _c.c_size_t, _P]),
    'vmp_svae_elbo_tail_workspace_bytes': (_c.c_size_t, []),
    'vmp_svae_elbo_tail': (_c.c_int, [_P] * 3 + [_c.c_int64] + [_c.c_int] * 3 + [_c.c_float] + [_P] * 5 + [_c.c_size_t, _P]),
    'vmp_svae_step_scalars': (_c.c_int, [_P, _c.c_uint64, _c.c_float, _c.c_float, _P]),
    'vmp_mlp_gauss_head_fwd_prep': (_c.c_int, [_P] * 10 + [_c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _P, _P] + [_P] * 8 + [_c.c_int]
                                    + [_P] * 7 + [_P, _c.c_int, _P, _P, _P]),
    'vmp_svae_step_inputs': (_c.c_int, [_P, _c.c_uint64, _c.c_float, _c.c_float, _P, _P, _c.c_int64, _P]),
    'vmp_decoder_bwd_blocks': (_c.c_int, [_c.c_int64]),
    'vmp_decoder_elbo_lazy': (_c.c_int, [_P] * 3 + [_c.c_float] + [_P] * 9 + [_c.c_int64] + [_c.c_int] * 5 + [_P, _P, _P, _c.c_size_t, _P]),
    'vmp_mlp_gauss_head_bwd_lazy': (_c.c_int, [_P] * 3 + [_c.c_float] + [_P] * 9 + [_c.c_int64] + [_c.c_int] * 3 + [_P, _P, _c.c_size_t, _P]),
    'vmp_svae_bwd_tail_applies': (_c.c_int, [_c.c_int64, _c.c_int, _c.c_int, _c.c_int]),
    'vmp_svae_estep_bwd_tail': (_c.c_int, [_P] * 11 + [_c.c_float, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _c.c_size_t, _P, _P,
                                           _c.c_size_t, _P]),
    'vmp_svae_bwd_reduce_prep': (_c.c_int, [_P, _c.c_int, _P, _P, _P, _P, _c.c_int, _c.c_int, _P, _P, _P, _P]),
    'vmp_svae_prep_fwd2': (_c.c_int, [_P] * 8 + [_c.c_int, _c.c_int] + [_P] * 8),
    'vmp_svae_step_pack': (_c.c_int, [_P, _c.c_size_t] + [_P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _P] * 2 + [_P, _c.c_int, _P, _P, _P, _P, _P,
                                      _c.c_int64, _c.c_int, _c.c_int, _P, _c.c_int, _c.c_int, _P, _P]),
    'vmp_svae_step_final': (_c.c_int, [_P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P] * 2 + [_P, _c.c_int, _P] + [_P] * 4
                            + [_P, _P, _c.c_int64, _P, _P, _P, _P, _c.c_float, _c.c_int, _c.c_int, _P, _P, _c.c_int, _c.c_int, _P]
                            + [_c.c_double] * 4 + [_P, _P]),
    'vmp_adam_step': (_c.c_int, [_c.c_int] + [_P] * 5 + [_c.c_double] * 4 + [_P, _P]),
    'vmp_pack_f64': (_c.c_int, [_c.c_int] + [_P] * 5),
    'vmp_adam_step_packed': (_c.c_int, [_c.c_int, _P, _P, _P, _c.c_double] + [_P] * 4 + [_c.c_double] * 4 + [_P, _P]),
    'vmp_svae_phi_prep_fwd': (_c.c_int, [_P, _P, _P, _c.c_int, _c.c_int, _P, _P, _P, _P]),
    'vmp_svae_prep_fwd': (_c.c_int, [_P] * 8 + [_c.c_int, _c.c_int] + [_P] * 7),
    'vmp_svae_phi_prep_bwd': (_c.c_int, [_P] * 6 + [_c.c_int, _c.c_int] + [_P] * 4),
    'vmp_svae_bwd_reduce': (_c.c_int, [_P, _c.c_int, _c.c_int, _c.c_int] + [_P] * 7),
    'vmp_svae_theta_pack': (_c.c_int, [_P] * 5 + [_c.c_int, _c.c_int] + [_P] * 4),
    'vmp_svae_stats_cvi': (_c.c_int, [_P, _P, _c.c_int64] + [_P] * 16 + [_c.c_float, _c.c_int, _c.c_int, _P, _P]),
    'vmp_svae_cvi_update': (_c.c_int, [_P] * 17 + [_c.c_float, _c.c_int, _c.c_int, _P]),
    'vmp_mlp_gauss_bwd': (_c.c_int, [_P] * 12 + [_c.c_int64] + [_c.c_int] * 3 + [_P, _P, _P, _c.c_size_t, _P]),
    'vmp_comm_unique_id': (_c.c_int, [_P]),
    'vmp_comm_init_rank': (_c.c_int, [_c.POINTER(_c.c_void_p), _c.c_int, _P, _c.c_int]),
    'vmp_comm_destroy': (_c.c_int, [_P]),
    'vmp_pack_allreduce': (_c.c_int, [_P, _P, _c.c_size_t, _P]),
    'vmp_exch_bytes': (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int]),
    'vmp_exch_alloc': (_c.c_int, [_c.POINTER(_c.c_void_p), _c.c_size_t]),
    'vmp_exch_free': (_c.c_int, [_P]),
    'vmp_exch_export': (_c.c_int, [_P, _P]),
    'vmp_exch_open': (_c.c_int, [_P, _c.POINTER(_c.c_void_p)]),
    'vmp_exch_close': (_c.c_int, [_P]),
    'vmp_mix_finalize_exchange': (_c.c_int, [_P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int] + [_P] * 6 + [_P] * 9
                                  + [_P, _P, _c.c_int, _c.c_int, _c.c_uint64, _P, _P]),
    'vmp_mix_finalize_ws': (_c.c_int, [_P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int] + [_P] * 6 + [_P] * 9 + [_P, _P]),
}


class VmpError(RuntimeError):
    pass


def exported_symbols():
    """Names include/vmp_hip.h declares (kept in sync by tests/test_abi.py)."""
    return sorted(_SIGNATURES)


def lib():
    """Load libvmp_hip.so once.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VmpError('libvmp_hip.so not found at %s - build it with `make -C %s` '
                           '(there is no CPU fallback)' % (LIB_PATH, os.path.join(_HERE, 'csrc')))
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                # an OLDER build selected with VMP_LIB_PATH for an A/B measurement (tools/build_variant.sh) may predate an entry
                # point; the product's own library must export every symbol of include/vmp_hip.h
                if os.environ.get('VMP_LIB_PATH'):
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        if handle.vmp_abi_version() != 1:
            raise VmpError('libvmp_hip.so ABI version %d, expected 1' % handle.vmp_abi_version())
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().vmp_last_error()
        raise VmpError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))


def dev_f32(t, name, shape=None):
    """Validate a kernel operand: fp32, contiguous, on a GPU; returns the tensor (made contiguous)."""
    if not torch.is_tensor(t):
        raise VmpError('%s must be a torch tensor' % name)
    if not t.is_cuda:
        raise VmpError('%s is on %s: the N-sized VMP ops only run in HIP kernels on a GPU (no CPU fallback)'
                       % (name, t.device))
    if t.dtype != torch.float32:
        raise VmpError('%s must be float32 (got %s)' % (name, t.dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise VmpError('%s has shape %s, expected %s' % (name, tuple(t.shape), tuple(shape)))
    return t.contiguous()


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _raw_stream(index=None):
    """hipStream_t of the current stream as an int (torch's C accessor: torch.cuda.current_stream() builds a Python Stream
    object through several device look-ups - 10 us per call, a tenth of the eager minibatch step's host time)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice() if index is None else index)


def stream():
    return ctypes.c_void_p(_raw_stream())


_WS = {}
_WS_MAX = 16          # scratch buffers kept at most (least recently used beyond that are released to torch's allocator)
_WS_LOCK = threading.Lock()     # the cache is shared by all host threads: pop / insert / evict are one critical section


def release_workspaces(stream=None):
    """Forget the scratch buffers of `stream` (all streams if None); the memory returns to torch's caching allocator once
    the kernels queued on it have run (stream-ordered)."""
    sid = None if stream is None else stream.cuda_stream
    with _WS_LOCK:
        for k in [k for k in _WS if sid is None or k[2] == sid]:
            del _WS[k]


def snapshot_workspaces():
    with _WS_LOCK:
        return dict(_WS)


def take_workspaces(stream, before):
    """Remove and return the scratch buffers of `stream` that are not in the snapshot `before` (a graph capture takes
    ownership of the buffers its captured launches point into)."""
    sid = stream.cuda_stream
    with _WS_LOCK:
        mine = {k: v for k, v in _WS.items() if k[2] == sid and (k not in before or before[k] is not v)}
        for k in mine:
            del _WS[k]
    return mine


def workspace(device, nbytes):
    """Scratch buffer owned by the host side (the library never allocates), private to the calling (device, stream,
    host thread): kernels of one stream serialise on it, calls on other streams / from other threads get their own, so
    the C ABI's "callable concurrently from any host thread" holds through this layer as well (the cache itself is
    guarded by a lock).  A buffer that has to grow is replaced; the old one is released to torch's stream-ordered caching
    allocator (same stream: safe)."""
    sid = _raw_stream(device.index) if device.type == 'cuda' else 0
    key = (device.type, device.index, sid, threading.get_ident())
    with _WS_LOCK:
        buf = _WS.pop(key, None)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf                               # (re-)inserted last: dict order = recency
        while len(_WS) > _WS_MAX:                    # streams / threads that went away do not pin memory for ever
            del _WS[next(iter(_WS))]
    return buf
