"""Data-parallel pure mixture VMP: rows sharded over ranks (one process per GPU), parameters replicated.

The reference's only parallelism is the in-graph tower loop (experiments.py:196-265) that splits the
minibatch (data.py:174-175) and gathers per-tower results on the parameter device.  Here every rank streams
its own rows; the only exchange per VMP iteration is ONE all-reduce(sum) over RCCL of the K x (2+D+D^2) fp64
raw moments [N_k | W_k | sum w x | sum w x x^T] - 9.5 KB at K=16, D=8, latency-bound - after which every rank
runs the identical K-sized posterior update.  Raw (un-centred) moments are summed, so the result equals the
single-process result on the concatenated rows up to fp64 summation order.
"""
import torch

from .. import _lib as L
from . import _mix


def allreduce_sum_(t, group=None):
    """In-place sum over ranks (no-op when torch.distributed is not initialised)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class DistributedVMPLoop(_mix.VMPLoop):
    """VMPLoop whose posterior update sees the statistics of ALL ranks' rows."""

    def __init__(self, x, r_init, flavour, kappa=None, u_init=None, prior=None, group=None):
        super().__init__(x, r_init, flavour, kappa=kappa, u_init=u_init, prior=prior)
        self.group = group
        self._stats = torch.empty((self.K, L.lib().vmp_mix_stats_words(self.D)), dtype=torch.float64,
                                  device=self.x.device)

    def finalize(self, stats_out=None):
        # local reduction of the per-block partials -> (K, SW) fp64; sum over ranks; global posterior + pack
        pr = self.prior
        L.check(L.lib().vmp_mix_finalize_ws(L.ptr(self.ws), L.ptr(self.pivot), self.N, self.D, self.K, self.flavour,
                                            L.ptr(pr[0]), L.ptr(pr[1]), L.ptr(pr[2]), L.ptr(pr[3]), L.ptr(pr[4]),
                                            L.ptr(self.kappa), *([None] * 9), L.ptr(self._stats), L.stream()),
                'vmp_mix_finalize_ws')                   # all outputs NULL: fixed-order reduction of the partials only
        allreduce_sum_(self._stats, self.group)
        p, pr = self.post, self.prior
        L.check(L.lib().vmp_mix_finalize(L.ptr(self._stats), self.D, self.K, self.flavour, L.ptr(pr[0]), L.ptr(pr[1]),
                                         L.ptr(pr[2]), L.ptr(pr[3]), L.ptr(pr[4]), L.ptr(self.kappa),
                                         L.ptr(p['alpha']), L.ptr(p['beta']), L.ptr(p['m']), L.ptr(p['C']),
                                         L.ptr(p['v']), L.ptr(p['xbar']), L.ptr(p['S']), L.ptr(p['pi']),
                                         L.ptr(p['pack']), L.stream()), 'vmp_mix_finalize')
        if stats_out is not None:
            stats_out.copy_(self._stats)
