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


def allreduce_sum_(t, group=None, comm=None):
    """In-place sum over ranks of the packed buffer `t`.
    comm (PackComm): the C ABI's own RCCL communicator (vmp_pack_allreduce) - the path a non-torch host binds;
    otherwise torch.distributed (backend nccl = RCCL; no-op when it is not initialised).  With the gloo backend (CPU
    rendezvous: tests, or several ranks sharing one GPU, which RCCL refuses) a device buffer is staged through the
    host - it is 10-80 KB."""
    if comm is not None:
        return comm.allreduce_(t)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if t.is_cuda and dist.get_backend(group) == 'gloo':
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class PackComm(object):
    """RCCL communicator owned through the C ABI (vmp_comm_* / vmp_pack_allreduce, include/vmp_hip.h): what a host
    without torch.distributed uses for the one collective of a step.  Rank 0 creates the id with PackComm.unique_id()
    and ships the 128 bytes to the other ranks out of band; every rank constructs PackComm(nranks, rank, id) with its
    device current."""

    @staticmethod
    def unique_id():
        import ctypes
        buf = ctypes.create_string_buffer(128)
        L.check(L.lib().vmp_comm_unique_id(buf), 'vmp_comm_unique_id')
        return buf.raw

    def __init__(self, nranks, rank, uid):
        import ctypes
        self.nranks, self.rank = int(nranks), int(rank)
        h = ctypes.c_void_p()
        L.check(L.lib().vmp_comm_init_rank(ctypes.byref(h), self.nranks, ctypes.create_string_buffer(uid, 128), self.rank),
                'vmp_comm_init_rank')
        self._h = h

    def allreduce_(self, t):
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise L.VmpError('PackComm.allreduce_ needs a contiguous fp64 GPU buffer')
        L.check(L.lib().vmp_pack_allreduce(self._h, L.ptr(t), t.numel(), L.stream()), 'vmp_pack_allreduce')
        return t

    def close(self):
        if self._h is not None:
            L.check(L.lib().vmp_comm_destroy(self._h), 'vmp_comm_destroy')
            self._h = None


class PeerExchange(object):
    """Peer-visible exchange buffers of the ONE-LAUNCH data-parallel finalize (include/vmp_hip.h,
    vmp_mix_finalize_exchange): every rank allocates vmp_exch_bytes(G, K, D) of uncached device memory, exports its IPC
    handle, gathers the G handles (any out-of-band channel; here torch.distributed's object all-gather - a one-time
    set-up exchange of 64 bytes per rank) and maps the peers' buffers.  No collective library is on the step path: the
    finalize kernels push their fp64 moments into each other's buffers and sum them in rank order.

    Lifetime: the mapped peer buffers and this rank's buffer are released by close_collective() (barrier, then close()) or by
    using the object as a context manager (`with PeerExchange(K, D) as ex:` - the exit is collective: EVERY rank must leave the
    block); close() alone is for after a barrier of the caller's own.  A multi-rank exchange that is garbage-collected unclosed
    keeps its mappings until the process exits and says so with a ResourceWarning.
    gather: a caller-supplied all-gather `gather(bytes64) -> list of world bytes64` used instead of torch.distributed.  It is called
    TWICE by the constructor, with 64-byte payloads both times: once with this rank's IPC handle, and - world > 1 - once more with
    the constant b'mapped' (zero-padded) as the rendezvous that tells every rank all buffers are mapped; the callback must accept
    arbitrary 64-byte payloads and may not assume a single call."""

    def __init__(self, K, D, group=None, rank=None, world=None, gather=None):
        import ctypes
        import torch.distributed as dist
        self.world = dist.get_world_size(group) if world is None else int(world)
        self.rank = dist.get_rank(group) if rank is None else int(rank)
        if not 1 <= self.world <= 16:
            raise L.VmpError('PeerExchange supports 1..16 ranks')
        self.K, self.D = int(K), int(D)
        nbytes = L.lib().vmp_exch_bytes(self.world, self.K, self.D)
        buf = ctypes.c_void_p()
        L.check(L.lib().vmp_exch_alloc(ctypes.byref(buf), nbytes), 'vmp_exch_alloc')
        self._buf = buf
        hb = ctypes.create_string_buffer(64)
        L.check(L.lib().vmp_exch_export(buf, hb), 'vmp_exch_export')
        if gather is not None:
            handles = gather(hb.raw)                      # caller-supplied all-gather of the 64-byte handles
        else:
            handles = [None] * self.world
            dist.all_gather_object(handles, hb.raw, group=group)
        self._peers = []
        for g, h in enumerate(handles):
            if g == self.rank:
                self._peers.append(ctypes.c_void_p(buf.value))
            else:
                pp = ctypes.c_void_p()
                L.check(L.lib().vmp_exch_open(ctypes.create_string_buffer(h, 64), ctypes.byref(pp)), 'vmp_exch_open')
                self._peers.append(pp)
        self.table = (ctypes.c_void_p * self.world)(*[p.value for p in self._peers])
        self.iteration = 0
        self.status = torch.zeros(1, dtype=torch.int32, device='cuda')
        # nobody may publish into a buffer a peer has not mapped yet, and the in-kernel wait is bounded (~4 s): without this
        # rendezvous a start skew between the ranks (data generation, uploads) could time a rank out before its peers arrive,
        # after which the sequence words never match again
        torch.cuda.synchronize()
        if self.world > 1:
            if gather is None:
                dist.barrier(group)
            else:
                gather(b'mapped'.ljust(64, b'\0'))        # a second round of the caller's all-gather is the rendezvous
        self._group, self._gather = group, gather

    @classmethod
    def try_open(cls, K, D, group=None):
        """Collective and exception-free: EVERY rank gets an exchange, or every rank gets None (some rank could not allocate, export or
        map a buffer - the reason is in .last_failure of the class).  Each stage is followed by an exchange of success flags, so that a
        rank that fails never leaves its peers waiting in a rendezvous.  For callers that want to fall back to the all-reduce form."""
        import ctypes
        import torch.distributed as dist
        self = cls.__new__(cls)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.K, self.D = int(K), int(D)
        self._peers, self._buf, self._group, self._gather = [], None, group, None
        why, raw = '', b''
        try:
            if not 1 <= self.world <= 16:
                raise L.VmpError('PeerExchange supports 1..16 ranks')
            buf = ctypes.c_void_p()
            L.check(L.lib().vmp_exch_alloc(ctypes.byref(buf), L.lib().vmp_exch_bytes(self.world, self.K, self.D)), 'vmp_exch_alloc')
            self._buf = buf
            hb = ctypes.create_string_buffer(64)
            L.check(L.lib().vmp_exch_export(buf, hb), 'vmp_exch_export')
            raw = hb.raw
        except Exception as e:                               # noqa: BLE001 - reported through the flags
            why = 'rank %d: %r' % (self.rank, e)
        got = [None] * self.world
        dist.all_gather_object(got, (why, raw), group=group)
        if not any(w for w, _ in got):
            try:
                for g, (_, h) in enumerate(got):
                    if g == self.rank:
                        self._peers.append(ctypes.c_void_p(self._buf.value))
                    else:
                        pp = ctypes.c_void_p()
                        L.check(L.lib().vmp_exch_open(ctypes.create_string_buffer(h, 64), ctypes.byref(pp)), 'vmp_exch_open')
                        self._peers.append(pp)
            except Exception as e:                           # noqa: BLE001
                why = 'rank %d: %r' % (self.rank, e)
            # pad for close(): it walks the list by rank index
            got2 = [None] * self.world
            dist.all_gather_object(got2, why, group=group)
        else:
            got2 = [w for w, _ in got]
        torch.cuda.synchronize()
        dist.barrier(group)                                  # nobody unmaps / publishes before everybody is here
        if any(got2):
            cls.last_failure = '; '.join(w for w in got2 if w)
            peers, self._peers = self._peers, []
            for g, pp in enumerate(peers):
                if g != self.rank:
                    L.lib().vmp_exch_close(pp)
            if self._buf is not None:
                L.lib().vmp_exch_free(self._buf)
                self._buf = None
            return None
        self.table = (ctypes.c_void_p * self.world)(*[p_.value for p_ in self._peers])
        self.iteration = 0
        self.status = torch.zeros(1, dtype=torch.int32, device='cuda')
        return self

    last_failure = ''

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close_collective()
        return False

    def close_collective(self):
        """barrier over the ranks (nobody is polling or pushing any more), then close(): the collective way to end an exchange"""
        if self.world > 1 and getattr(self, '_buf', None) is not None:
            torch.cuda.synchronize()
            if self._gather is None:
                import torch.distributed as dist
                dist.barrier(self._group)
            else:
                self._gather(b'closing'.ljust(64, b'\0'))
        self.close()

    def __del__(self):
        # Freeing or unmapping a buffer that a peer's finalize kernel may still be pushing into or polling is a GPU fault on
        # THAT rank, not a time-out.  Only a single-rank exchange is released implicitly; with peers the buffers live until the
        # owner calls close() after a barrier (or until the process exits).
        try:
            if self.world == 1:
                self.close()
            elif getattr(self, '_buf', None) is not None:
                import warnings
                warnings.warn('PeerExchange of %d ranks garbage-collected without close_collective(): its IPC mappings and device '
                              'buffer stay allocated until the process exits' % self.world, ResourceWarning)
        except Exception:
            pass

    def check(self):
        """Raise if a wait of an in-kernel exchange timed out (a peer never published: the iterations since are invalid).
        Reads one device word: call it where a host synchronisation is acceptable, not once per iteration."""
        if int(self.status.item()) != 0:
            raise L.VmpError('vmp_mix_finalize_exchange: a rank did not publish its moments within the time-out '
                             '(iteration %d, rank %d of %d)' % (self.iteration, self.rank, self.world))

    def close(self):
        """Unmap the peers' buffers and free this rank's (idempotent).  With several ranks call it after a barrier: a peer
        may still be polling or pushing.  This rank's own queued kernels are drained first."""
        if torch.cuda.is_available() and getattr(self, '_buf', None) is not None:
            torch.cuda.synchronize()
        peers, self._peers = getattr(self, '_peers', []), []
        for g, pp in enumerate(peers):
            if g != self.rank:
                L.lib().vmp_exch_close(pp)
        if getattr(self, '_buf', None) is not None:
            L.lib().vmp_exch_free(self._buf)
            self._buf = None


class DistributedVMPLoop(_mix.VMPLoop):
    """VMPLoop whose posterior update sees the statistics of ALL ranks' rows.  exchange=PeerExchange(...): the whole
    distributed finalize is ONE launch (moments pushed into the peers' buffers by the kernel itself); otherwise three
    launches around one all-reduce (RCCL through torch.distributed or the C ABI's PackComm)."""

    def __init__(self, x, r_init, flavour, kappa=None, u_init=None, prior=None, group=None, comm=None, exchange=None):
        # (the data-parallel iteration keeps two launches: the exchange sits between the partial sums and the posterior)
        super().__init__(x, r_init, flavour, kappa=kappa, u_init=u_init, prior=prior)
        self.group, self.comm, self.exchange = group, comm, exchange
        self._stats = torch.empty((self.K, L.lib().vmp_mix_stats_words(self.D)), dtype=torch.float64,
                                  device=self.x.device)

    def finalize(self, stats_out=None):
        if self.exchange is not None:
            ex, p, pr = self.exchange, self.post, self.prior
            L.check(L.lib().vmp_mix_finalize_exchange(
                L.ptr(self.ws), L.ptr(self.pivot), self.N, self.D, self.K, self.flavour, L.ptr(pr[0]), L.ptr(pr[1]),
                L.ptr(pr[2]), L.ptr(pr[3]), L.ptr(pr[4]), L.ptr(self.kappa), L.ptr(p['alpha']), L.ptr(p['beta']),
                L.ptr(p['m']), L.ptr(p['C']), L.ptr(p['v']), L.ptr(p['xbar']), L.ptr(p['S']), L.ptr(p['pi']),
                L.ptr(p['pack']), L.ptr(self._stats), ex.table, ex.world, ex.rank, ex.iteration, L.ptr(ex.status),
                L.stream()), 'vmp_mix_finalize_exchange')
            ex.iteration += 1
            if stats_out is not None:
                stats_out.copy_(self._stats)
            return
        # local reduction of the per-block partials -> (K, SW) fp64; sum over ranks; global posterior + pack
        pr = self.prior
        L.check(L.lib().vmp_mix_finalize_ws(L.ptr(self.ws), L.ptr(self.pivot), self.N, self.D, self.K, self.flavour,
                                            L.ptr(pr[0]), L.ptr(pr[1]), L.ptr(pr[2]), L.ptr(pr[3]), L.ptr(pr[4]),
                                            L.ptr(self.kappa), *([None] * 9), L.ptr(self._stats), L.stream()),
                'vmp_mix_finalize_ws')                   # all outputs NULL: fixed-order reduction of the partials only
        allreduce_sum_(self._stats, self.group, self.comm)
        p, pr = self.post, self.prior
        L.check(L.lib().vmp_mix_finalize(L.ptr(self._stats), self.D, self.K, self.flavour, L.ptr(pr[0]), L.ptr(pr[1]),
                                         L.ptr(pr[2]), L.ptr(pr[3]), L.ptr(pr[4]), L.ptr(self.kappa),
                                         L.ptr(p['alpha']), L.ptr(p['beta']), L.ptr(p['m']), L.ptr(p['C']),
                                         L.ptr(p['v']), L.ptr(p['xbar']), L.ptr(p['S']), L.ptr(p['pi']),
                                         L.ptr(p['pack']), L.stream()), 'vmp_mix_finalize')
        if stats_out is not None:
            stats_out.copy_(self._stats)

    def check(self):
        """Raise if an in-kernel wait of the peer exchange timed out (reads one device word: a host synchronisation)."""
        if self.exchange is not None:
            self.exchange.check()

    def theta(self):
        self.check()                                      # the caller is about to read the posterior: a rank that timed out must not pass
        return super().theta()

    def run(self, iterations):
        """`iterations` data-parallel VMP iterations (the base class enqueues the SINGLE-process iteration through one C call;
        here every iteration has its exchange)."""
        for _ in range(int(iterations)):
            self.step()
        self.check()
        return self.r
