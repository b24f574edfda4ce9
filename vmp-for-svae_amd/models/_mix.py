"""Host-side engine of the pure mixture VMP (T1): thin wrappers over the vmp_mix_* C ABI
(include/vmp_hip.h) shared by models/gmm.py and models/smm.py."""
import os

import torch

from .. import _lib as L


def _dims(x, K):
    N, D = x.shape
    if not (1 <= D <= L.MAX_D):
        raise L.VmpError('D=%d outside the compiled range 1..%d' % (D, L.MAX_D))
    if not (1 <= K <= L.MAX_K):
        raise L.VmpError('K=%d outside the compiled range 1..%d' % (K, L.MAX_K))
    return N, D


def _ws(x, N, D, K):
    nbytes = L.lib().vmp_mix_workspace_bytes(N, D, K)
    return L.workspace(x.device, nbytes), nbytes


def pivot_of(x):
    """(D,) fp32 pivot near the data mean (vmp_mix_pivot) - see include/vmp_hip.h."""
    x = L.dev_f32(x, 'x')
    N, D = x.shape
    pv = torch.empty(D, dtype=torch.float32, device=x.device)
    L.check(L.lib().vmp_mix_pivot(L.ptr(x), N, D, L.ptr(pv), L.stream()), 'vmp_mix_pivot')
    return pv


SMALL_STATS_MAX_N = 512        # csrc/vmp_mix.hip: small_stats_kernel


def raw_stats(x, r, u=None, pivot=None):
    """(K, 2+D+D*D) fp64 raw moments [Nk | Wk | sum w x | sum w x x^T] (vmp_mix_stats).
    pivot=None: the data are shifted by a pivot near their mean before the products are formed (vmp_mix_pivot; what makes the
    one-pass moments match the reference's two-pass CENTRED S_k of gmm.py:39-46); pivot=False: no shift - for callers that use
    the raw moments as they are (the SVAE M-step in natural parameters, svae.py:154-176: theta* = prior + raw moments)."""
    x = L.dev_f32(x, 'x')
    N, K = r.shape
    if pivot is False:
        pivot = None
    elif pivot is None and N > SMALL_STATS_MAX_N:        # small batches: the library sums them directly in fp64 (one launch)
        pivot = pivot_of(x)
    _, D = _dims(x, K)
    r = L.dev_f32(r, 'r_nk', (N, K))
    if u is not None:
        u = L.dev_f32(u, 'u_nk', (N, K))
    stats = torch.empty((K, L.lib().vmp_mix_stats_words(D)), dtype=torch.float64, device=x.device)
    ws, nb = _ws(x, N, D, K)
    L.check(L.lib().vmp_mix_stats(L.ptr(x), L.ptr(r), L.ptr(u), L.ptr(pivot), N, D, K, L.ptr(stats), L.ptr(ws), nb,
                                  L.stream()), 'vmp_mix_stats')
    return stats


def _prior(prior, K, D, dev):
    a0, b0, m0, C0, v0 = prior
    return (L.dev_f32(a0.to(dev, torch.float32), 'alpha_0', (K,)), L.dev_f32(b0.to(dev, torch.float32).reshape(K), 'beta_0', (K,)),
            L.dev_f32(m0.to(dev, torch.float32), 'm_0', (K, D)), L.dev_f32(C0.to(dev, torch.float32), 'C_0', (K, D, D)),
            L.dev_f32(v0.to(dev, torch.float32), 'v_0', (K,)))


def finalize(stats, prior, flavour, kappa=None, want_pack=True):
    """Posterior (alpha, beta, m, C, v, xbar, S, pi) and the E-step pack from raw stats (vmp_mix_finalize)."""
    K = stats.shape[0]
    dev = stats.device
    D = prior[2].shape[1]
    a0, b0, m0, C0, v0 = _prior(prior, K, D, dev)
    f32 = dict(dtype=torch.float32, device=dev)
    out = dict(alpha=torch.empty(K, **f32), beta=torch.empty(K, **f32), m=torch.empty(K, D, **f32),
               C=torch.empty(K, D, D, **f32), v=torch.empty(K, **f32), xbar=torch.empty(K, D, **f32),
               S=torch.empty(K, D, D, **f32), pi=torch.empty(K, **f32))
    pack = torch.empty(K, L.lib().vmp_mix_pack_words(D), **f32) if want_pack else None
    kap = None if kappa is None else L.dev_f32(kappa.to(dev, torch.float32), 'kappa', (K,))
    L.check(L.lib().vmp_mix_finalize(L.ptr(stats), D, K, flavour, L.ptr(a0), L.ptr(b0), L.ptr(m0), L.ptr(C0), L.ptr(v0),
                                     L.ptr(kap), L.ptr(out['alpha']), L.ptr(out['beta']), L.ptr(out['m']),
                                     L.ptr(out['C']), L.ptr(out['v']), L.ptr(out['xbar']), L.ptr(out['S']),
                                     L.ptr(out['pi']), L.ptr(pack), L.stream()), 'vmp_mix_finalize')
    out['pack'] = pack
    return out


def pack_from_params(alpha_k, beta_k, m_k, P_k, v_k, flavour, kappa=None):
    K, D = m_k.shape
    dev = m_k.device
    f32 = dict(dtype=torch.float32, device=dev)
    args = [L.dev_f32(t.to(torch.float32), n) for t, n in ((alpha_k, 'alpha_k'), (beta_k, 'beta_k'), (m_k, 'm_k'),
                                                           (P_k, 'P_k'), (v_k, 'v_k'))]
    kap = None if kappa is None else L.dev_f32(kappa.to(dev, torch.float32), 'kappa', (K,))
    pack = torch.empty(K, L.lib().vmp_mix_pack_words(D), **f32)
    pi = torch.empty(K, **f32)
    L.check(L.lib().vmp_mix_pack_from_params(D, K, flavour, *[L.ptr(t) for t in args], L.ptr(kap), L.ptr(pack),
                                             L.ptr(pi), L.stream()), 'vmp_mix_pack_from_params')
    return pack, pi


def estep(x, pack, flavour, miss_mask=None, want_logr=False, want_stats=False, r_out=None, u_out=None):
    """Responsibilities (and SMM scales) for all rows; optionally fused raw stats of the new r."""
    x = L.dev_f32(x, 'x')
    K = pack.shape[0]
    N, D = _dims(x, K)
    f32 = dict(dtype=torch.float32, device=x.device)
    r = torch.empty(N, K, **f32) if r_out is None else r_out
    u = None
    if flavour == L.VMP_SMM:
        u = torch.empty(N, K, **f32) if u_out is None else u_out
    logr = torch.empty(N, K, **f32) if want_logr else None
    stats = torch.empty((K, L.lib().vmp_mix_stats_words(D)), dtype=torch.float64, device=x.device) if want_stats else None
    ws, nb, pivot = (None, 0, None)
    if want_stats:
        ws, nb = _ws(x, N, D, K)
        pivot = pivot_of(x)
    mask = None
    if miss_mask is not None:
        if not miss_mask.is_cuda:
            raise L.VmpError('missing_data_mask must be on the GPU')
        mask = miss_mask.to(torch.uint8).contiguous()
    L.check(L.lib().vmp_mix_estep(L.ptr(x), N, D, K, flavour, L.ptr(pack), L.ptr(mask), L.ptr(r), L.ptr(u), L.ptr(logr),
                                  L.ptr(pivot), L.ptr(stats), L.ptr(ws), nb, L.stream()), 'vmp_mix_estep')
    return r, u, logr, stats


def default_prior(K, D, device):
    """The prior gmm.inference / smm.inference hard-code (reference gmm.py:252-256): init_mm_params(K, D,
    alpha_scale=0.05/K, beta_scale=0.5, m_scale=0, C_scale=D+0.5, v_init=D+0.5), in standard form."""
    f32 = dict(dtype=torch.float32, device=device)
    alpha_0 = torch.full((K,), 0.05 / K, **f32)
    beta_0 = torch.full((K,), 0.5, **f32)
    m_0 = torch.zeros(K, D, **f32)
    C_0 = (D + 0.5) * torch.eye(D, **f32).expand(K, D, D).contiguous()
    v_0 = torch.full((K,), float(D + D + 0.5), **f32)
    return alpha_0, beta_0, m_0, C_0, v_0


class VMPLoop(object):
    """The iteration `sess.run(step)` drives in the reference (gmm.py:258-263 / smm.py:232-238):
    M-step from the current (r, u) -> E-step -> assign.  One iteration is TWO launches:
      vmp_mix_finalize_ws  (K blocks)   reduce the per-block fp64 partial moments + posterior + E-step pack
      vmp_mix_estep_fused  (streaming)  E-pass that also accumulates the raw moments of ITS OWN output,
    which are exactly the M-pass input of the next iteration; only the very first iteration needs a
    stand-alone M-pass (vmp_mix_stats_ws).
    accurate=True (round 6, opt-in): the E-part runs entirely in fp64 from an fp64 copy of the pack (vmp_mix_finalize_ws64 +
    vmp_mix_estep_accurate) and the moments of its output come from a separate fp64 M-pass (vmp_mix_stats_ws_accurate): three launches
    per iteration instead of one.  It is what meets the stated 1e-5 on the SMM's responsibilities at C5 (whose log rho is linear in
    the Mahalanobis distance with a factor (D + kappa) / 2 and reaches 1e2..1e3: beyond fp32); the default stays the fused pass.
    (A one-launch form of the iteration - posterior in the heads of the streaming launch - was built and measured in round 5:
    bit-identical and no faster; removed in round 6.)"""

    def __init__(self, x, r_init, flavour, kappa=None, u_init=None, prior=None, accurate=False):
        self.x = L.dev_f32(x, 'x')
        self.N, self.D = self.x.shape
        self.K = K = r_init.shape[1]
        _dims(self.x, K)
        dev = self.x.device
        self.flavour = flavour
        self.kappa = None if kappa is None else L.dev_f32(kappa.to(dev, torch.float32), 'kappa', (K,))
        prior = prior if prior is not None else default_prior(K, self.D, dev)
        self.prior = _prior(prior, K, self.D, dev)
        self.r = L.dev_f32(r_init, 'r_nk', (self.N, K)).clone()
        self.u = None
        if flavour == L.VMP_SMM:
            self.u = (torch.ones_like(self.r) if u_init is None else L.dev_f32(u_init, 'u_nk', (self.N, K)).clone())
        f32 = dict(dtype=torch.float32, device=dev)
        D = self.D
        self.post = dict(alpha=torch.empty(K, **f32), beta=torch.empty(K, **f32), m=torch.empty(K, D, **f32),
                         C=torch.empty(K, D, D, **f32), v=torch.empty(K, **f32), xbar=torch.empty(K, D, **f32),
                         S=torch.empty(K, D, D, **f32), pi=torch.empty(K, **f32),
                         pack=torch.empty(K, L.lib().vmp_mix_pack_words(D), **f32))
        self.logr = None
        self.nb = L.lib().vmp_mix_workspace_bytes(self.N, D, K)
        self.ws = torch.empty(self.nb, dtype=torch.uint8, device=dev)      # private: partials live across calls
        self.pivot = pivot_of(self.x)                                     # once per dataset
        self.accurate = bool(accurate)
        seed_fn = L.lib().vmp_mix_stats_ws_accurate if self.accurate else L.lib().vmp_mix_stats_ws
        L.check(seed_fn(L.ptr(self.x), L.ptr(self.r), L.ptr(self.u), L.ptr(self.pivot), self.N, D, K,
                        L.ptr(self.ws), self.nb, L.stream()), 'vmp_mix_stats_ws')
        self.iterations = 0
        self.pack64 = torch.empty(K, L.lib().vmp_mix_pack_words(D), dtype=torch.float64, device=dev) if self.accurate else None

    def finalize(self, stats_out=None):
        p, pr = self.post, self.prior
        if self.accurate:
            L.check(L.lib().vmp_mix_finalize_ws64(L.ptr(self.ws), L.ptr(self.pivot), self.N, self.D, self.K, self.flavour, L.ptr(pr[0]),
                                                  L.ptr(pr[1]), L.ptr(pr[2]), L.ptr(pr[3]), L.ptr(pr[4]), L.ptr(self.kappa),
                                                  L.ptr(p['alpha']), L.ptr(p['beta']), L.ptr(p['m']), L.ptr(p['C']),
                                                  L.ptr(p['v']), L.ptr(p['xbar']), L.ptr(p['S']), L.ptr(p['pi']),
                                                  L.ptr(p['pack']), L.ptr(self.pack64), L.ptr(stats_out), L.stream()), 'vmp_mix_finalize_ws64')
            return
        L.check(L.lib().vmp_mix_finalize_ws(L.ptr(self.ws), L.ptr(self.pivot), self.N, self.D, self.K, self.flavour, L.ptr(pr[0]),
                                            L.ptr(pr[1]), L.ptr(pr[2]), L.ptr(pr[3]), L.ptr(pr[4]), L.ptr(self.kappa),
                                            L.ptr(p['alpha']), L.ptr(p['beta']), L.ptr(p['m']), L.ptr(p['C']),
                                            L.ptr(p['v']), L.ptr(p['xbar']), L.ptr(p['S']), L.ptr(p['pi']),
                                            L.ptr(p['pack']), L.ptr(stats_out), L.stream()), 'vmp_mix_finalize_ws')

    def finalize_phase(self):
        """everything of an iteration that is not the streaming launch (bench.py brackets that launch with events): the finalize launch"""
        self.finalize()

    def estep(self, want_logr=False):
        """E-pass with fused moments on the current pack (the second launch of an iteration)"""
        if want_logr and self.logr is None:
            self.logr = torch.empty_like(self.r)
        if self.accurate:
            L.check(L.lib().vmp_mix_estep_accurate(L.ptr(self.x), self.N, self.D, self.K, self.flavour, L.ptr(self.pack64), L.ptr(self.r),
                                                   L.ptr(self.u), L.ptr(self.logr if want_logr else None), L.stream()), 'vmp_mix_estep_accurate')
            L.check(L.lib().vmp_mix_stats_ws_accurate(L.ptr(self.x), L.ptr(self.r), L.ptr(self.u), L.ptr(self.pivot), self.N, self.D,
                                                      self.K, L.ptr(self.ws), self.nb, L.stream()), 'vmp_mix_stats_ws_accurate')
            return
        L.check(L.lib().vmp_mix_estep_fused(L.ptr(self.x), self.N, self.D, self.K, self.flavour, L.ptr(self.post['pack']),
                                            L.ptr(self.r), L.ptr(self.u), L.ptr(self.logr if want_logr else None),
                                            L.ptr(self.pivot), L.ptr(self.ws), self.nb, L.stream()), 'vmp_mix_estep_fused')

    def stream_phase(self, want_logr=False):
        """the streaming launch of an iteration: the E-pass with the moments of its own output"""
        self.estep(want_logr)

    def step(self, want_logr=False):
        self.finalize_phase()
        self.stream_phase(want_logr)
        self.iterations += 1
        return self.r

    def run(self, iterations):
        """`iterations` VMP iterations enqueued by one C call (vmp_mix_iterate): no host work between launches."""
        if self.accurate:
            for _ in range(int(iterations)):
                self.step()
            return self.r
        p, pr = self.post, self.prior
        L.check(L.lib().vmp_mix_iterate(L.ptr(self.x), self.N, self.D, self.K, self.flavour, L.ptr(pr[0]), L.ptr(pr[1]),
                                        L.ptr(pr[2]), L.ptr(pr[3]), L.ptr(pr[4]), L.ptr(self.kappa), L.ptr(self.pivot),
                                        L.ptr(self.r), L.ptr(self.u), L.ptr(p['alpha']), L.ptr(p['beta']), L.ptr(p['m']),
                                        L.ptr(p['C']), L.ptr(p['v']), L.ptr(p['xbar']), L.ptr(p['S']), L.ptr(p['pi']),
                                        L.ptr(p['pack']), L.ptr(self.ws), self.nb, int(iterations), L.stream()),
                'vmp_mix_iterate')
        self.iterations += int(iterations)
        return self.r

    @property
    def stats(self):
        """Raw moments of the current r (reduces the partials the last pass left in the workspace)."""
        st = torch.empty((self.K, L.lib().vmp_mix_stats_words(self.D)), dtype=torch.float64, device=self.x.device)
        keep = {k: v.clone() for k, v in self.post.items()}
        self.finalize(stats_out=st)
        for k, v in keep.items():
            self.post[k].copy_(v)
        return st

    def theta(self):
        p = self.post
        return p['alpha'], p['beta'], p['m'], p['C'], p['v']

    def aux(self):
        p = self.post
        return p['xbar'], p['S'], p['pi']
