"""The training step of the reference driver (experiments.py:143-151, 196-267) for one process per GPU.

    st = SVAETrainer(K, L, U, Dy, ...)            # parameters as experiments.py:154-181 creates them
    out = st.step(y_shard, noise=..., z_draws=...)

Reference semantics kept: the ELBO of a tower is a SUM over its shard; tower gradients are AVERAGED
(helpers/tf_utils.py:52-87); the M-step sees the statistics of the whole minibatch without N_data/N_batch
rescaling (svae.py:167-176); the CVI update and the Adam step both read OLD values (experiments.py:267);
lrcvi = lrcvi0 * decay_rate ** (global_step / 1000) (experiments.py:146); Adam is TensorFlow-1.3's formulation
(epsilon outside the bias correction).  What the reference does with an in-graph tower loop + gather on the
parameter device + stack/mean becomes ONE all-reduce (RCCL) of a packed fp64 buffer
    [ raw moments (K, 2+L+L*L) | flat gradients | elbo, neg_rec_err, regulariser ]
after which every rank applies the identical theta update and Adam step.

GraphedSVAEStep captures the whole single-process step of a fixed minibatch size as one HIP graph (the reference's
operating point - minibatches of 64-100 rows - is bound by launch count, not by the GPU); VAETrainer is the plain-VAE
baseline of models/vae.py.
"""
import math

import os

import torch

from . import _lib as L
from .models import _mix, _svae_ops, svae, vae


def exponential_decay(lr0, global_step, decay_steps, decay_rate):
    """tf.train.exponential_decay(..., staircase=False) (experiments.py:146)."""
    return lr0 * decay_rate ** (global_step / float(decay_steps))


class TFAdam(object):
    """tf.train.AdamOptimizer (TF 1.3): lr_t = lr sqrt(1-b2^t)/(1-b1^t); var -= lr_t m / (sqrt(v) + eps).
    Multi-tensor (torch._foreach_*) updates: a handful of launches for all 21 tensors instead of 7 per tensor."""

    def __init__(self, params, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def _fused_ok(self):
        return all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in self.params)

    def lr_t(self, t):
        return self.lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)

    @torch.no_grad()
    def apply_packed(self, gbuf, goffsets, gscale, grads_out, lr_t_dev=None):
        """Adam step whose gradient of tensor i is gscale * gbuf[goffsets[i]:...] (the all-reduced fp64 exchange buffer of the
        data-parallel step, gscale = 1 / ranks); the averaged fp32 gradients are also written to grads_out.  One launch."""
        import ctypes
        if not (self._fused_ok() and gbuf.is_cuda):
            gs = [(gbuf[o:o + p.numel()] * gscale).to(p.dtype).reshape(p.shape) for o, p in zip(goffsets, self.params)]
            for go, g in zip(grads_out, gs):
                go.copy_(g)
            return self.apply_gradients(gs, lr_t_dev=lr_t_dev)
        n = len(self.params)
        arr = ctypes.c_void_p * n
        if lr_t_dev is None:
            self.t += 1
            lr_t, lr_p = self.lr_t(self.t), None
        else:
            lr_t, lr_p = 0.0, L.ptr(lr_t_dev)
        L.check(L.lib().vmp_adam_step_packed(n, arr(*[p.data_ptr() for p in self.params]), L.ptr(gbuf),
                                             (ctypes.c_int64 * n)(*goffsets), float(gscale),
                                             arr(*[g.data_ptr() for g in grads_out]), arr(*[m.data_ptr() for m in self.m]),
                                             arr(*[v.data_ptr() for v in self.v]),
                                             (ctypes.c_int64 * n)(*[p.numel() for p in self.params]), self.b1, self.b2, self.eps,
                                             lr_t, lr_p, L.stream()), 'vmp_adam_step_packed')
        for p in self.params:
            torch.autograd.graph.increment_version(p)

    @torch.no_grad()
    def apply_gradients(self, grads, lr_t_dev=None):
        """lr_t_dev: the bias-corrected step size as a 0-dim device tensor (graph-captured steps: the caller advances
        self.t and refreshes the tensor before every replay); default: computed here from the step count."""
        grads = [g.to(p.dtype).contiguous() for g, p in zip(grads, self.params)]
        if self._fused_ok():
            # one launch for all tensors (csrc/vmp_step.hip) instead of 7 multi-tensor launches
            import ctypes
            n = len(self.params)
            arr = ctypes.c_void_p * n
            if lr_t_dev is None:
                self.t += 1
                lr_t, lr_p = self.lr_t(self.t), None
            else:
                lr_t, lr_p = 0.0, L.ptr(lr_t_dev)
            L.check(L.lib().vmp_adam_step(n, arr(*[p.data_ptr() for p in self.params]), arr(*[g.data_ptr() for g in grads]),
                                          arr(*[m.data_ptr() for m in self.m]), arr(*[v.data_ptr() for v in self.v]),
                                          (ctypes.c_int64 * n)(*[p.numel() for p in self.params]), self.b1, self.b2,
                                          self.eps, lr_t, lr_p, L.stream()), 'vmp_adam_step')
            for p in self.params:                          # written behind autograd's back: a graph that saved them must notice
                torch.autograd.graph.increment_version(p)
            return
        # torch fallback for what the kernel does not take (non-fp32 / non-contiguous / CPU parameters): contiguous
        # gradients with the parameters' strides keep torch on the multi-tensor fast path
        torch._foreach_mul_(self.m, self.b1)
        torch._foreach_add_(self.m, grads, alpha=1.0 - self.b1)
        torch._foreach_mul_(self.v, self.b2)
        torch._foreach_addcmul_(self.v, grads, grads, value=1.0 - self.b2)
        denom = torch._foreach_sqrt(self.v)
        torch._foreach_add_(denom, self.eps)
        data = [p.data for p in self.params]
        if lr_t_dev is None:
            self.t += 1
            torch._foreach_addcdiv_(data, self.m, denom, value=-self.lr_t(self.t))
        else:
            upd = torch._foreach_div(self.m, denom)
            torch._foreach_mul_(upd, lr_t_dev)
            torch._foreach_sub_(data, upd)


def pack_exchange_buffer(stats, grads, scalars):
    """[stats (fp64) | grads (flattened, fp64) | scalars] -> one contiguous fp64 buffer with ONE launch (vmp_pack_f64, pointer
    table in the launch packet) instead of a torch.cat over ~25 .double() copies.  Returns (buffer, gradient offsets).
    Replaces the tower gather of experiments.py:247-260."""
    import ctypes
    ts = [stats.reshape(-1)] + [g.reshape(-1) for g in grads] + [s.reshape(-1) for s in scalars]
    ok = all(t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.float64) for t in ts)
    sizes = [t.numel() for t in ts]
    offs, o = [], 0
    for n_ in sizes:
        offs.append(o)
        o += n_
    goffs = offs[1:1 + len(grads)]
    if not ok:                                                # host tensors (gloo tests on CPU): the torch form
        return torch.cat([t.double() for t in ts]), goffs
    buf = torch.empty(o, dtype=torch.float64, device=stats.device)
    n = len(ts)
    L.check(L.lib().vmp_pack_f64(n, (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts]),
                                 (ctypes.c_int * n)(*[int(t.dtype == torch.float64) for t in ts]),
                                 (ctypes.c_int64 * n)(*sizes), L.ptr(buf), L.stream()), 'vmp_pack_f64')
    return buf, goffs


def pack_for_allreduce(stats, grads, scalars):
    """[stats (fp64) | grads (flattened, fp64) | scalars] -> one contiguous fp64 buffer."""
    parts = [stats.reshape(-1).double()] + [g.reshape(-1).double() for g in grads] + [torch.stack([s.double().reshape(()) for s in scalars])]
    return torch.cat(parts)


def unpack_after_allreduce(buf, stats_shape, grad_shapes, n_scalars):
    o = 0
    n = 1
    for s in stats_shape:
        n *= s
    stats = buf[o:o + n].reshape(stats_shape)
    o += n
    grads = []
    for shp in grad_shapes:
        n = 1
        for s in shp:
            n *= s
        grads.append(buf[o:o + n].reshape(shp))
        o += n
    return stats, grads, buf[o:o + n_scalars]


class SVAETrainer(object):
    def __init__(self, K, Ld, U, Dy, nb_samples=10, lr=3e-4, lrcvi=0.2, decay_rate=0.95, stddev_init_nn=0.01, seed=0,
                 device='cuda', m_uniform=None, pi_normal=None, group=None, smm=False, dof=5.0, fused_decoder=True,
                 rng='philox', reference_call_order=False, direct_step=True):
        self.K, self.L, self.S = K, Ld, nb_samples
        # True (default): a whole-minibatch single-process GMM step on in-kernel noise runs as the 8-launch kernel sequence of
        # _step_direct (round 6) instead of the autograd graph over the same kernels (13 launches); False: always autograd
        self.direct_step = bool(direct_step)
        self._direct_shapes = {}
        # True: build the step exactly as experiments.py:209-229 does - svae.inference(...) WITHOUT theta, then
        # svae.compute_elbo(..., theta, phi_tilde, ...), which evaluates the theta term in a second launch of the fused
        # kernel (models/svae.py PhiTilde.theta_term).  False (default): theta goes into the E-step, one launch.
        self.reference_call_order = bool(reference_call_order)
        self.lr, self.lrcvi0, self.decay_rate = lr, lrcvi, decay_rate
        self.group = group
        self.device = torch.device(device)
        tanh = torch.tanh
        self.encoder_layers = [(U, tanh), (U, tanh), (Ld, 'natparam')]            # experiments.py:139
        self.decoder_layers = [(U, tanh), (U, tanh), (Dy, 'standard')]            # experiments.py:140
        self.stddev_init_nn = stddev_init_nn
        self.seed = seed
        self.smm = smm
        self.fused_decoder = fused_decoder      # decoder + reconstruction term in the fused MFMA kernels when covered
        if rng not in ('torch', 'philox'):
            raise ValueError("rng must be 'torch' (noise tensor from torch.randn) or 'philox' (drawn inside the E-step kernel)")
        # where eps comes from when the caller injects none.  'philox' (default): drawn inside the fused E-step kernel, as
        # the reference's tf.random_normal is drawn inside its step (svae.py:113-114) - no (N,K,L,S) tensor exists, and at
        # C3 the kernel is faster than the one that reads a noise tensor (1.85 vs 2.0 ms + 1.45 ms of randn).  Shapes the
        # in-kernel generator does not cover fall back to the same Philox stream materialised by its stand-alone kernel.
        self.rng = rng
        self.gmm_prior, self.theta = svae.init_mm(K, Ld, seed=seed, param_device=self.device, m_uniform=m_uniform)
        self.phi_gmm = list(svae.init_recognition_params(self.theta, K, seed=seed, param_device=self.device,
                                                         pi_normal=pi_normal))
        if smm:
            # experiments.py:154-176: Student-t point estimates mu_k, L_k (trainable, initialised from the PRIOR),
            # constant DoF, Dirichlet alpha_k updated by CVI; only the Dirichlet part of the prior is kept
            mu_k, L_k = svae.make_loc_scale_variables(self.gmm_prior, self.device)
            DoF = torch.full((K,), float(dof), dtype=torch.float32, device=self.device)
            self.theta = [self.theta[0].clone(), mu_k, L_k, DoF]
            self.gmm_prior = self.gmm_prior[0]
        self.global_step = 0
        self.opt = None
        self._neg_one = None        # GradSeed(-1): the step differentiates loss = -elbo by passing this as grad_outputs
        # the MLP variables are created HERE from `seed` (not lazily by the first forward pass, whose seed also carries the
        # step number and the rank): every replica starts from the same weights
        if not vae.net_variables('encoder_net'):
            vae.make_encoder(torch.zeros(1, Dy, dtype=torch.float32, device=self.device), self.encoder_layers,
                             self.stddev_init_nn, seed=seed)
        if not vae.net_variables('decoder_net'):
            vae.decoder_variables(Ld, self.decoder_layers, self.stddev_init_nn, seed, self.device)

    def trainables(self):
        """21 tensors in the reference's order: phi_gmm (3), encoder_net (9), decoder_net (9)."""
        names = ['phi_gmm/mu_k', 'phi_gmm/L_k', 'phi_gmm/log_pi_k']
        ts = list(self.phi_gmm)
        if self.smm:                                             # experiments.py:160-161
            names += ['theta/mu_k', 'theta/L_k']
            ts += [self.theta[1], self.theta[2]]
        for scope in ('encoder_net', 'decoder_net'):
            for n, p in vae.net_variables(scope):
                names.append(n)
                ts.append(p)
        return names, ts

    def _step_seed_at(self, global_step, chunk_index=0):
        import torch.distributed as dist
        rank = dist.get_rank(self.group) if (dist.is_available() and dist.is_initialized()) else 0
        return self.seed + int(global_step) + 1000003 * rank + 15485863 * int(chunk_index)

    def _step_seed(self, chunk_index=0):
        """Seed of this step's draws: every tower (rank) draws its own noise, as the reference's per-tower ops do, and
        every row chunk of a chunked step its own stream (the Philox counter of the in-kernel generator is the
        chunk-RELATIVE cell index: one seed for all chunks would give row n of every chunk the same eps)."""
        import torch.distributed as dist
        rank = dist.get_rank(self.group) if (dist.is_available() and dist.is_initialized()) else 0
        return self.seed + self.global_step + 1000003 * rank + 15485863 * int(chunk_index)

    def forward(self, y, noise=None, z_draws=None, u=None, chunk_index=0, _seed_dev=None):
        if noise is None and _seed_dev is not None:
            # graph-captured step: the key sits in a device word.  Round 6: where the E-step kernel's own generator covers the shape,
            # eps is drawn INSIDE it and its epilogue also does the one-draw sub-sampling - the stand-alone generator node
            # (philox_noise_kernel, ~5 us per replay) and the sub-sampling node (~6.5 us) are gone from the graph (rounds 4 / 5 kept the
            # stand-alone generator + the noise-tensor kernel there: in-kernel generation alone cost +8 / +1.7 us at N = 64).  Other
            # shapes materialise the same stream first.
            if L.lib().vmp_svae_rng_in_kernel(self.K, self.L, self.S):
                noise = _svae_ops.PhiloxNoise(0, self.S, seed_dev=_seed_dev, epilogue=True)
            else:
                pn = _svae_ops.PhiloxNoise(0, self.S, seed_dev=_seed_dev)
                noise = pn.materialise(y.shape[0], self.K, self.L, y.device)
                if u is None and z_draws is None:
                    u = pn
        elif noise is None and self.rng == 'philox':
            noise = 'philox'
        theta_in = None if self.reference_call_order else self.theta
        prep = None
        out = svae.inference(y, self.phi_gmm, self.encoder_layers, self.decoder_layers, self.S,
                             stddev_init_nn=self.stddev_init_nn, seed=self._step_seed(chunk_index), noise=noise,
                             z_draws=z_draws, theta=theta_in, lazy_decoder=self.fused_decoder, u=u, prep=prep)
        y_rec, phi_enc, x_k, x_s, log_z, _, phi_tilde = out
        elbo_fn = svae.compute_elbo_smm if self.smm else svae.compute_elbo
        if self._neg_one is None:
            self._neg_one = _svae_ops.GradSeed(-1.0, self.device)
        elbo, details = elbo_fn(y, y_rec, self.theta, phi_tilde, x_k, log_z, 'standard', grad_seed=self._neg_one)
        return elbo, details, x_k, x_s, log_z

    def step(self, y, noise=None, z_draws=None, chunk=None, u=None, _dev_scalars=None):
        """One training step.  `chunk` rows at a time (the ELBO is a sum over datapoints, so gradients and
        moments simply accumulate over chunks) - needed when N*K*S decoder rows do not fit at once.
        `u` (N,1) supplies the uniforms of the categorical sub-sampling; `_dev_scalars` = (lrcvi, lr_t[, Philox key]) as
        one-element device tensors is what GraphedSVAEStep captures with (step counters are then advanced by the caller).
        The step is three pieces - everything a rank does on its own rows (_step_front), the ONE exchange of a data-parallel
        step (_step_exchange) and the identical update every rank applies (_step_back) - so that a data-parallel step can be
        captured as two HIP graphs around its collective (GraphedSVAEStep)."""
        if self._direct_ok(y, noise, z_draws, chunk, u):
            if self._world() == 1:
                return self._step_direct(y, _dev_scalars)
            ctx = self._step_direct(y, _dev_scalars, pack=True)       # data-parallel: this rank's rows, up to the packed exchange buffer
        else:
            ctx = self._step_front(y, noise, z_draws, chunk, u, _dev_scalars)
        self._step_exchange(ctx)
        return self._step_back(ctx, _dev_scalars)

    def _direct_ok(self, y, noise, z_draws, chunk, u):
        """Whether _step_direct covers this call: GMM-SVAE, this process's whole minibatch (shard) at once (<= 512 rows, the minibatch
        forms of the E-step kernels), noise drawn in the kernels, fused encoder / decoder."""
        if not self.direct_step or self.smm or self.reference_call_order or not self.fused_decoder or self.rng != 'philox':
            return False
        if noise is not None or z_draws is not None or u is not None:
            return False
        if not (torch.is_tensor(y) and y.is_cuda and y.dtype == torch.float32 and y.dim() == 2):
            return False
        rows, Dy = y.shape
        if (chunk is not None and rows > int(chunk)) or not (0 < rows <= _svae_ops.STATS_CVI_MAX_ROWS):
            return False
        key = (rows, Dy)                                     # (the shape-only part of the answer is remembered: three library queries)
        ok = self._direct_shapes.get(key)
        if ok is None:
            lib = L.lib()
            ok = bool(vae._fused_mlp_eligible(Dy, self.encoder_layers) and vae.fused_decoder_eligible(self.L, self.decoder_layers)
                      and lib.vmp_svae_rng_in_kernel(self.K, self.L, self.S) and lib.vmp_svae_bwd_tail_applies(rows, self.K, self.L, self.S))
            self._direct_shapes[key] = ok
        return ok

    @torch.no_grad()
    def _step_direct(self, y, _dev_scalars=None, pack=False):
        """experiments.py:196-267 for one whole minibatch as SIX launches, no autograd graph (include/vmp_hip.h, "The minibatch
        training step"): encoder + recognition / theta prep, E-step (+ sub-sample), decoder value + gradients, ELBO tail + E-step
        backward, encoder backward, and one closing launch (partial rows -> phi_gmm gradients, both parameter reductions, Adam on
        the 21 tensors, M-step moments + CVI update, ELBO scalars).  Same kernels / device functions as the autograd step: every
        gradient, moment and parameter it leaves is bit-identical to that step's (tests/test_svae_gpu.py); the three ELBO scalars
        are summed per tile (fp64) and agree to fp32 rounding.
        pack=True (several ranks): the closing launch updates nothing - moments, gradients and scalars go into the packed fp64 exchange
        buffer (vmp_svae_step_pack) and the context of _step_exchange / _step_back is returned."""
        import ctypes
        lib, dev = L.lib(), y.device
        f32 = dict(dtype=torch.float32, device=dev)
        y = L.dev_f32(y, 'y')
        N, Dy = y.shape
        K, Ld, S, U = self.K, self.L, self.S, self.encoder_layers[0][0]
        names, params = self.trainables()
        if self.opt is None:
            self.opt = TFAdam(params, self.lr)
        opt = self.opt
        if not opt._fused_ok():
            raise L.VmpError('SVAETrainer: parameters must be contiguous fp32 GPU tensors')
        phi, enc, dec = params[:3], params[3:12], params[12:21]
        prior = [L.dev_f32(t.detach(), 'prior') for t in self.gmm_prior]
        for t in self.theta:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise L.VmpError('theta must be contiguous fp32 GPU tensors')
        st = L.stream()
        pp = lambda ts: [L.ptr(t) for t in ts]
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() if t is not None else None for t in ts])
        seed_dev = None if _dev_scalars is None or len(_dev_scalars) < 3 else _dev_scalars[2]
        rho_dev = None if _dev_scalars is None else _dev_scalars[0]
        lr_dev = None if _dev_scalars is None else _dev_scalars[1]
        lrcvi = exponential_decay(self.lrcvi0, self.global_step, 1000, self.decay_rate)
        # Scratch that nothing outside this step reads - encoder outputs, the K-sized prep, T', dL/dx, the per-sample reconstruction sums,
        # partial rows, the two MLP workspaces ... - is ONE allocation addressed by offsets (twenty torch.empty calls cost more host time
        # than the step's six launches); what the caller gets back (x, log z, the sub-sample, gradients, theta*, scalars) are tensors.
        nt = lib.vmp_svae_bwd_blocks_for(N, K, Ld, S, 0)
        PWp = lib.vmp_svae_bwd_partial_words(Ld)
        nb_dec, nb_enc = lib.vmp_decoder_bwd_blocks(N * K * S), lib.vmp_decoder_bwd_blocks(N)
        wsb_dec, wsb_enc = lib.vmp_decoder_workspace_bytes(N, K, S, Ld, U, Dy), lib.vmp_decoder_workspace_bytes(N, 1, 1, Dy, U, Ld)
        sizes = dict(eta1=4 * N * Ld, eta2d=4 * N * Ld, Lk=4 * K * Ld * Ld, P=4 * K * Ld * Ld, bias=4 * K, mk=4 * K * Ld, Wk=4 * K * Ld * Ld,
                     kappa=4 * K, logpi=8 * K, Tp=4 * N * K, r_epi=4 * N * K, dx=4 * N * K * S * Ld, ll=4 * N * K * S, g_eta1=4 * N * Ld,
                     g_eta2d=4 * N * Ld, partials=4 * nt * K * PWp, r=4 * N * K, tail_part=16 * nt, ws_dec=wsb_dec, ws_enc=wsb_enc)
        off, o = {}, 0
        for k_, nb_ in sizes.items():
            off[k_] = o
            o += (nb_ + 255) & ~255
        arena = torch.empty(o, dtype=torch.uint8, device=dev)
        base = arena.data_ptr()
        A = lambda k_: ctypes.c_void_p(base + off[k_])
        # 1: encoder (natparam head: eta1, -1/2 var) + recognition unpacking + theta packing (+ a replayed step's scalars from its table)
        mu_k, L_raw, pi_raw = phi
        tab = _dev_scalars[3] if (_dev_scalars is not None and len(_dev_scalars) > 3) else None     # (table, rows, counter, dst16)
        L.check(lib.vmp_mlp_gauss_head_fwd_prep(L.ptr(y), *pp(enc), N, Dy, Ld, U, -0.5, A('eta1'), A('eta2d'), L.ptr(mu_k), L.ptr(L_raw),
                                                L.ptr(pi_raw), *pp(self.theta), K, A('Lk'), A('P'), A('bias'), A('mk'), A('Wk'),
                                                A('kappa'), A('logpi'), L.ptr(tab[0]) if tab else None, tab[1] if tab else 0,
                                                L.ptr(tab[2]) if tab else None, L.ptr(tab[3]) if tab else None, st),
                'vmp_mlp_gauss_head_fwd_prep')
        # 2: E-step on in-kernel noise; its epilogue draws the one sub-sample per row
        x = torch.empty(N, K, S, Ld, **f32)
        lz = torch.empty(N, K, **f32)
        xs = torch.empty(N, Ld, **f32)
        key = 0 if seed_dev is not None else (self._step_seed(0) & 0xFFFFFFFFFFFFFFFF)
        L.check(lib.vmp_svae_estep_fwd_rng_epi(A('eta1'), A('eta2d'), L.ptr(mu_k), A('P'), A('bias'), key, L.ptr(seed_dev),
                                               A('mk'), A('Wk'), A('kappa'), None, N, K, Ld, S, L.ptr(x), L.ptr(lz), A('Tp'),
                                               L.ptr(xs), A('r_epi'), None, 0, st), 'vmp_svae_estep_fwd_rng_epi')
        # 3: decoder value + gradients of loss = -elbo (sigma = -1); parameter partials stay in ws_dec
        L.check(lib.vmp_decoder_elbo_lazy(L.ptr(x), L.ptr(y), L.ptr(lz), -1.0, *pp(dec), N, K, S, Ld, Dy, U, A('dx'), A('ll'),
                                          A('ws_dec'), wsb_dec, st), 'vmp_decoder_elbo_lazy')
        # 4: ELBO tail + E-step backward
        L.check(lib.vmp_svae_estep_bwd_tail(A('eta1'), A('eta2d'), L.ptr(mu_k), A('P'), A('bias'), A('mk'), A('Wk'),
                                            L.ptr(x), L.ptr(lz), A('Tp'), A('ll'), -1.0, A('dx'), N, K, Ld, S, A('g_eta1'),
                                            A('g_eta2d'), A('partials'), sizes['partials'], A('r'), A('tail_part'),
                                            sizes['tail_part'], st), 'vmp_svae_estep_bwd_tail')
        # 5: encoder backward; parameter partials stay in ws_enc
        L.check(lib.vmp_mlp_gauss_head_bwd_lazy(L.ptr(y), A('g_eta1'), A('g_eta2d'), -0.5, *pp(enc), N, Dy, Ld, U, None,
                                                A('ws_enc'), wsb_enc, st), 'vmp_mlp_gauss_head_bwd_lazy')
        # 6: the closing launch (phi_gmm gradients from the partial rows, both MLP reductions, Adam, moments + CVI, ELBO scalars)
        g_phi = [torch.empty_like(t) for t in phi]
        g_enc, g_dec = [torch.empty_like(t) for t in enc], [torch.empty_like(t) for t in dec]
        stats = torch.empty(K, 2 + Ld + Ld * Ld, dtype=torch.float64, device=dev)
        star = [torch.empty_like(t) for t in self.theta]
        scal = torch.empty(3, **f32)
        if pack:
            SW = 2 + Ld + Ld * Ld
            sizes = [p.numel() for p in params]
            goffs, o = [], K * SW
            for n_ in sizes:
                goffs.append(o)
                o += n_
            buf = torch.empty(o + 3, dtype=torch.float64, device=dev)
            L.check(lib.vmp_svae_step_pack(L.ptr(buf), buf.numel(), A('ws_dec'), nb_dec, Ld, U, Dy, arr(dec), arr(g_dec), A('ws_enc'),
                                           nb_enc, Dy, U, Ld, arr(enc), arr(g_enc), A('partials'), nt, A('logpi'), arr(phi), arr(g_phi),
                                           L.ptr(xs), A('r'), N, K, Ld, A('tail_part'), nt, Dy, L.ptr(scal), st), 'vmp_svae_step_pack')
            return dict(world=self._world(), names=names, params=params, grads=g_phi + g_enc + g_dec, stats=buf[:K * SW].view(K, SW),
                        fused_m=False, keep=dict(log_z=lz, x_samples=xs, x_k=x), r_whole=None, scal=(scal[0], scal[1], scal[2]), buf=buf,
                        goffs=goffs, mom_whole=None)
        if lr_dev is None:
            opt.t += 1
            lr_t = opt.lr_t(opt.t)
        else:
            lr_t = 0.0
        m, v = opt.m, opt.v
        L.check(lib.vmp_svae_step_final(A('ws_dec'), nb_dec, Ld, U, Dy, arr(dec), arr(m[12:21]), arr(v[12:21]), arr(g_dec),
                                        A('ws_enc'), nb_enc, Dy, U, Ld, arr(enc), arr(m[3:12]), arr(v[3:12]), arr(g_enc),
                                        A('partials'), nt, A('logpi'), arr(phi), arr(g_phi), arr(m[:3]), arr(v[:3]), L.ptr(xs),
                                        A('r'), N,
                                        arr(prior), arr(self.theta), arr(star), L.ptr(rho_dev),
                                        0.0 if rho_dev is not None else float(lrcvi), K, Ld, L.ptr(stats), A('tail_part'), nt, Dy,
                                        L.ptr(scal), opt.b1, opt.b2, opt.eps, lr_t, L.ptr(lr_dev), st), 'vmp_svae_step_final')
        for t in list(params) + list(self.theta):
            torch.autograd.graph.increment_version(t)
        if _dev_scalars is None:
            self.global_step += 1
        return dict(elbo=scal[0], neg_rec_err=scal[1], regulariser=scal[2], grads=dict(zip(names, g_phi + g_enc + g_dec)),
                    theta_star=star, lrcvi=lrcvi, log_z=lz, x_samples=xs, x_k=x, stats=stats)

    def _world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def _step_front(self, y, noise=None, z_draws=None, chunk=None, u=None, _dev_scalars=None):
        """Forward, backward, M-step moments of this rank's rows (experiments.py:196-244 for one tower) and, with more than one
        rank, the packed exchange buffer [moments | all gradients | elbo, rec, reg] (one launch)."""
        world = self._world()
        rows = y.shape[0]
        chunk = rows if chunk is None else int(chunk)
        names, params = None, None
        grads = stats = None
        elbo_t = rec_t = reg_t = 0.0
        keep = {}
        r_whole = None
        # the M-step moments and the CVI update are one launch when nothing has to be summed in between (chunks, ranks)
        fused_m = (not self.smm) and world == 1 and rows <= chunk and 0 < rows <= _svae_ops.STATS_CVI_MAX_ROWS
        mom_whole = None            # one chunk, one rank, and the E-step kernel left moment partials: reduce + CVI in one launch (_step_back)
        for ci, i in enumerate(range(0, rows, chunk)):
            ys = y[i:i + chunk]
            ns = None if noise is None else noise[i:i + chunk]
            zs = None if z_draws is None else z_draws[i:i + chunk]
            us = None if u is None else u[i:i + chunk]
            elbo, details, x_k, x_s, log_z = self.forward(ys, ns, zs, us, chunk_index=ci,
                                                          _seed_dev=None if _dev_scalars is None or len(_dev_scalars) < 3 else _dev_scalars[2])
            if params is None:
                names, params = self.trainables()
            g = torch.autograd.grad(elbo, params, grad_outputs=self._neg_one.tensor, allow_unused=True)   # loss = -elbo
            g = [torch.zeros_like(p) if gi is None else gi for gi, p in zip(g, params)]
            r_nk = details.r_nk if details.r_nk is not None else torch.exp(log_z.detach())
            mom = None if self.smm else getattr(details, 'mom', None)
            if fused_m:
                st = None                                                             # moments + CVI in one launch below
            elif mom is not None and world == 1 and rows <= chunk:
                st, mom_whole = None, mom                                             # partials -> moments -> CVI in one launch below
            elif mom is not None:                                                     # summed over chunks / ranks first
                st = _svae_ops.mom_cvi(mom)[0]
            elif self.smm:                                                            # svae.m_step_smm: N_k only
                from .models import gmm as _gmm
                st = _gmm.update_Nk(r_nk.contiguous()).double().reshape(-1, 1)
            else:
                st = _mix.raw_stats(x_s.detach().contiguous(), r_nk.contiguous(), pivot=False)  # HIP: (K, 2+L+L*L) fp64; raw, un-centred: no pivot pass
            grads = g if grads is None else [a + b for a, b in zip(grads, g)]
            if not fused_m and mom_whole is None:
                stats = st if stats is None else stats + st
            rec, reg = details[0], details[3]                # the two debug scalars in between are computed on access only
            if ci == 0:
                elbo_t, rec_t, reg_t = elbo.detach(), rec.detach(), reg.detach()
            else:
                elbo_t, rec_t, reg_t = elbo_t + elbo.detach(), rec_t + rec.detach(), reg_t + reg.detach()
            if rows <= chunk:
                keep = dict(log_z=log_z.detach(), x_samples=x_s.detach(), x_k=x_k.detach())
                r_whole = r_nk
            del elbo, details, x_k, x_s, log_z
        ctx = dict(world=world, names=names, params=params, grads=grads, stats=stats, fused_m=fused_m, keep=keep,
                   r_whole=r_whole, scal=(elbo_t, rec_t, reg_t), buf=None, goffs=None, mom_whole=mom_whole)
        if world > 1:
            # ONE pack launch, ONE all-reduce, and (in _step_back) ONE Adam launch that reads the averaged gradients from the
            # buffer: the exchange adds three launches to the step (it was a torch.cat over ~25 fp64 copies, ~25 slices and 21 divisions)
            ctx['grads'] = [g.contiguous() for g in grads]
            ctx['buf'], ctx['goffs'] = pack_exchange_buffer(stats, ctx['grads'], [elbo_t, rec_t, reg_t])
        return ctx

    def _step_exchange(self, ctx):
        """the one collective of a data-parallel step: sum of the packed buffer over the ranks (experiments.py:247-260 + tf_utils.py:52-87)"""
        if ctx['world'] > 1:
            from .models.parallel_mix import allreduce_sum_
            allreduce_sum_(ctx['buf'], self.group)

    def _step_back(self, ctx, _dev_scalars=None):
        """CVI update of theta and the Adam step from the (summed) moments and gradients: identical on every rank"""
        world, names, params, grads, stats = ctx['world'], ctx['names'], ctx['params'], ctx['grads'], ctx['stats']
        fused_m, keep, r_whole = ctx['fused_m'], ctx['keep'], ctx['r_whole']
        elbo_t, rec_t, reg_t = ctx['scal']
        packed = None
        if world > 1:
            buf, goffs = ctx['buf'], ctx['goffs']
            ns = stats.numel()
            stats = buf[:ns].reshape(stats.shape)
            elbo_t, rec_t, reg_t = buf[-3], buf[-2], buf[-1]
            packed = (buf, goffs, 1.0 / world)                                      # average_gradients (tf_utils.py:79)
        lrcvi = exponential_decay(self.lrcvi0, self.global_step, 1000, self.decay_rate)
        if self.opt is None:
            self.opt = TFAdam(params, self.lr)
        if fused_m:                                                                 # whole minibatch, one process, <= 512 rows
            rho_dev = None if _dev_scalars is None else _dev_scalars[0]
            stats, theta_star = _svae_ops.stats_cvi(keep['x_samples'], r_whole, self.gmm_prior, self.theta,
                                                    0.0 if rho_dev is not None else lrcvi, rho_dev=rho_dev)
        elif ctx.get('mom_whole') is not None:                                      # large single-process batch, K = 16, L = 8
            rho_dev = None if _dev_scalars is None else _dev_scalars[0]
            stats, theta_star = _svae_ops.mom_cvi(ctx['mom_whole'], self.gmm_prior, self.theta,
                                                  0.0 if rho_dev is not None else lrcvi, rho_dev=rho_dev, want_stats=False)
        elif self.smm:                                                              # experiments.py:252-256
            theta_star = [self.gmm_prior + stats[:, 0].float()]
            if _dev_scalars is not None:                                            # graph capture: the step size is a device word
                with torch.no_grad():
                    rho_d = _dev_scalars[0].reshape(())
                    self.theta[0].mul_(1.0 - rho_d).add_(theta_star[0].to(self.theta[0].dtype) * rho_d)
            else:
                svae.update_gmm_params(self.theta[:1], theta_star, lrcvi)
        elif _dev_scalars is not None:
            theta_star = svae.cvi_update_from_stats(self.gmm_prior, self.theta, stats.double(), 0.0,
                                                    step_size_dev=_dev_scalars[0])
        else:                                                                       # experiments.py:258-260
            theta_star = svae.cvi_update_from_stats(self.gmm_prior, self.theta, stats.double(), lrcvi)
        if packed is not None:
            grads = [g if g.dtype == torch.float32 else g.float() for g in grads]   # receive the averaged gradients (reported)
            self.opt.apply_packed(packed[0], packed[1], packed[2], grads, lr_t_dev=None if _dev_scalars is None else _dev_scalars[1])
            if _dev_scalars is None:
                self.global_step += 1
        elif _dev_scalars is not None:
            self.opt.apply_gradients(grads, lr_t_dev=_dev_scalars[1])
        else:
            self.opt.apply_gradients(grads)                                         # experiments.py:264-265
            self.global_step += 1
        out = dict(elbo=elbo_t, neg_rec_err=rec_t, regulariser=reg_t, grads=dict(zip(names, grads)),
                   theta_star=theta_star, lrcvi=lrcvi)
        out.update(keep)
        return out


class GraphedSVAEStep(object):
    """The whole training step of a fixed minibatch size captured ONCE as a HIP graph and replayed: at the reference's
    operating point (minibatches of 64-100 rows, experiments.py:26) the step is ~150 launches of microsecond kernels
    and is bound by launch overhead, not by the GPU.  Per call: copy the minibatch into the static input, refresh the
    noise / uniforms in place, write the two step-dependent scalars (CVI step size, bias-corrected Adam step size) to
    device memory, replay.  GMM- and SMM-SVAE; with several ranks (one process per GPU) the step is TWO graphs around its one
    collective (round 5).
    Noise.  Trainer with rng='philox' (the default): eps and the uniforms of the categorical draw are generated INSIDE the
    captured kernels from a Philox key the kernels read from a device word at run time; a call writes [key | CVI step size |
    Adam step size] with ONE launch (vmp_svae_step_scalars) and replays - the very stream of the same trainer stepped eagerly
    (call i here == step i there).  Trainer with rng='torch': the graph reads eps / u from static tensors that every call
    refills with torch's generator (three more launches per call), again the stream of that trainer stepped eagerly."""

    TABLE_ROWS = 1024

    def __init__(self, trainer, y_example, warmup=3, steps_per_replay=1):
        """steps_per_replay = n > 1 (round 6; needs the table mode below: the direct kernel sequence of a single-process GMM step on
        in-kernel noise): n CONSECUTIVE training steps are captured in one graph - step i reads minibatch i of the static input
        self.ys (n, N, Dy) and row `counter + i` of the scalar table, exactly the n steps the eager trainer would take - and a call
        takes the n minibatches, replays once and returns the n steps' outputs.  The ~5 us a replay costs besides its kernels (and the
        copy of the minibatches, one launch for all n) are paid once per n steps.
        (Parallel graph branches on side streams - noise generator / recognition prep beside the encoder, CVI beside Adam - were
        built and measured in round 5: bit-identical and 42 % SLOWER, a cross-stream edge of a HIP graph costs ~6 us on this runtime
        (profiles/r05_minibatch_fork_ab.txt); removed in round 6.)"""
        tr = self.tr = trainer
        dev = tr.device
        N = y_example.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        self.n_steps = int(steps_per_replay)
        if self.n_steps < 1:
            raise ValueError('steps_per_replay must be >= 1')
        self.ys = y_example.to(**f32).unsqueeze(0).repeat(self.n_steps, 1, 1).contiguous()    # static input: n minibatches
        self.y = self.ys[0]
        # the trainer's own Philox stream, keyed by a device word, generated by captured kernels (any shape: the graph uses the
        # stand-alone generator, which covers the shapes the E-step kernel's built-in generator does not)
        self.in_kernel_rng = tr.rng == 'philox'
        # [Philox key (int64) | CVI step size (f32) | Adam step size (f32)]: 16 bytes, refreshed by one launch per call
        self._dev16 = torch.zeros(16, dtype=torch.uint8, device=dev)
        self.seed_dev = self._dev16[:8].view(torch.int64)
        self.rho = self._dev16[8:12].view(torch.float32)
        self.lr_t = self._dev16[12:16].view(torch.float32)
        if self.in_kernel_rng:
            self.noise = self.u = None
        else:
            self.noise = torch.empty(N, tr.K, tr.L, tr.S, **f32)
            self.u = torch.empty(N, 1, **f32)
        self.gen = torch.Generator(device=dev).manual_seed(int(tr.seed))
        # Round 6: where the captured step is the trainer's direct kernel sequence, its FIRST launch also fetches the step's scalars from
        # a table of the coming TABLE_ROWS steps that the host fills in advance (exactly the values the eager launch passed by value) -
        # a replay whose minibatch already sits in the static input (self.y: the loader's copy target) is graph.replay() and nothing else
        self.table_mode = bool(self.in_kernel_rng and tr._world() == 1 and tr._direct_ok(self.y, None, None, None, None))
        self._table = torch.zeros(self.TABLE_ROWS, 2, dtype=torch.int64, device=dev) if self.table_mode else None
        self._counter = torch.zeros(1, dtype=torch.int64, device=dev) if self.table_mode else None
        self._table_base, self._table_used = None, 0
        if self.n_steps > 1 and not self.table_mode:
            raise L.VmpError('GraphedSVAEStep: steps_per_replay > 1 needs the direct single-process GMM step on in-kernel noise')
        # Warm-up steps (they create the variables / Adam slots and size the workspaces) must not train: everything a
        # step mutates is snapshotted first and put back before the capture, so that call number i of this object
        # is training step number i of the eager trainer (and of the reference).
        if not vae.net_variables('decoder_net'):
            Dy = tr.decoder_layers[-1][0]
            vae.make_encoder(torch.zeros(1, Dy, **f32), tr.encoder_layers, tr.stddev_init_nn, seed=tr.seed)
            vae.decoder_variables(tr.L, tr.decoder_layers, tr.stddev_init_nn, tr.seed, dev)
        _, params = tr.trainables()
        had_opt = tr.opt is not None
        snap = dict(params=[p.detach().clone() for p in params], theta=[t.detach().clone() for t in tr.theta],
                    step=tr.global_step, t=tr.opt.t if had_opt else 0,
                    m=[m.clone() for m in tr.opt.m] if had_opt else None,
                    v=[v.clone() for v in tr.opt.v] if had_opt else None,
                    gen=self.gen.get_state())
        ws_before = L.snapshot_workspaces()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        self._warm_and_capture(side, params, snap, had_opt, dev, warmup, ws_before)

    def _warm_and_capture(self, side, params, snap, had_opt, dev, warmup, ws_before):
        tr = self.tr
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._refresh()
                tr.step(self.y, noise=self.noise, u=self.u)      # (in-kernel noise: both None - the trainer's own default)
            with torch.no_grad():
                for p, q in zip(params, snap['params']):
                    p.copy_(q)
                for t, q in zip(tr.theta, snap['theta']):
                    t.copy_(q)
                for i in range(len(params)):
                    tr.opt.m[i].copy_(snap['m'][i]) if had_opt else tr.opt.m[i].zero_()
                    tr.opt.v[i].copy_(snap['v'][i]) if had_opt else tr.opt.v[i].zero_()
        torch.cuda.current_stream().wait_stream(side)
        tr.global_step, tr.opt.t = snap['step'], snap['t']
        self.gen.set_state(snap['gen'])
        self._refresh()
        self.graph = torch.cuda.CUDAGraph()
        self.graph_back = None
        cap_stream = torch.cuda.Stream(device=dev)
        dev_scalars = (self.rho, self.lr_t) + ((self.seed_dev,) if self.in_kernel_rng else ())
        if self.table_mode:
            dev_scalars = dev_scalars + ((self._table, self.TABLE_ROWS, self._counter, self._dev16),)
        self.world = tr._world()
        if self.world == 1:
            with torch.cuda.graph(self.graph, stream=cap_stream):
                self.outs = [tr.step(self.ys[i], noise=self.noise, u=self.u, _dev_scalars=dev_scalars) for i in range(self.n_steps)]
            self.out = self.outs[-1]
        else:
            # Data-parallel step (one process per GPU): TWO graphs around the ONE collective of the step.  Graph 1 = everything this
            # rank does on its own rows up to the packed exchange buffer (SVAETrainer._step_front), then the all-reduce of that
            # buffer is issued eagerly on the replay stream (RCCL; gloo stages it through the host), graph 2 = the CVI update and
            # the packed Adam step every rank applies identically (_step_back).  The buffer and everything graph 2 reads live in
            # the shared graph pool.  (Round 4 fell back to the eager step here: 0.87 ms instead of 0.12 ms at minibatch 64.)
            with torch.cuda.graph(self.graph, stream=cap_stream):
                if tr._direct_ok(self.y, self.noise, None, None, self.u):
                    self._ctx = tr._step_direct(self.y, dev_scalars, pack=True)
                else:
                    self._ctx = tr._step_front(self.y, self.noise, None, None, self.u, dev_scalars)
            self.graph_back = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_back, stream=cap_stream, pool=self.graph.pool()):
                self.out = tr._step_back(self._ctx, dev_scalars)
        self.gen.set_state(snap['gen'])     # the capture-time refresh drew nothing that a replay uses
        # the captured kernels hold raw pointers into the scratch buffers in use during the capture: keep exactly those
        # alive with the graph, and drop the warm-up side stream's buffers (never used again)
        L.release_workspaces(side)                # first: the side stream's scratch must not end up among the graph's references
        self._ws_refs = L.take_workspaces(cap_stream, ws_before)    # graph-pool memory: owned by this graph alone from here on
        _svae_ops.release_tail_workspaces(side)
        self._tail_refs = _svae_ops.release_tail_workspaces(cap_stream)    # graph-pool memory the captured tail launch points into

    def _refresh(self):
        tr = self.tr
        if not self.in_kernel_rng:
            self.noise.normal_(generator=self.gen)
            self.u.uniform_(generator=self.gen)
        L.check(L.lib().vmp_svae_step_scalars(L.ptr(self._dev16), tr._step_seed(0) & 0xFFFFFFFFFFFFFFFF,
                                              exponential_decay(tr.lrcvi0, tr.global_step, 1000, tr.decay_rate),
                                              tr.opt.lr_t(tr.opt.t + 1) if tr.opt is not None else 0.0, L.stream()),
                'vmp_svae_step_scalars')

    def _fill_table(self):
        """[Philox key | CVI step size | Adam step size] of the next TABLE_ROWS steps, from the trainer's counters as they stand"""
        import numpy as np
        tr, R = self.tr, self.TABLE_ROWS
        rows = np.zeros(R, dtype=[('key', '<u8'), ('rho', '<f4'), ('lr', '<f4')])
        for j in range(R):
            rows['key'][j] = tr._step_seed_at(tr.global_step + j) & 0xFFFFFFFFFFFFFFFF
            rows['rho'][j] = exponential_decay(tr.lrcvi0, tr.global_step + j, 1000, tr.decay_rate)
            rows['lr'][j] = tr.opt.lr_t(tr.opt.t + 1 + j)
        self._table[:R].copy_(torch.from_numpy(rows.view(np.int64).reshape(R, 2)))
        self._counter.zero_()
        self._table_base, self._table_used = (tr.global_step, tr.opt.t), 0

    def __call__(self, y):
        """y: the minibatch (N, Dy) - steps_per_replay = n > 1: the n minibatches (n, N, Dy); passing self.y / self.ys themselves (a loader
        that wrote into the static input) skips the copy.  Returns the step's output dict (n > 1: the list of the n steps' dicts)."""
        tr = self.tr
        n = self.n_steps
        if self.table_mode:
            b = self._table_base
            if b is None or self._table_used + n > self.TABLE_ROWS or (b[0] + self._table_used, b[1] + self._table_used) != (tr.global_step, tr.opt.t):
                self._fill_table()                  # first call, table used up, or the trainer was stepped outside this object
            dst = self.ys if n > 1 else self.y
            if not (torch.is_tensor(y) and y.is_cuda and y.data_ptr() == dst.data_ptr()):
                dst.copy_(y if torch.is_tensor(y) else torch.stack(list(y)))     # (a loader that writes into the static input saves this launch)
            self._table_used += n
            for i, o in enumerate(self.outs):
                o['lrcvi'] = exponential_decay(tr.lrcvi0, tr.global_step + i, 1000, tr.decay_rate)
            self.graph.replay()
            tr.opt.t += n
            tr.global_step += n
            return self.outs if n > 1 else self.out
        elif (self.in_kernel_rng and torch.is_tensor(y) and y.is_cuda and y.dtype == torch.float32 and y.is_contiguous()
                and tuple(y.shape) == tuple(self.y.shape)):
            # scalars of the step + the minibatch into the static input: ONE eager launch (round 6; it was a copy + a launch)
            same = y.data_ptr() == self.y.data_ptr()
            L.check(L.lib().vmp_svae_step_inputs(L.ptr(self._dev16), tr._step_seed(0) & 0xFFFFFFFFFFFFFFFF,
                                                 exponential_decay(tr.lrcvi0, tr.global_step, 1000, tr.decay_rate),
                                                 tr.opt.lr_t(tr.opt.t + 1), None if same else L.ptr(y), None if same else L.ptr(self.y),
                                                 0 if same else y.numel(), L.stream()), 'vmp_svae_step_inputs')
        else:
            self.y.copy_(y)
            self._refresh()
        self.out['lrcvi'] = exponential_decay(tr.lrcvi0, tr.global_step, 1000, tr.decay_rate)
        self.graph.replay()
        if self.graph_back is not None:
            tr._step_exchange(self._ctx)                     # the one collective of the step, between the two replays
            self.graph_back.replay()
        tr.opt.t += 1
        tr.global_step += 1
        return self.out


class VAETrainer(object):
    """The plain-VAE baseline of the reference (the training loop of models/vae.py:298-end): encoder with a 'standard'
    Gaussian head, reparameterised samples, Gaussian or Bernoulli decoder, ELBO of vae.compute_elbo, TF-Adam."""

    def __init__(self, Ld, U, Dy, nb_samples=10, lr=3e-4, stddev_init_nn=0.01, seed=0, decoder_type='standard',
                 device='cuda'):
        tanh = torch.tanh
        self.encoder_layers = [(U, tanh), (U, tanh), (Ld, 'standard')]
        self.decoder_layers = [(U, tanh), (U, tanh), (Dy, decoder_type)]
        self.decoder_type, self.S, self.lr = decoder_type, nb_samples, lr
        self.stddev_init_nn, self.seed, self.device = stddev_init_nn, seed, torch.device(device)
        self.global_step = 0
        self.opt = None

    def forward(self, y, noise=None):
        mu, var = vae.make_encoder(y, self.encoder_layers, self.stddev_init_nn, seed=self.seed)
        x = vae.reparam_trick_sampling(mu, var, self.S, self.seed + self.global_step, noise=noise)
        dec = vae.make_decoder(x, self.decoder_layers, self.stddev_init_nn, seed=self.seed)
        return vae.compute_elbo(y, mu, var, dec, self.decoder_type), (mu, var, x, dec)

    def step(self, y, noise=None):
        elbo, _ = self.forward(y, noise)
        names = [n for scope in ('encoder_net', 'decoder_net') for n in sorted(vae.VARIABLES) if n.startswith(scope + '/')]
        params = [vae.VARIABLES[n] for n in names]
        grads = torch.autograd.grad(-elbo, params)
        if self.opt is None:
            self.opt = TFAdam(params, self.lr)
        self.opt.apply_gradients(grads)
        self.global_step += 1
        return dict(elbo=elbo.detach(), grads=dict(zip(names, grads)))
