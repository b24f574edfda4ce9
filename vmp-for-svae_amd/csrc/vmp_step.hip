// Scalar tail of the SVAE ELBO and the optimiser update as single launches.
//
// At the reference's operating point (minibatches of 64-100 rows, experiments.py:26) a training step is bound by the NUMBER
// of launches.  Between the decoder kernel and the E-step backward kernel the step evaluates (reference svae.py:216-254,
// vae.py:232-250)
//     rec = -1/(2S) sum_nk r_nk A_nk - N Dy/2 log(2 pi),   reg = sum_nk r_nk (T'_nk + log z_nk),   elbo = rec - reg
// with r = exp(log z) and A_nk = sum_s ll_nks, and autograd runs the same chain backwards: ~24 elementwise / reduction
// launches of 2 us each for (N,K)-sized tensors.  elbo_tail_kernel produces r, the gradients of sigma * elbo w.r.t. log z
// and T' and per-block partial sums of the three scalars in one launch; elbo_final_kernel (one wave) adds the partials.  adam_kernel is tf.train.AdamOptimizer's update (TF 1.3:
// lr_t = lr sqrt(1-b2^t)/(1-b1^t); var -= lr_t m / (sqrt(v) + eps)) for ALL parameter tensors in one launch.
#include "vmp_common.h"
#include "vmp_tail.h"
#include "vmp_step_parts.h"
#include "vmp_prep_parts.h"

using namespace vmp;

namespace {

constexpr int TAIL_THREADS = 256;

__global__ __launch_bounds__(TAIL_THREADS) void elbo_tail_kernel(TailArgs a) { elbo_tail_body(a, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(WAVE) void elbo_final_kernel(TailArgs a, unsigned ntb) { elbo_final_body(a, ntb); }

// ---- Adam ------------------------------------------------------------------------------------------------------------
constexpr int ADAM_MAX_TENSORS = 32;
constexpr int ADAM_CHUNK = 1024;       // elements per block
constexpr int ADAM_THREADS = 256;

struct AdamArgs {
    float* p[ADAM_MAX_TENSORS];
    const float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    unsigned n[ADAM_MAX_TENSORS];
    unsigned end_block[ADAM_MAX_TENSORS];   // inclusive prefix sums of ceil(n / ADAM_CHUNK)
    const float* lr_t_dev;
    float lr_t, b1, b2, c1, c2, eps;   // c = 1 - b, rounded from the fp64 difference
    int nt;
};

__global__ __launch_bounds__(ADAM_THREADS) void adam_kernel(AdamArgs a) {
    int t = 0;
    while (t < a.nt - 1 && blockIdx.x >= a.end_block[t]) ++t;
    const unsigned first = t ? a.end_block[t - 1] : 0u;
    const unsigned base = (blockIdx.x - first) * ADAM_CHUNK;
    const float lr_t = a.lr_t_dev ? *a.lr_t_dev : a.lr_t;
    const float c1 = a.c1, c2 = a.c2;
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const unsigned n = a.n[t];
#pragma unroll
    for (int j = 0; j < ADAM_CHUNK / ADAM_THREADS; ++j) {
        const unsigned i = base + j * ADAM_THREADS + threadIdx.x;
        if (i >= n) break;
        adam_update(p, m, v, i, g[i], lr_t, a.b1, a.b2, c1, c2, a.eps);
    }
}

// ---- data-parallel step: one packed fp64 buffer [moments | all gradients | scalars] ------------------------------------
// (the reference gathers per-tower gradients and averages them tensor by tensor, experiments.py:247-260 / tf_utils.py:52-87;
//  here every rank packs once, ONE all-reduce sums the buffer, and Adam reads the averaged gradients straight from it)
struct PackArgs {
    const void* src[ADAM_MAX_TENSORS];
    unsigned n[ADAM_MAX_TENSORS];
    unsigned off[ADAM_MAX_TENSORS];         // element offset in dst
    unsigned end_block[ADAM_MAX_TENSORS];
    unsigned f64mask;                        // bit t: src[t] holds doubles (else floats)
    double* dst;
    int nt;
};
__global__ __launch_bounds__(ADAM_THREADS) void pack_f64_kernel(PackArgs a) {
    int t = 0;
    while (t < a.nt - 1 && blockIdx.x >= a.end_block[t]) ++t;
    const unsigned first = t ? a.end_block[t - 1] : 0u;
    const unsigned base = (blockIdx.x - first) * ADAM_CHUNK;
    const unsigned n = a.n[t];
    double* __restrict__ d = a.dst + a.off[t];
    const bool f64 = (a.f64mask >> t) & 1u;
#pragma unroll
    for (int j = 0; j < ADAM_CHUNK / ADAM_THREADS; ++j) {
        const unsigned i = base + j * ADAM_THREADS + threadIdx.x;
        if (i >= n) break;
        d[i] = f64 ? static_cast<const double*>(a.src[t])[i] : (double)static_cast<const float*>(a.src[t])[i];
    }
}

struct AdamPackedArgs {
    float* p[ADAM_MAX_TENSORS];
    float* gout[ADAM_MAX_TENSORS];          // nullable: the averaged gradient is also written here (fp32)
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    unsigned n[ADAM_MAX_TENSORS];
    unsigned off[ADAM_MAX_TENSORS];         // element offset of the tensor's gradient in gbuf
    unsigned end_block[ADAM_MAX_TENSORS];
    const double* gbuf;
    const float* lr_t_dev;
    double gscale;
    float lr_t, b1, b2, c1, c2, eps;
    int nt;
};
__global__ __launch_bounds__(ADAM_THREADS) void adam_packed_kernel(AdamPackedArgs a) {
    int t = 0;
    while (t < a.nt - 1 && blockIdx.x >= a.end_block[t]) ++t;
    const unsigned first = t ? a.end_block[t - 1] : 0u;
    const unsigned base = (blockIdx.x - first) * ADAM_CHUNK;
    const float lr_t = a.lr_t_dev ? *a.lr_t_dev : a.lr_t;
    const float c1 = a.c1, c2 = a.c2;
    float* __restrict__ p = a.p[t];
    const double* __restrict__ g = a.gbuf + a.off[t];
    float* __restrict__ go = a.gout[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const unsigned n = a.n[t];
#pragma unroll
    for (int j = 0; j < ADAM_CHUNK / ADAM_THREADS; ++j) {
        const unsigned i = base + j * ADAM_THREADS + threadIdx.x;
        if (i >= n) break;
        const float gi = (float)(g[i] * a.gscale);          // mean over the ranks in fp64, rounded once (tf_utils.py:79)
        if (go) go[i] = gi;
        adam_update(p, m, v, i, gi, lr_t, a.b1, a.b2, c1, c2, a.eps);
    }
}

// [Philox key | CVI step size | Adam step size] of a graph-captured training step: the values travel in the launch packet
// (by value), so the caller needs no staging buffer that an asynchronous copy could still be reading when it is rewritten.
struct Words16 { unsigned long long key; float rho, lr_t; };
__global__ void step_scalars_kernel(Words16* dst, Words16 v) { *dst = v; }
// the same, and the minibatch copied into the captured step's static input by the same launch (round 6: one eager launch per replay, not two)
constexpr int INPUT_THREADS = 256;
__global__ __launch_bounds__(INPUT_THREADS) void step_inputs_kernel(Words16* dst, Words16 v, const float* __restrict__ src, float* __restrict__ y,
                                                                    unsigned n) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *dst = v;
    for (unsigned i = blockIdx.x * INPUT_THREADS + threadIdx.x; i < n; i += gridDim.x * INPUT_THREADS) y[i] = src[i];
}

// ---- the closing launch of the minibatch training step (round 6) -------------------------------------------------------------
// Everything that waits for the encoder's backward kernel and for nothing else, as ONE grid whose blocks take roles:
//   [decoder net | encoder net]  64 parameters per block: reduce the fused MLP backward kernel's per-block partials (dec_reduce_sum:
//                                the order of dec_reduce_kernel), store the gradient, apply Adam to the 64 parameters
//   [K blocks]                   phi_gmm: component k's rows of the E-step backward kernel's partials summed, the backward of the
//                                recognition unpacking on them, Adam on component k's elements (phi_prep_body<L, true, true>)
//   [K blocks]                   M-step moments of the minibatch + CVI update of theta_k (stats_cvi_body: vmp_svae_stats_cvi)
//   [1 block]                    the three ELBO scalars from the per-tile sums of vmp_svae_estep_bwd_tail (elbo_final_body)
// It replaces dec_reduce_tail_kernel's reduce blocks, svae_bwd_reduce_kernel, phi_prep_kernel<L, true>, dec_reduce_kernel,
// stats_cvi_kernel and adam_kernel: 6 launches -> 1.
// theta and the parameters are only written here; every role reads what the step's earlier launches left (experiments.py:267: the
// CVI update and the Adam step both read OLD values).
constexpr int FIN_THREADS = 64 * DEC_RED_GROUPS;
constexpr int FIN_NET_TENSORS = 9;
static_assert(FIN_THREADS == 64 * RED_GROUPS, "the phi role's reduction uses the block shape of svae_bwd_reduce_kernel");
struct FinNet {
    DecRedArgs red;                          // (out unused)
    float* p[FIN_NET_TENSORS];
    float* m[FIN_NET_TENSORS];
    float* v[FIN_NET_TENSORS];
    float* g[FIN_NET_TENSORS];               // reduced gradients out
    int off[FIN_NET_TENSORS + 1];            // flat offsets of the tensors inside a partial row
    int nb;                                  // blocks
};
struct FinArgs {
    FinNet net[2];
    PhiArgs phi;                             // (backward form: mu, Lraw, piraw, logpi in; g_mu, g_Lraw, g_piraw out)
    PhiAdam phi_adam;
    const float* partials;                   // (nblk, K, 2 (L + TRI + 1)) of the E-step backward kernel
    int nblk;
    SmallStatsArgs sa;
    CviArgs cvi;
    TailArgs tail;
    unsigned tail_n;
    const float* lr_t_dev;
    float lr_t, b1, b2, c1, c2, eps;
    // data-parallel form (vmp_svae_step_pack): nothing is updated - moments, gradients and scalars go, as doubles, into the packed
    // exchange buffer [moments | gradients in parameter order | elbo, rec, reg] that the step's ONE all-reduce sums
    double* xnet[2];                         // where the two nets' gradients start in the buffer (NULL: the single-process form)
    double* xscal;
};
template <int L>
__global__ __launch_bounds__(FIN_THREADS) void step_final_kernel(FinArgs a) {
    __shared__ double part[DEC_RED_GROUPS][64];
    __shared__ double spart[SMALL_STATS_GROUPS][80];
    __shared__ double st[80];
    static_assert(SMALL_STATS_GROUPS * 80 <= FIN_THREADS, "the moment role needs 12 x 80 threads");
    const float lr_t = a.lr_t_dev ? *a.lr_t_dev : a.lr_t;
    int b = blockIdx.x;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const FinNet& q = a.net[n];
        if (b < q.nb) {
            const double s = dec_reduce_sum(q.red, b, part);
            const int i = b * 64 + (threadIdx.x & 63);
            if ((threadIdx.x >> 6) == 0 && i < q.red.PW) {
                int t = 0;
                while (t < FIN_NET_TENSORS - 1 && i >= q.off[t + 1]) ++t;
                const unsigned j = (unsigned)(i - q.off[t]);
                const float gi = (float)s;
                q.g[t][j] = gi;
                if (a.xnet[n]) a.xnet[n][i] = (double)gi;
                else adam_update(q.p[t], q.m[t], q.v[t], j, gi, lr_t, a.b1, a.b2, a.c1, a.c2, a.eps);
            }
            return;
        }
        b -= q.nb;
    }
    if (b < a.phi.K) {
        PhiAdam ad = a.phi_adam;
        ad.lr_t = lr_t;
        phi_prep_body<L, true, true>(a.phi, b, a.partials, a.nblk, &ad);
        return;
    }
    b -= a.phi.K;
    if (b < a.cvi.K) {
        stats_cvi_body(a.sa, a.cvi, b, spart, st, a.xscal == nullptr);
        return;
    }
    if (threadIdx.x < WAVE) {
        elbo_final_body(a.tail, a.tail_n);
        if (threadIdx.x == 0 && a.xscal) {
#pragma unroll
            for (int i = 0; i < 3; ++i) a.xscal[i] = (double)a.tail.scal[i];       // (this thread wrote them: program order)
        }
    }
}

}  // namespace

extern "C" {

int vmp_svae_step_scalars(void* dst16, uint64_t philox_key, float cvi_step, float adam_step, void* stream) {
    if (!dst16 || (reinterpret_cast<uintptr_t>(dst16) & 7)) {
        set_error("vmp_svae_step_scalars: dst16 must be an 8-byte aligned device pointer to 16 bytes");
        return VMP_E_BADARG;
    }
    hipLaunchKernelGGL(step_scalars_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), static_cast<Words16*>(dst16),
                       Words16{(unsigned long long)philox_key, cvi_step, adam_step});
    return check_launch("vmp_svae_step_scalars");
}

int vmp_svae_step_inputs(void* dst16, uint64_t philox_key, float cvi_step, float adam_step, const float* y_src, float* y_dst,
                         int64_t n_floats, void* stream) {
    if (!dst16 || (reinterpret_cast<uintptr_t>(dst16) & 7)) {
        set_error("vmp_svae_step_inputs: dst16 must be an 8-byte aligned device pointer to 16 bytes");
        return VMP_E_BADARG;
    }
    if (n_floats < 0 || n_floats >= 4294967296LL || (n_floats > 0 && (!y_src || !y_dst))) {
        set_error("vmp_svae_step_inputs: bad minibatch copy (n = %lld)", (long long)n_floats);
        return VMP_E_BADARG;
    }
    long long blocks = (n_floats + INPUT_THREADS - 1) / INPUT_THREADS;
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(step_inputs_kernel, dim3((unsigned)blocks), dim3(INPUT_THREADS), 0, static_cast<hipStream_t>(stream),
                       static_cast<Words16*>(dst16), Words16{(unsigned long long)philox_key, cvi_step, adam_step}, y_src, y_dst,
                       (unsigned)n_floats);
    return check_launch("vmp_svae_step_inputs");
}

namespace {
int step_final_impl(const char* what, double* xbuf, const float* dec_part, int dec_blocks, int dec_in, int dec_units, int dec_out, float* const* dec_p,
                        float* const* dec_m, float* const* dec_v, float* const* dec_g, const float* enc_part, int enc_blocks,
                        int enc_in, int enc_units, int enc_out, float* const* enc_p, float* const* enc_m, float* const* enc_v,
                        float* const* enc_g, const float* partials, int nblk, const double* logpi, float* const* phi_p,
                        float* const* phi_g, float* const* phi_m, float* const* phi_v, const float* x_samples, const float* r, int64_t N, const float* const* prior,
                        float* const* theta, float* const* theta_star, const float* rho_dev, float rho, int K, int L,
                        double* stats_out, const double* tail_part, int tail_n, int Dy, float* scalars, double beta1, double beta2,
                        double eps, double lr_t, const float* lr_t_dev, void* stream) {
    const bool pack = xbuf != nullptr;
    if (K < 1 || K > VMP_MAX_K || L < 1 || L > VMP_MAX_D || N < 1 || N > SMALL_STATS_MAX_N || tail_n < 1 || tail_n > TAIL_MAX_BLOCKS || Dy < 1) {
        set_error("vmp_svae_step_final / _pack: K=%d L=%d N=%lld tail_n=%d outside the minibatch step's range (N <= %d)", K, L, (long long)N, tail_n,
                  SMALL_STATS_MAX_N);
        return VMP_E_DIM;
    }
    if (!dec_part || !enc_part || dec_blocks < 1 || enc_blocks < 1 || !dec_p || !dec_m || !dec_v || !dec_g || !enc_p || !enc_m || !enc_v ||
        !enc_g || !partials || nblk < 1 || !logpi || !phi_p || !phi_g || !phi_m || !phi_v || !x_samples || !r || (!pack && (!prior || !theta)) ||
        (!pack && !stats_out) || !tail_part || !scalars) {
        set_error("vmp_svae_step_final: NULL argument");
        return VMP_E_BADARG;
    }
    FinArgs a{};
    unsigned blocks = 0;
    const struct { const float* part; int nblk, in, units, out; float* const* p; float* const* m; float* const* v; float* const* g; } nets[2] = {
        {dec_part, dec_blocks, dec_in, dec_units, dec_out, dec_p, dec_m, dec_v, dec_g},
        {enc_part, enc_blocks, enc_in, enc_units, enc_out, enc_p, enc_m, enc_v, enc_g}};
    for (int n = 0; n < 2; ++n) {
        const int Li = nets[n].in, U = nets[n].units, Do = nets[n].out;
        if (Li < 1 || Li > 8 || Do < 1 || Do > 8 || U < 1 || U > 64) {
            set_error("vmp_svae_step_final: net %d sizes in=%d units=%d out=%d outside the fused MLP's range", n, Li, U, Do);
            return VMP_E_DIM;
        }
        const int sizes[FIN_NET_TENSORS] = {Li * U, U, U * U, U, U * 2 * Do, 2 * Do, Li * Do, Do, Do};   // the reference's variable order
        FinNet& q = a.net[n];
        int o = 0;
        for (int t = 0; t < FIN_NET_TENSORS; ++t) {
            if (!nets[n].p[t] || !nets[n].m[t] || !nets[n].v[t] || !nets[n].g[t]) {
                set_error("vmp_svae_step_final: net %d tensor %d: NULL pointer", n, t);
                return VMP_E_BADARG;
            }
            q.p[t] = nets[n].p[t]; q.m[t] = nets[n].m[t]; q.v[t] = nets[n].v[t]; q.g[t] = nets[n].g[t];
            q.off[t] = o;
            o += sizes[t];
        }
        q.off[FIN_NET_TENSORS] = o;
        if (o != vmp_decoder_param_words(Li, U, Do)) {
            set_error("vmp_svae_step_final: parameter layout mismatch (%d != %d words)", o, vmp_decoder_param_words(Li, U, Do));
            return VMP_E_DIM;
        }
        q.red = DecRedArgs{nets[n].part, q.p[FIN_NET_TENSORS - 1], nullptr, nets[n].nblk, o, o - Do, Do};
        q.nb = (o + 63) / 64;
        blocks += (unsigned)q.nb;
    }
    for (int t = 0; t < 3; ++t) {                           // phi_gmm/mu_k (K,L), L_k (K,L,L), log_pi_k (K)
        if (!phi_p[t] || !phi_g[t] || !phi_m[t] || !phi_v[t]) { set_error("vmp_svae_step_final: phi tensor %d: NULL pointer", t); return VMP_E_BADARG; }
        a.phi_adam.p[t] = phi_p[t]; a.phi_adam.m[t] = phi_m[t]; a.phi_adam.v[t] = phi_v[t];
    }
    a.phi.mu = phi_p[0]; a.phi.Lraw = phi_p[1]; a.phi.piraw = phi_p[2]; a.phi.logpi = logpi;
    a.phi.g_mu = phi_g[0]; a.phi.g_Lraw = phi_g[1]; a.phi.g_piraw = phi_g[2]; a.phi.K = K; a.phi.L = L;
    a.partials = partials; a.nblk = nblk;
    blocks += (unsigned)K;
    if (pack) {
        // [moments (K, 2+L+L*L) | phi_gmm/mu_k, L_k, log_pi_k | encoder net | decoder net | elbo, rec, reg]: the parameter order of
        // SVAETrainer.trainables() (experiments.py:160-181), as training.pack_exchange_buffer lays it out
        const int SW = 2 + L + L * L;
        double* o = xbuf + (size_t)K * SW;
        a.phi_adam.gx[0] = o; o += K * L;
        a.phi_adam.gx[1] = o; o += K * L * L;
        a.phi_adam.gx[2] = o; o += K;
        a.xnet[1] = o; o += a.net[1].red.PW;                 // encoder first
        a.xnet[0] = o; o += a.net[0].red.PW;
        a.xscal = o;
        for (int t = 0; t < 3; ++t) a.phi_adam.p[t] = nullptr;
        a.sa = SmallStatsArgs{x_samples, r, nullptr, xbuf, (int)N, L, K};
        a.cvi.K = K; a.cvi.L = L;
    } else {
    for (int t = 0; t < 5; ++t)
        if (!prior[t] || !theta[t]) { set_error("vmp_svae_step_final: prior / theta tensor %d: NULL pointer", t); return VMP_E_BADARG; }
    a.sa = SmallStatsArgs{x_samples, r, nullptr, stats_out, (int)N, L, K};
    a.cvi = CviArgs{stats_out, prior[0], prior[1], prior[2], prior[3], prior[4], theta[0], theta[1], theta[2], theta[3], theta[4],
                    theta_star ? theta_star[0] : nullptr, theta_star ? theta_star[1] : nullptr, theta_star ? theta_star[2] : nullptr,
                    theta_star ? theta_star[3] : nullptr, theta_star ? theta_star[4] : nullptr, rho_dev, rho, K, L};
    }
    blocks += (unsigned)K;
    a.tail.part = const_cast<double*>(tail_part);
    a.tail.scal = scalars;
    a.tail.cst = (double)N * Dy * 0.5 * 1.8378770664093453;            // log(2 pi): as tail_setup
    a.tail_n = (unsigned)tail_n;
    blocks += 1;
    a.lr_t_dev = lr_t_dev; a.lr_t = (float)lr_t; a.b1 = (float)beta1; a.b2 = (float)beta2;
    a.c1 = (float)(1.0 - beta1); a.c2 = (float)(1.0 - beta2); a.eps = (float)eps;
    a.phi_adam.b1 = a.b1; a.phi_adam.b2 = a.b2; a.phi_adam.c1 = a.c1; a.phi_adam.c2 = a.c2; a.phi_adam.eps = a.eps;
#define FIN_CALL(LL) case LL: hipLaunchKernelGGL((step_final_kernel<LL>), dim3(blocks), dim3(FIN_THREADS), 0, static_cast<hipStream_t>(stream), a); break
    switch (L) { FIN_CALL(1); FIN_CALL(2); FIN_CALL(3); FIN_CALL(4); FIN_CALL(5); FIN_CALL(6); FIN_CALL(7); default: FIN_CALL(8); }
#undef FIN_CALL
    return check_launch(what);
}
}  // namespace

int vmp_svae_step_final(const float* dec_part, int dec_blocks, int dec_in, int dec_units, int dec_out, float* const* dec_p,
                        float* const* dec_m, float* const* dec_v, float* const* dec_g, const float* enc_part, int enc_blocks,
                        int enc_in, int enc_units, int enc_out, float* const* enc_p, float* const* enc_m, float* const* enc_v,
                        float* const* enc_g, const float* partials, int nblk, const double* logpi, float* const* phi_p,
                        float* const* phi_g, float* const* phi_m, float* const* phi_v, const float* x_samples, const float* r, int64_t N,
                        const float* const* prior, float* const* theta, float* const* theta_star, const float* rho_dev, float rho, int K,
                        int L, double* stats_out, const double* tail_part, int tail_n, int Dy, float* scalars, double beta1, double beta2,
                        double eps, double lr_t, const float* lr_t_dev, void* stream) {
    return step_final_impl("vmp_svae_step_final", nullptr, dec_part, dec_blocks, dec_in, dec_units, dec_out, dec_p, dec_m, dec_v, dec_g,
                           enc_part, enc_blocks, enc_in, enc_units, enc_out, enc_p, enc_m, enc_v, enc_g, partials, nblk, logpi, phi_p, phi_g,
                           phi_m, phi_v, x_samples, r, N, prior, theta, theta_star, rho_dev, rho, K, L, stats_out, tail_part, tail_n, Dy,
                           scalars, beta1, beta2, eps, lr_t, lr_t_dev, stream);
}

// The closing launch of a DATA-PARALLEL minibatch step: the same block roles, but nothing is updated - this rank's moments, its 21
// gradients and its three scalars go as doubles into xbuf [moments (K, 2+L+L*L) | phi_gmm (3 tensors) | encoder (9) | decoder (9) |
// elbo, rec, reg], the packed buffer the step's one all-reduce sums (training.pack_exchange_buffer's layout; experiments.py:247-260);
// vmp_svae_cvi_update and vmp_adam_step_packed follow the all-reduce.  The fp32 gradients are also left in *_g.
int vmp_svae_step_pack(double* xbuf, size_t xbuf_doubles, const float* dec_part, int dec_blocks, int dec_in, int dec_units, int dec_out,
                       float* const* dec_p, float* const* dec_g, const float* enc_part, int enc_blocks, int enc_in, int enc_units,
                       int enc_out, float* const* enc_p, float* const* enc_g, const float* partials, int nblk, const double* logpi,
                       float* const* phi_p, float* const* phi_g, const float* x_samples, const float* r, int64_t N, int K, int L,
                       const double* tail_part, int tail_n, int Dy, float* scalars, void* stream) {
    if (!xbuf) { set_error("vmp_svae_step_pack: NULL exchange buffer"); return VMP_E_BADARG; }
    if (K < 1 || L < 1 || dec_in < 1 || dec_units < 1 || dec_out < 1 || enc_in < 1 || enc_units < 1 || enc_out < 1) {
        set_error("vmp_svae_step_pack: bad sizes");
        return VMP_E_DIM;
    }
    const size_t need = (size_t)K * (2 + L + L * L) + (size_t)K * L + (size_t)K * L * L + K +
                        (size_t)vmp_decoder_param_words(enc_in, enc_units, enc_out) + (size_t)vmp_decoder_param_words(dec_in, dec_units, dec_out) + 3;
    if (xbuf_doubles < need) {
        set_error("vmp_svae_step_pack: exchange buffer too small (%zu < %zu doubles)", xbuf_doubles, need);
        return VMP_E_WS;
    }
    // (Adam slots are not touched in this form: the parameter tensors stand in for the pointer checks)
    return step_final_impl("vmp_svae_step_pack", xbuf, dec_part, dec_blocks, dec_in, dec_units, dec_out, dec_p, dec_p, dec_p, dec_g, enc_part,
                           enc_blocks, enc_in, enc_units, enc_out, enc_p, enc_p, enc_p, enc_g, partials, nblk, logpi, phi_p, phi_g, phi_p,
                           phi_p, x_samples, r, N, nullptr, nullptr, nullptr, nullptr, 0.f, K, L, nullptr, tail_part, tail_n, Dy, scalars,
                           0.9, 0.999, 1e-8, 0.0, nullptr, stream);
}

size_t vmp_svae_elbo_tail_workspace_bytes(void) { return tail_workspace_bytes(); }

int vmp_svae_elbo_tail(const float* log_z, const float* T_prime, const float* ll, int64_t N, int K, int S, int Dy, float sigma,
                       float* scalars, float* g_log_z, float* g_T_prime, float* r, void* ws, size_t ws_bytes, void* stream) {
    if (N < 0 || K < 1 || S < 1 || Dy < 1) {
        set_error("vmp_svae_elbo_tail: bad sizes N=%lld K=%d S=%d Dy=%d", (long long)N, K, S, Dy);
        return VMP_E_DIM;
    }
    if (!scalars || !ws || (N > 0 && (!log_z || !T_prime || !ll || !g_log_z || !g_T_prime || !r))) {
        set_error("vmp_svae_elbo_tail: NULL argument");
        return VMP_E_BADARG;
    }
    if (ws_bytes < tail_workspace_bytes()) {
        set_error("vmp_svae_elbo_tail: workspace too small (%zu < %zu bytes)", ws_bytes, tail_workspace_bytes());
        return VMP_E_WS;
    }
    TailArgs a{};
    const unsigned blocks = tail_setup(a, log_z, T_prime, ll, N, K, S, Dy, sigma, scalars, g_log_z, g_T_prime, r, ws, TAIL_THREADS);
    hipLaunchKernelGGL(elbo_tail_kernel, dim3(blocks), dim3(TAIL_THREADS), 0, static_cast<hipStream_t>(stream), a);
    if (blocks > 1) hipLaunchKernelGGL(elbo_final_kernel, dim3(1), dim3(WAVE), 0, static_cast<hipStream_t>(stream), a, blocks);   // one block: it wrote the scalars itself
    return check_launch("vmp_svae_elbo_tail");
}

int vmp_pack_f64(int n_tensors, const void* const* src, const int* src_is_f64, const int64_t* sizes, double* dst, void* stream) {
    if (n_tensors < 0 || (n_tensors > 0 && (!src || !src_is_f64 || !sizes || !dst))) {
        set_error("vmp_pack_f64: bad arguments");
        return VMP_E_BADARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int t = 0;
    unsigned long long off = 0;
    while (t < n_tensors) {
        PackArgs a{};
        unsigned blocks = 0;
        int nt = 0;
        for (; t < n_tensors && nt < ADAM_MAX_TENSORS; ++t) {
            if (sizes[t] < 0 || off + (unsigned long long)sizes[t] >= 4294967296ULL || (sizes[t] > 0 && !src[t])) {
                set_error("vmp_pack_f64: tensor %d: NULL pointer or size out of range", t);
                return VMP_E_BADARG;
            }
            if (sizes[t] == 0) continue;
            a.src[nt] = src[t]; a.n[nt] = (unsigned)sizes[t]; a.off[nt] = (unsigned)off;
            if (src_is_f64[t]) a.f64mask |= 1u << nt;
            off += (unsigned long long)sizes[t];
            blocks += (unsigned)((sizes[t] + ADAM_CHUNK - 1) / ADAM_CHUNK);
            a.end_block[nt] = blocks;
            ++nt;
        }
        if (!nt) continue;
        a.nt = nt; a.dst = dst;
        hipLaunchKernelGGL(pack_f64_kernel, dim3(blocks), dim3(ADAM_THREADS), 0, s, a);
    }
    return check_launch("vmp_pack_f64");
}

int vmp_adam_step_packed(int n_tensors, float* const* params, const double* gbuf, const int64_t* goffsets, double gscale,
                         float* const* grads_out, float* const* m, float* const* v, const int64_t* sizes, double beta1,
                         double beta2, double eps, double lr_t, const float* lr_t_dev, void* stream) {
    if (n_tensors < 0 || (n_tensors > 0 && (!params || !gbuf || !goffsets || !m || !v || !sizes))) {
        set_error("vmp_adam_step_packed: bad arguments");
        return VMP_E_BADARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int t = 0;
    while (t < n_tensors) {
        AdamPackedArgs a{};
        unsigned blocks = 0;
        int nt = 0;
        for (; t < n_tensors && nt < ADAM_MAX_TENSORS; ++t) {
            if (sizes[t] < 0 || sizes[t] >= 2147483648LL || goffsets[t] < 0 || goffsets[t] + sizes[t] >= 4294967296LL || !params[t] ||
                !m[t] || !v[t]) {
                set_error("vmp_adam_step_packed: tensor %d: NULL pointer or size / offset out of range", t);
                return VMP_E_BADARG;
            }
            if (sizes[t] == 0) continue;
            a.p[nt] = params[t]; a.gout[nt] = grads_out ? grads_out[t] : nullptr; a.m[nt] = m[t]; a.v[nt] = v[t];
            a.n[nt] = (unsigned)sizes[t]; a.off[nt] = (unsigned)goffsets[t];
            blocks += (unsigned)((sizes[t] + ADAM_CHUNK - 1) / ADAM_CHUNK);
            a.end_block[nt] = blocks;
            ++nt;
        }
        if (!nt) continue;
        a.nt = nt; a.gbuf = gbuf; a.gscale = gscale; a.lr_t_dev = lr_t_dev; a.lr_t = (float)lr_t; a.b1 = (float)beta1;
        a.b2 = (float)beta2; a.c1 = (float)(1.0 - beta1); a.c2 = (float)(1.0 - beta2); a.eps = (float)eps;
        hipLaunchKernelGGL(adam_packed_kernel, dim3(blocks), dim3(ADAM_THREADS), 0, s, a);
    }
    return check_launch("vmp_adam_step_packed");
}

int vmp_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* m, float* const* v,
                  const int64_t* sizes, double beta1, double beta2, double eps, double lr_t, const float* lr_t_dev, void* stream) {
    if (n_tensors < 0 || (n_tensors > 0 && (!params || !grads || !m || !v || !sizes))) {
        set_error("vmp_adam_step: bad arguments");
        return VMP_E_BADARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int t = 0;
    while (t < n_tensors) {
        AdamArgs a{};
        unsigned blocks = 0;
        int nt = 0;
        for (; t < n_tensors && nt < ADAM_MAX_TENSORS; ++t) {
            if (sizes[t] < 0 || sizes[t] >= 2147483648LL || !params[t] || !grads[t] || !m[t] || !v[t]) {
                set_error("vmp_adam_step: tensor %d: NULL pointer or size out of range", t);
                return VMP_E_BADARG;
            }
            if (sizes[t] == 0) continue;
            a.p[nt] = params[t]; a.g[nt] = grads[t]; a.m[nt] = m[t]; a.v[nt] = v[t];
            a.n[nt] = (unsigned)sizes[t];
            blocks += (unsigned)((sizes[t] + ADAM_CHUNK - 1) / ADAM_CHUNK);
            a.end_block[nt] = blocks;
            ++nt;
        }
        if (!nt) continue;
        a.nt = nt; a.lr_t_dev = lr_t_dev; a.lr_t = (float)lr_t; a.b1 = (float)beta1; a.b2 = (float)beta2;
        a.c1 = (float)(1.0 - beta1); a.c2 = (float)(1.0 - beta2); a.eps = (float)eps;
        hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(ADAM_THREADS), 0, s, a);
    }
    return check_launch("vmp_adam_step");
}

}  // extern "C"
