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
        const float gi = g[i];
        const float mi = m[i] * a.b1 + c1 * gi;
        const float vi = v[i] * a.b2 + c2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * (mi / (__fsqrt_rn(vi) + a.eps));
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
        const float mi = m[i] * a.b1 + c1 * gi;
        const float vi = v[i] * a.b2 + c2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * (mi / (__fsqrt_rn(vi) + a.eps));
    }
}

// [Philox key | CVI step size | Adam step size] of a graph-captured training step: the values travel in the launch packet
// (by value), so the caller needs no staging buffer that an asynchronous copy could still be reading when it is rewritten.
struct Words16 { unsigned long long key; float rho, lr_t; };
__global__ void step_scalars_kernel(Words16* dst, Words16 v) { *dst = v; }

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
