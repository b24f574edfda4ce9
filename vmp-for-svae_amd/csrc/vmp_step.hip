// Scalar tail of the SVAE ELBO and the optimiser update as single launches.
//
// At the reference's operating point (minibatches of 64-100 rows, experiments.py:26) a training step is bound by the NUMBER
// of launches.  Between the decoder kernel and the E-step backward kernel the step evaluates (reference svae.py:216-254,
// vae.py:232-250)
//     rec = -1/(2S) sum_nk r_nk A_nk - N Dy/2 log(2 pi),   reg = sum_nk r_nk (T'_nk + log z_nk),   elbo = rec - reg
// with r = exp(log z) and A_nk = sum_s ll_nks, and autograd runs the same chain backwards: ~24 elementwise / reduction
// launches of 2 us each for (N,K)-sized tensors.  elbo_tail_kernel produces the three scalars, r, and the gradients of
// sigma * elbo w.r.t. log z and T' in one launch.  adam_kernel is tf.train.AdamOptimizer's update (TF 1.3:
// lr_t = lr sqrt(1-b2^t)/(1-b1^t); var -= lr_t m / (sqrt(v) + eps)) for ALL parameter tensors in one launch.
#include "vmp_common.h"

using namespace vmp;

namespace {

constexpr int TAIL_THREADS = 256;
constexpr int TAIL_MAX_BLOCKS = 1024;

struct TailArgs {
    const float* lz;      // (NK)
    const float* Tp;      // (NK)
    const float* ll;      // (NK, S) per-sample reconstruction sums of the decoder kernel
    float* g_lz;          // (NK)  sigma * d elbo / d log z
    float* g_Tp;          // (NK)  sigma * d elbo / d T'
    float* r;             // (NK)  exp(log z)
    float* scal;          // [elbo, rec, reg]
    double* part;         // (blocks, 2)
    unsigned* ticket;     // zero between launches (the last block resets it)
    long long NK;
    int S;
    float sigma;
    double cst;           // N Dy / 2 log(2 pi)
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(TAIL_THREADS) void elbo_tail_kernel(TailArgs a) {
    __shared__ double sm[2][TAIL_THREADS / WAVE];
    __shared__ unsigned last;
    const float hs = 0.5f / (float)a.S;
    double wa = 0.0, rg = 0.0;
    for (long long c = (long long)blockIdx.x * TAIL_THREADS + threadIdx.x; c < a.NK; c += (long long)gridDim.x * TAIL_THREADS) {
        const float lz = a.lz[c], tp = a.Tp[c];
        const float r = expf(lz);
        const float* __restrict__ lr = a.ll + c * a.S;
        float A = 0.f;
        for (int s = 0; s < a.S; ++s) A += lr[s];
        const float w = hs * r;
        wa += (double)w * (double)A;
        rg += (double)r * (double)(tp + lz);
        a.r[c] = r;
        a.g_lz[c] = -a.sigma * (w * A + r * (tp + lz + 1.0f));
        a.g_Tp[c] = -a.sigma * r;
    }
    wa = wave_sum_d(wa);
    rg = wave_sum_d(rg);
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    if (lane == 0) { sm[0][wave] = wa; sm[1][wave] = rg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s0 = 0.0, s1 = 0.0;
        for (int i = 0; i < TAIL_THREADS / WAVE; ++i) { s0 += sm[0][i]; s1 += sm[1][i]; }
        // partials are published and read with device-scope atomics: the blocks of a launch sit on different XCDs (own L2s)
        __hip_atomic_store(a.part + 2 * blockIdx.x, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.part + 2 * blockIdx.x + 1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        const unsigned t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (!last || threadIdx.x != 0) return;
    __threadfence();
    double s0 = 0.0, s1 = 0.0;
    for (unsigned b = 0; b < gridDim.x; ++b) {                       // fixed order: deterministic
        s0 += __hip_atomic_load(a.part + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s1 += __hip_atomic_load(a.part + 2 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double rec = -s0 - a.cst;
    a.scal[0] = (float)(rec - s1);
    a.scal[1] = (float)rec;
    a.scal[2] = (float)s1;
    __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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

}  // namespace

extern "C" {

size_t vmp_svae_elbo_tail_workspace_bytes(void) { return (size_t)TAIL_MAX_BLOCKS * 2 * sizeof(double) + 64; }

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
    if (ws_bytes < vmp_svae_elbo_tail_workspace_bytes()) {
        set_error("vmp_svae_elbo_tail: workspace too small (%zu < %zu bytes)", ws_bytes, vmp_svae_elbo_tail_workspace_bytes());
        return VMP_E_WS;
    }
    TailArgs a{};
    a.lz = log_z; a.Tp = T_prime; a.ll = ll; a.g_lz = g_log_z; a.g_Tp = g_T_prime; a.r = r; a.scal = scalars;
    a.ticket = static_cast<unsigned*>(ws);
    a.part = reinterpret_cast<double*>(static_cast<char*>(ws) + 64);
    a.NK = (long long)N * K; a.S = S; a.sigma = sigma;
    a.cst = (double)N * Dy * 0.5 * 1.8378770664093453;             // log(2 pi)
    long long blocks = (a.NK + TAIL_THREADS - 1) / TAIL_THREADS;
    if (blocks > TAIL_MAX_BLOCKS) blocks = TAIL_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(elbo_tail_kernel, dim3((unsigned)blocks), dim3(TAIL_THREADS), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("vmp_svae_elbo_tail");
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
