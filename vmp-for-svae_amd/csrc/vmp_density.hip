// Stand-alone per-cell log-densities of the reference's distributions package, for callers that do not go through
// the fused E-step:  gaussian.log_probability_nat (distributions/gaussian.py:30-71),
// gaussian.log_probability_nat_per_samp (gaussian.py:74-105), student_t.log_probability_per_samp
// (distributions/student_t.py:7-39,59-61).  Forward only (the training step differentiates the fused kernels).
// One (n,k) cell per lane; the general (N,K,D,D) natural parameter is read per cell and factorised once
// (the reference LU-solves and Cholesky-factorises it separately).
#include "vmp_common.h"

using namespace vmp;

namespace {

constexpr float LOG_2PI_D = 1.8378770664093454836f;
__host__ __device__ constexpr int trd(int i, int j) { return i * (i + 1) / 2 + j; }

struct GArgs {
    const float* x;        // per_samp: (N,K,S,D);  normalised: (N,D)
    const float* eta1;     // (N,K,D)
    const float* eta2;     // (N,K,D,D)
    const float* logw;     // (K) log weights or NULL          (normalised form only)
    float* out;            // per_samp: (N,K,S);  normalised: (N,K)
    long long N;
    int K, S, normalise;
};

template <int D>
__global__ __launch_bounds__(256) void gauss_nat_kernel(GArgs a) {
    constexpr int TRI = D * (D + 1) / 2;
    __shared__ float scr[4][WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int K = a.K, S = a.S;
    const int RPT = WAVE / K, CT = RPT * K;
    const bool lane_on = lane < CT;
    const int r = lane / K, k = lane - r * K, rbase = lane_on ? r * K : 0;
    const long long ntiles = (a.N + RPT - 1) / RPT;
    for (long long t = (long long)blockIdx.x * nw + wave; t < ntiles; t += (long long)gridDim.x * nw) {
        const long long row = t * RPT + r;
        const bool on = lane_on && row < a.N;
        const long long cell = row * K + k;
        float Lm[TRI], e1[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            e1[i] = on ? a.eta1[cell * D + i] : 0.f;
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                // P = -2 eta2, symmetrised (+1e-20 I of gaussian.py:61 is a no-op in fp32)
                const float v = on ? -(a.eta2[(cell * D + i) * D + j] + a.eta2[(cell * D + j) * D + i]) : (i == j ? 1.f : 0.f);
                Lm[trd(i, j)] = v;
            }
        }
        // Cholesky, diagonal stored as reciprocal
        float ld = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            float s = Lm[trd(j, j)];
#pragma unroll
            for (int p = 0; p < j; ++p) s = fmaf(-Lm[trd(j, p)], Lm[trd(j, p)], s);
            const float rd = rsqrtf(s);
            ld += 0.5f * logf(s);
#pragma unroll
            for (int i = j + 1; i < D; ++i) {
                float tt = Lm[trd(i, j)];
#pragma unroll
                for (int p = 0; p < j; ++p) tt = fmaf(-Lm[trd(i, p)], Lm[trd(j, p)], tt);
                Lm[trd(i, j)] = tt * rd;
            }
            Lm[trd(j, j)] = rd;
        }
        // b = L^-1 eta1;  log N = -1/2 |L^T x - b|^2 - D/2 log 2pi + sum log L_ii
        float b[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float tt = e1[i];
#pragma unroll
            for (int p = 0; p < i; ++p) tt = fmaf(-Lm[trd(i, p)], b[p], tt);
            b[i] = tt * Lm[trd(i, i)];
        }
        const float cst = -0.5f * D * LOG_2PI_D + ld;
        if (!a.normalise) {
            for (int s = 0; s < S; ++s) {
                float q = 0.f;
                if (on) {
                    float xv[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) xv[i] = a.x[(cell * S + s) * D + i];
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        float y = xv[j] / Lm[trd(j, j)];                 // (L^T x)_j = sum_{i>=j} L_ij x_i
#pragma unroll
                        for (int i = j + 1; i < D; ++i) y = fmaf(Lm[trd(i, j)], xv[i], y);
                        y -= b[j];
                        q = fmaf(y, y, q);
                    }
                    a.out[cell * S + s] = cst - 0.5f * q;
                }
            }
        } else {
            float q = 0.f;
            if (on) {
                float xv[D];
#pragma unroll
                for (int i = 0; i < D; ++i) xv[i] = a.x[row * D + i];
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    float y = xv[j] / Lm[trd(j, j)];
#pragma unroll
                    for (int i = j + 1; i < D; ++i) y = fmaf(Lm[trd(i, j)], xv[i], y);
                    y -= b[j];
                    q = fmaf(y, y, q);
                }
            }
            float lp = on ? cst - 0.5f * q + (a.logw ? a.logw[k] : 0.f) : -INFINITY;
            // log-sum-exp over the K lanes of the row (gaussian.py:66-71)
            scr[wave][lane] = lp;
            __builtin_amdgcn_wave_barrier();
            float mx = -INFINITY;
            for (int j = 0; j < K; ++j) mx = fmaxf(mx, scr[wave][rbase + j]);
            float se = 0.f;
            for (int j = 0; j < K; ++j) se += expf(scr[wave][rbase + j] - mx);
            __builtin_amdgcn_wave_barrier();
            if (on) a.out[cell] = lp - mx - logf(se);
        }
    }
}

struct TArgs {
    const float* y;        // (N,K,S,D)
    const float* mu;       // (K,D)
    const float* W;        // (K,D,D) lower, W^T W = Sigma^-1
    const float* cst;      // (K) lgamma((v+D)/2) - lgamma(v/2) - D/2 log(pi v) - 1/2 logdet Sigma
    const float* nu;       // (K)
    float* out;            // (N,K,S)
    long long cells;       // N*K
    int K, S;
};

template <int D>
__global__ __launch_bounds__(256) void student_t_kernel(TArgs a) {
    const long long tot = a.cells * a.S;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < tot; g += (long long)gridDim.x * blockDim.x) {
        const long long cell = g / a.S;
        const int k = (int)(cell % a.K);
        float d[D];
#pragma unroll
        for (int i = 0; i < D; ++i) d[i] = a.y[g * D + i] - a.mu[k * D + i];
        float del2 = 0.f;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float yv = 0.f;
#pragma unroll
            for (int j = 0; j <= i; ++j) yv = fmaf(a.W[(k * D + i) * D + j], d[j], yv);
            del2 = fmaf(yv, yv, del2);
        }
        const float v = a.nu[k];
        a.out[g] = a.cst[k] - 0.5f * (v + (float)D) * log1p_f(del2 / v);      // student_t.py:34-37
    }
}

#define VMP_DISPATCH_DD(Dv, CALL)          \
    switch (Dv) {                           \
        case 1: { constexpr int DD = 1; CALL; } break; \
        case 2: { constexpr int DD = 2; CALL; } break; \
        case 3: { constexpr int DD = 3; CALL; } break; \
        case 4: { constexpr int DD = 4; CALL; } break; \
        case 5: { constexpr int DD = 5; CALL; } break; \
        case 6: { constexpr int DD = 6; CALL; } break; \
        case 7: { constexpr int DD = 7; CALL; } break; \
        case 8: { constexpr int DD = 8; CALL; } break; \
        default: break;                     \
    }

int chk(long long N, int K, int D, int S) {
    if (N <= 0 || S <= 0) { set_error("N and S must be positive"); return VMP_E_BADARG; }
    if (D < 1 || D > VMP_MAX_D) { set_error("D=%d outside compiled range 1..%d", D, VMP_MAX_D); return VMP_E_DIM; }
    if (K < 1 || K > VMP_MAX_K) { set_error("K=%d outside compiled range 1..%d", K, VMP_MAX_K); return VMP_E_DIM; }
    return 0;
}

// Stand-alone expected Mahalanobis distance of the mixture E-step (reference gmm.py:84-94, its missing-data variant
// gmm.py:97-114, smm.py:88-96): out_nk = v_k (x_n - m_k)^T P_k (x_n - m_k) + D / beta_k, masked entries of (x - m) zeroed.
// One (n,k) cell per lane.  (The VMP iteration never calls this: the distance lives inside the fused pass kernel.)
struct MArgs {
    const float *x, *m, *P, *v, *beta;
    const uint8_t* mask;
    float* out;
    long long cells;
    int K, D;
};

__global__ __launch_bounds__(256) void maha_kernel(MArgs a) {
    const int D = a.D;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < a.cells; c += (long long)gridDim.x * blockDim.x) {
        const long long n = c / a.K;
        const int k = (int)(c - n * a.K);
        float d[VMP_MAX_D];
        for (int i = 0; i < D; ++i) {
            const bool miss = a.mask && a.mask[n * D + i] != 0;
            d[i] = miss ? 0.f : a.x[n * D + i] - a.m[k * D + i];
        }
        float q = 0.f;
        for (int i = 0; i < D; ++i) {
            float t = 0.f;
            for (int j = 0; j < D; ++j) t = fmaf(a.P[((long long)k * D + i) * D + j], d[j], t);
            q = fmaf(d[i], t, q);
        }
        a.out[c] = a.v[k] * q + (float)D / a.beta[k];
    }
}

}  // namespace

extern "C" {

int vmp_gauss_logprob_nat_per_samp(const float* x, const float* eta1, const float* eta2, int64_t N, int K, int S, int D,
                                   float* out, void* stream) {
    int rc = chk(N, K, D, S);
    if (rc) return rc;
    if (!x || !eta1 || !eta2 || !out) { set_error("vmp_gauss_logprob_nat_per_samp: null pointer"); return VMP_E_BADARG; }
    GArgs a{x, eta1, eta2, nullptr, out, N, K, S, 0};
    const int RPT = WAVE / K;
    long long blocks = ((N + RPT - 1) / RPT + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    rc = -1;
    VMP_DISPATCH_DD(D, {
        hipLaunchKernelGGL((gauss_nat_kernel<DD>), dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
        rc = check_launch("gauss_nat_kernel");
    });
    return rc;
}

int vmp_gauss_logprob_nat(const float* x, const float* eta1, const float* eta2, const float* log_weights, int64_t N,
                          int K, int D, float* out, void* stream) {
    int rc = chk(N, K, D, 1);
    if (rc) return rc;
    if (!x || !eta1 || !eta2 || !out) { set_error("vmp_gauss_logprob_nat: null pointer"); return VMP_E_BADARG; }
    GArgs a{x, eta1, eta2, log_weights, out, N, K, 1, 1};
    const int RPT = WAVE / K;
    long long blocks = ((N + RPT - 1) / RPT + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    rc = -1;
    VMP_DISPATCH_DD(D, {
        hipLaunchKernelGGL((gauss_nat_kernel<DD>), dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
        rc = check_launch("gauss_nat_kernel");
    });
    return rc;
}

int vmp_student_t_logprob(const float* y, const float* mu, const float* W, const float* cst, const float* nu, int64_t N,
                          int K, int S, int D, float* out, void* stream) {
    int rc = chk(N, K, D, S);
    if (rc) return rc;
    if (!y || !mu || !W || !cst || !nu || !out) { set_error("vmp_student_t_logprob: null pointer"); return VMP_E_BADARG; }
    TArgs a{y, mu, W, cst, nu, out, (long long)N * K, K, S};
    long long blocks = ((long long)N * K * S + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    rc = -1;
    VMP_DISPATCH_DD(D, {
        hipLaunchKernelGGL((student_t_kernel<DD>), dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
        rc = check_launch("student_t_kernel");
    });
    return rc;
}

int vmp_mix_mahalanobis(const float* x, const float* m, const float* P, const float* v, const float* beta,
                        const uint8_t* miss_mask, int64_t N, int D, int K, float* out, void* stream) {
    int rc = chk(N, K, D, 1);
    if (rc) return rc;
    if (!x || !m || !P || !v || !beta || !out) { set_error("vmp_mix_mahalanobis: null pointer"); return VMP_E_BADARG; }
    MArgs a{x, m, P, v, beta, miss_mask, out, (long long)N * K, K, D};
    long long blocks = ((long long)N * K + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(maha_kernel, dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("maha_kernel");
}

}  // extern "C"
