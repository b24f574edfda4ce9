// Stand-alone per-cell log-densities of the reference's distributions package, for callers that do not go through
// the fused E-step:  gaussian.log_probability_nat (distributions/gaussian.py:30-71),
// gaussian.log_probability_nat_per_samp (gaussian.py:74-105), student_t.log_probability_per_samp
// (distributions/student_t.py:7-39,59-61), and - round 3 - the adjoints of the two per-sample densities (what TF's autodiff
// does through gaussian.py:74-105 and student_t.py:7-39 when the reference differentiates compute_elbo, svae.py:236-243,
// 291-300; experiments.py:232).  The training step itself differentiates the fused E-step kernels.
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


// ---------------------------------------------------------------------------------------------------------
// Adjoint of log N(x_s | eta1, eta2) (gaussian.py:74-105) per cell: with P = -2 sym(eta2), Sigma = P^-1, mu = Sigma eta1
// (the exponential-family identities d A / d eta = E[t(x)]):
//   d/dx_s   = g_s (eta1 - P x_s)
//   d/deta1  = sum_s g_s (x_s - mu)
//   d/deta2  = sum_s g_s (x_s x_s^T - mu mu^T - Sigma)          (full symmetric D x D; what TF's autodiff of the three
//              terms x^T eta2 x, 1/4 eta1^T eta2^-1 eta1 and 1/2 logdet(-2 eta2) adds up to for a symmetric eta2)
// One cell per thread; the S samples are walked once for the three weighted sums.
// ---------------------------------------------------------------------------------------------------------
struct GBArgs {
    const float *x, *eta1, *eta2, *g;      // (N,K,S,D), (N,K,D), (N,K,D,D), (N,K,S)
    float *gx, *geta1, *geta2;             // same shapes as x, eta1, eta2
    long long cells;
    int S;
};

template <int D>
__global__ __launch_bounds__(256) void gauss_nat_bwd_kernel(GBArgs a) {
    constexpr int TRI = D * (D + 1) / 2;
    const int S = a.S;
    for (long long cell = (long long)blockIdx.x * blockDim.x + threadIdx.x; cell < a.cells; cell += (long long)gridDim.x * blockDim.x) {
        float P[TRI], Lm[TRI], e1[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            e1[i] = a.eta1[cell * D + i];
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                P[trd(i, j)] = -(a.eta2[(cell * D + i) * D + j] + a.eta2[(cell * D + j) * D + i]);
                Lm[trd(i, j)] = P[trd(i, j)];
            }
        }
        // Cholesky P = L L^T, diagonal stored as reciprocal
#pragma unroll
        for (int j = 0; j < D; ++j) {
            float s = Lm[trd(j, j)];
#pragma unroll
            for (int p = 0; p < j; ++p) s = fmaf(-Lm[trd(j, p)], Lm[trd(j, p)], s);
            const float rd = rsqrtf(s);
#pragma unroll
            for (int i = j + 1; i < D; ++i) {
                float tt = Lm[trd(i, j)];
#pragma unroll
                for (int p = 0; p < j; ++p) tt = fmaf(-Lm[trd(i, p)], Lm[trd(j, p)], tt);
                Lm[trd(i, j)] = tt * rd;
            }
            Lm[trd(j, j)] = rd;
        }
        // Y = L^-1 (lower);  Sigma = Y^T Y;  mu = Sigma eta1
        float Y[TRI];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            Y[trd(j, j)] = Lm[trd(j, j)];
#pragma unroll
            for (int i = j + 1; i < D; ++i) {
                float s2 = 0.f;
#pragma unroll
                for (int p = j; p < i; ++p) s2 = fmaf(Lm[trd(i, p)], Y[trd(p, j)], s2);
                Y[trd(i, j)] = -s2 * Lm[trd(i, i)];
            }
        }
        float Sg[TRI], mu[D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                float s2 = 0.f;
#pragma unroll
                for (int p = i; p < D; ++p) s2 = fmaf(Y[trd(p, i)], Y[trd(p, j)], s2);
                Sg[trd(i, j)] = s2;
            }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float s2 = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) s2 = fmaf(i >= j ? Sg[trd(i, j)] : Sg[trd(j, i)], e1[j], s2);
            mu[i] = s2;
        }
        float G = 0.f, sx[D], sxx[TRI];
#pragma unroll
        for (int i = 0; i < D; ++i) sx[i] = 0.f;
#pragma unroll
        for (int i = 0; i < TRI; ++i) sxx[i] = 0.f;
        for (int s = 0; s < S; ++s) {
            const float gs = a.g[cell * S + s];
            float xv[D];
#pragma unroll
            for (int i = 0; i < D; ++i) xv[i] = a.x[(cell * S + s) * D + i];
            G += gs;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                float px = 0.f;
#pragma unroll
                for (int j = 0; j < D; ++j) px = fmaf(i >= j ? P[trd(i, j)] : P[trd(j, i)], xv[j], px);
                a.gx[(cell * S + s) * D + i] = gs * (e1[i] - px);
                const float gxi = gs * xv[i];
                sx[i] += gxi;
#pragma unroll
                for (int j = 0; j <= i; ++j) sxx[trd(i, j)] = fmaf(gxi, xv[j], sxx[trd(i, j)]);
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            a.geta1[cell * D + i] = sx[i] - G * mu[i];
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                const float v = sxx[trd(i, j)] - G * (mu[i] * mu[j] + Sg[trd(i, j)]);
                a.geta2[(cell * D + i) * D + j] = v;
                a.geta2[(cell * D + j) * D + i] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Adjoint of the Student-t log-density (student_t.py:7-39) in the kernel's parametrisation out = cst_k - 1/2 (nu_k + D)
// log1p(|W_k (y - mu_k)|^2 / nu_k):  with z = W d, d = y - mu, c = (nu + D) / (nu + |z|^2)
//   d/dy = -g c W^T z;   d/dmu_k = -sum d/dy;   d/dW_k = -sum g c z d^T (lower);   d/dcst_k = sum g
// Block (bx, k) walks the (n, s) pairs of component k; per-block partial sums [D | TRI | 1] in a fixed order.
// ---------------------------------------------------------------------------------------------------------
struct TBArgs {
    const float *y, *mu, *W, *nu, *g;      // (N,K,S,D), (K,D), (K,D,D) lower, (K), (N,K,S)
    float* gy;                             // (N,K,S,D)
    float* part;                           // (gridDim.x, K, D + TRI + 1)
    long long N;
    int K, S;
};

template <int D>
__global__ __launch_bounds__(256) void student_t_bwd_kernel(TBArgs a) {
    constexpr int TRI = D * (D + 1) / 2, PWT = D + TRI + 1;
    __shared__ float red[4][PWT];
    const int k = blockIdx.y, S = a.S, K = a.K;
    float mu[D], W[TRI];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        mu[i] = a.mu[k * D + i];
#pragma unroll
        for (int j = 0; j <= i; ++j) W[trd(i, j)] = a.W[(k * D + i) * D + j];
    }
    const float nu = a.nu[k];
    float acc[PWT];
#pragma unroll
    for (int i = 0; i < PWT; ++i) acc[i] = 0.f;
    const long long tot = a.N * S;
    for (long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i0 < tot; i0 += (long long)gridDim.x * blockDim.x) {
        const long long n = i0 / S;
        const int s = (int)(i0 - n * S);
        const long long row = (n * K + k) * S + s;
        const float gs = a.g[row];
        float d[D], z[D];
#pragma unroll
        for (int i = 0; i < D; ++i) d[i] = a.y[row * D + i] - mu[i];
        float del2 = 0.f;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float zz = 0.f;
#pragma unroll
            for (int j = 0; j <= i; ++j) zz = fmaf(W[trd(i, j)], d[j], zz);
            z[i] = zz;
            del2 = fmaf(zz, zz, del2);
        }
        const float gc = -gs * (nu + (float)D) / (nu + del2);
        float gy[D];
#pragma unroll
        for (int j = 0; j < D; ++j) gy[j] = 0.f;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const float gz = gc * z[i];
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                gy[j] = fmaf(W[trd(i, j)], gz, gy[j]);
                acc[D + trd(i, j)] = fmaf(gz, d[j], acc[D + trd(i, j)]);
            }
        }
#pragma unroll
        for (int j = 0; j < D; ++j) { a.gy[row * D + j] = gy[j]; acc[j] -= gy[j]; }
        acc[D + TRI] += gs;
    }
    // block reduction: lanes by DPP-free shuffles in a fixed tree, waves in order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < PWT; ++i) {
        float v = acc[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < PWT) {
        const float v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        a.part[((long long)blockIdx.x * K + k) * PWT + threadIdx.x] = v;
    }
}

}  // namespace

extern "C" {

int vmp_gauss_logprob_nat_per_samp_bwd(const float* x, const float* eta1, const float* eta2, const float* g, int64_t N, int K,
                                       int S, int D, float* gx, float* geta1, float* geta2, void* stream) {
    int rc = chk(N, K, D, S);
    if (rc) return rc;
    if (!x || !eta1 || !eta2 || !g || !gx || !geta1 || !geta2) { set_error("vmp_gauss_logprob_nat_per_samp_bwd: null pointer"); return VMP_E_BADARG; }
    GBArgs a{x, eta1, eta2, g, gx, geta1, geta2, (long long)N * K, S};
    long long blocks = ((long long)N * K + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    rc = -1;
    VMP_DISPATCH_DD(D, {
        hipLaunchKernelGGL((gauss_nat_bwd_kernel<DD>), dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
        rc = check_launch("gauss_nat_bwd_kernel");
    });
    return rc;
}

int vmp_student_t_bwd_blocks(int64_t N, int S) {
    long long b = ((long long)N * S + 255) / 256;
    return (int)(b > 64 ? 64 : (b < 1 ? 1 : b));
}

int vmp_student_t_logprob_bwd(const float* y, const float* mu, const float* W, const float* nu, const float* g, int64_t N,
                              int K, int S, int D, float* gy, float* partials, void* stream) {
    int rc = chk(N, K, D, S);
    if (rc) return rc;
    if (!y || !mu || !W || !nu || !g || !gy || !partials) { set_error("vmp_student_t_logprob_bwd: null pointer"); return VMP_E_BADARG; }
    TBArgs a{y, mu, W, nu, g, gy, partials, (long long)N, K, S};
    rc = -1;
    VMP_DISPATCH_DD(D, {
        hipLaunchKernelGGL((student_t_bwd_kernel<DD>), dim3(vmp_student_t_bwd_blocks(N, S), K), dim3(256), 0, static_cast<hipStream_t>(stream), a);
        rc = check_launch("student_t_bwd_kernel");
    });
    return rc;
}

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
