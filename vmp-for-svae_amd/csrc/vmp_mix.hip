// T1: pure mixture VMP (GMM: reference models/gmm.py:25-269; SMM: models/smm.py:25-245) for gfx950.
//
// One streaming "pass" kernel does, per wave, tiles of 64 data rows (one row per lane):
//   E-part  (VALU, per lane):  q_nk = ||W_k (x_n - m_k)||^2  for all k,  softmax over k  -> r_nk (u_nk)
//   M-part  (MFMA, per wave):  sum_n w_nk * [1 | x_n | x_n x_n^T]  as a 16x16x4 fp32 MFMA GEMM whose inner
//                              index is the data row: A = w (K x rows), B = features (rows x F)
// Both parts exchange data through a per-wave LDS image laid out [row-of-values][64 lanes] with stride 66
// floats (66 = 2 mod 32 makes the MFMA operand reads (16 components x 4 rows per instruction) bank-conflict
// free, and lane-private accesses are conflict free by construction).  fp32 MFMA accumulators are flushed
// into fp64 registers after every tile; per-block fp64 partials go to the workspace and are reduced in a
// fixed order by the finalize kernel (deterministic, no atomics).
#include "vmp_common.h"

using namespace vmp;

namespace {

constexpr int TR = 64;        // data rows per wave tile
constexpr int LS = 66;        // LDS stride (floats) between value-rows
constexpr int MAX_NW = 8;     // waves per block
constexpr int MAX_BLOCKS = 512;

struct PassArgs {
    const float* x;
    const float* r_in;
    const float* u_in;
    const uint8_t* mask;
    const float* pack;
    float* r_out;
    float* u_out;
    float* logr_out;
    double* partials;
    long long N;
    long long ntiles;
    int K;
    int vec_ok;       // x / r pointers 16-byte aligned (vector path allowed)
};

template <int D>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&o)[D], bool vec) {
    if constexpr (D % 4 == 0) {
        if (vec) {
#pragma unroll
            for (int j = 0; j < D / 4; ++j) {
                float4 v = reinterpret_cast<const float4*>(p)[j];
                o[4 * j] = v.x; o[4 * j + 1] = v.y; o[4 * j + 2] = v.z; o[4 * j + 3] = v.w;
            }
            return;
        }
    } else if constexpr (D % 2 == 0) {
        if (vec) {
#pragma unroll
            for (int j = 0; j < D / 2; ++j) {
                float2 v = reinterpret_cast<const float2*>(p)[j];
                o[2 * j] = v.x; o[2 * j + 1] = v.y;
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) o[j] = p[j];
}

// LDS area [K][LS] (64 rows of a tile, transposed)  ->  global row-major chunk g[rows*K], coalesced.
__device__ __forceinline__ void tile_store(float* __restrict__ g, const float* area, int K, float invK, int rows,
                                           int lane, bool vec, bool as_log) {
    const int tot = rows * K;
    if (vec && (K & 3) == 0) {
        for (int q = 4 * lane; q < tot; q += 4 * WAVE) {
            const int n = (int)(((float)q + 0.5f) * invK);
            const int k = q - n * K;
            float4 v;
            v.x = area[(k + 0) * LS + n]; v.y = area[(k + 1) * LS + n];
            v.z = area[(k + 2) * LS + n]; v.w = area[(k + 3) * LS + n];
            if (as_log) { v.x = logf(v.x); v.y = logf(v.y); v.z = logf(v.z); v.w = logf(v.w); }
            *reinterpret_cast<float4*>(g + q) = v;
        }
    } else {
        for (int q = lane; q < tot; q += WAVE) {
            const int n = (int)(((float)q + 0.5f) * invK);
            const int k = q - n * K;
            float v = area[k * LS + n];
            g[q] = as_log ? logf(v) : v;
        }
    }
}

// global row-major chunk -> LDS area (transposed); rows beyond `rows` are zero-filled.
__device__ __forceinline__ void tile_load(const float* __restrict__ g, float* area, int K, float invK, int rows,
                                          int lane, bool vec) {
    const int tot = rows * K, full = TR * K;
    if (vec && (K & 3) == 0) {
        for (int q = 4 * lane; q < full; q += 4 * WAVE) {
            const int n = (int)(((float)q + 0.5f) * invK);
            const int k = q - n * K;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < tot) v = *reinterpret_cast<const float4*>(g + q);
            area[(k + 0) * LS + n] = v.x; area[(k + 1) * LS + n] = v.y;
            area[(k + 2) * LS + n] = v.z; area[(k + 3) * LS + n] = v.w;
        }
    } else {
        for (int q = lane; q < full; q += WAVE) {
            const int n = (int)(((float)q + 0.5f) * invK);
            const int k = q - n * K;
            area[k * LS + n] = (q < tot) ? g[q] : 0.f;
        }
    }
}

template <int D, int KT, int FLAV, bool ESTEP, bool STATS, bool MASK>
__global__ __launch_bounds__(MAX_NW * WAVE) void pass_kernel(PassArgs a) {
    using G = Geo<D>;
    constexpr int FT = G::FT;
    constexpr bool SMM = (FLAV == VMP_SMM);
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int K = a.K;
    const float invK = 1.0f / (float)K;
    const int nareas = SMM ? 3 : 1;
    const int wreg = (G::XROWS + nareas * K) * LS;
    float* xl = smem + wave * wreg;            // [XROWS][LS]: x columns, ones, zeros
    float* wl = xl + G::XROWS * LS;            // [K][LS]: logits -> e -> w (= r, or r*u for SMM)
    float* rl = wl + K * LS;                   // SMM: r
    float* ul = rl + K * LS;                   // SMM: u
    constexpr int ONE = D, ZERO = D + 1;
    xl[ONE * LS + lane] = 1.0f;
    xl[ZERO * LS + lane] = 0.0f;
    if (lane < LS - WAVE) { xl[ONE * LS + WAVE + lane] = 0.f; xl[ZERO * LS + WAVE + lane] = 0.f; }

    // ---- per-lane MFMA operand addressing: lane = (i16 = M/N index, kk = inner index = data row n0+kk)
    const int i16 = lane & 15, kk = lane >> 4;
    int offA[FT], offB[FT], offW[KT], offR[KT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
        const int f = ft * 16 + i16;
        int ra = ZERO, rb = ZERO;
        if (f == 0) { ra = ONE; rb = ONE; }
        else if (f <= D) { ra = f - 1; rb = ONE; }
        else if (f < G::F) {
            int p = f - D - 1, d = 0;
            while (p >= D - d) { p -= D - d; ++d; }
            ra = d; rb = d + p;
        }
        offA[ft] = ra * LS + kk;
        offB[ft] = rb * LS + kk;
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int k = kt * 16 + i16;
        offW[kt] = (k < K) ? (G::XROWS + k) * LS + kk : ZERO * LS + kk;
        offR[kt] = (k < K) ? (G::XROWS + K + k) * LS + kk : ZERO * LS + kk;
    }

    f32x4 acc[KT][FT];
    f32x4 nacc[KT];
    double dacc[KT][FT][4];
    double dn[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        nacc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) dn[kt][c] = 0.0;
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
            acc[kt][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) dacc[kt][ft][c] = 0.0;
        }
    }

    const long long tstride = (long long)gridDim.x * nw;
    const bool vec = a.vec_ok != 0;
    long long t = (long long)blockIdx.x * nw + wave;
    float xr[D];
    {
        const long long n = t * TR + lane;
#pragma unroll
        for (int j = 0; j < D; ++j) xr[j] = 0.f;
        if (t < a.ntiles && n < a.N) load_row<D>(a.x + n * D, xr, vec);
    }
    for (; t < a.ntiles; t += tstride) {
        const long long n = t * TR + lane;
        const bool valid = n < a.N;
        const long long rem = a.N - t * TR;
        const int rows = rem < TR ? (int)rem : TR;
        // prefetch the next tile's row while this one is processed
        float xn[D];
        {
            const long long n2 = (t + tstride) * TR + lane;
#pragma unroll
            for (int j = 0; j < D; ++j) xn[j] = 0.f;
            if (t + tstride < a.ntiles && n2 < a.N) load_row<D>(a.x + n2 * D, xn, vec);
        }

        if constexpr (ESTEP) {
            bool mk[D];
            if constexpr (MASK) {
#pragma unroll
                for (int j = 0; j < D; ++j) mk[j] = valid ? (a.mask[n * D + j] != 0) : false;
            }
            float mx = -INFINITY;
            for (int k = 0; k < K; ++k) {
                const float* __restrict__ p = a.pack + k * G::PACK;      // wave-uniform -> scalar loads
                float dv[D];
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    dv[j] = xr[j] - p[j];
                    if constexpr (MASK) dv[j] = mk[j] ? 0.f : dv[j];
                }
                float q = 0.f;
                int idx = D;
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    float y = 0.f;
#pragma unroll
                    for (int j = 0; j <= i; ++j) y = fmaf(p[idx++], dv[j], y);
                    q = fmaf(y, y, q);
                }
                const float lg = fmaf(-p[D + G::TRI + 1], q, p[D + G::TRI]);
                wl[k * LS + lane] = lg;
                mx = fmaxf(mx, lg);
                if constexpr (SMM) ul[k * LS + lane] = p[D + G::TRI + 2] / (q + p[D + G::TRI + 3]);
            }
            float s = 0.f;
            for (int k = 0; k < K; ++k) {
                const float e = __expf(wl[k * LS + lane] - mx);
                wl[k * LS + lane] = e;
                s += e;
            }
            const float inv = 1.0f / s;
            for (int k = 0; k < K; ++k) {
                const float r = wl[k * LS + lane] * inv;
                if constexpr (SMM) {
                    rl[k * LS + lane] = valid ? r : 0.f;
                    wl[k * LS + lane] = valid ? r * ul[k * LS + lane] : 0.f;
                } else {
                    wl[k * LS + lane] = valid ? r : 0.f;
                }
            }
            __builtin_amdgcn_wave_barrier();
            const float* rsrc = SMM ? rl : wl;
            tile_store(a.r_out + t * TR * K, rsrc, K, invK, rows, lane, vec, false);
            if (a.logr_out) tile_store(a.logr_out + t * TR * K, rsrc, K, invK, rows, lane, vec, true);
            if constexpr (SMM) tile_store(a.u_out + t * TR * K, ul, K, invK, rows, lane, vec, false);
        } else {
            // stats only: bring r (and u) in, coalesced, transposed into LDS
            if constexpr (SMM) {
                tile_load(a.r_in + t * TR * K, rl, K, invK, rows, lane, vec);
                tile_load(a.u_in + t * TR * K, ul, K, invK, rows, lane, vec);
                __builtin_amdgcn_wave_barrier();
                for (int k = 0; k < K; ++k) wl[k * LS + lane] = rl[k * LS + lane] * ul[k * LS + lane];
            } else {
                tile_load(a.r_in + t * TR * K, wl, K, invK, rows, lane, vec);
            }
        }

        if constexpr (STATS) {
#pragma unroll
            for (int j = 0; j < D; ++j) xl[j * LS + lane] = valid ? xr[j] : 0.f;
            __builtin_amdgcn_wave_barrier();
#pragma unroll 4
            for (int n0 = 0; n0 < TR; n0 += 4) {
                float b[FT];
#pragma unroll
                for (int ft = 0; ft < FT; ++ft) b[ft] = xl[offA[ft] + n0] * xl[offB[ft] + n0];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const float aw = xl[offW[kt] + n0];
#pragma unroll
                    for (int ft = 0; ft < FT; ++ft)
                        acc[kt][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, b[ft], acc[kt][ft], 0, 0, 0);
                    if constexpr (SMM) {
                        const float ar = xl[offR[kt] + n0];
                        nacc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar, b[0], nacc[kt], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if constexpr (SMM) dn[kt][c] += (double)nacc[kt][c];
#pragma unroll
                    for (int ft = 0; ft < FT; ++ft) dacc[kt][ft][c] += (double)acc[kt][ft][c];
                }
                nacc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ft = 0; ft < FT; ++ft) acc[kt][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int j = 0; j < D; ++j) xr[j] = xn[j];
    }

    if constexpr (STATS) {
        // ---- block reduction in fp64, waves in fixed order, then one partial per block
        __syncthreads();
        double* sc = reinterpret_cast<double*>(smem);          // [KT][FT+1][4][64]
        for (int w = 0; w < nw; ++w) {
            if (wave == w) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
#pragma unroll
                        for (int ft = 0; ft < FT; ++ft) {
                            const int idx = ((kt * (FT + 1) + ft) * 4 + c) * WAVE + lane;
                            sc[idx] = (w == 0 ? 0.0 : sc[idx]) + dacc[kt][ft][c];
                        }
                        const int idn = ((kt * (FT + 1) + FT) * 4 + c) * WAVE + lane;
                        sc[idn] = (w == 0 ? 0.0 : sc[idn]) + (SMM ? dn[kt][c] : dacc[kt][0][c]);
                    }
            }
            __syncthreads();
        }
        double* out = a.partials + (long long)blockIdx.x * K * G::PF;
        for (int e = threadIdx.x; e < KT * (FT + 1) * 4 * WAVE; e += blockDim.x) {
            const int l = e & 63, c = (e >> 6) & 3, tf = (e >> 8) % (FT + 1), kt = (e >> 8) / (FT + 1);
            const int k = kt * 16 + (l >> 4) * 4 + c;
            if (k >= K) continue;
            if (tf < FT) {
                const int f = tf * 16 + (l & 15);
                if (f < G::F) out[k * G::PF + f] = sc[e];
            } else if ((l & 15) == 0) {
                out[k * G::PF + G::F] = sc[e];                 // Nk = sum_n r_nk
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// K-sized posterior update (one block per component), fp64.
// ---------------------------------------------------------------------------------------------------------
__device__ double digamma_d(double x) {
    double r = 0.0;
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x
           - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760))))));
}

struct FinArgs {
    const double* partials;    // [nblk][K][PF]   (src == 0)
    const double* stats_in;    // [K][SW]         (src == 1)
    int nblk, K, flavour, src, do_post;
    const float *alpha0, *beta0, *m0, *C0, *v0, *kappa;
    float *alpha, *beta, *m, *C, *v, *xbar, *S, *pi, *pack;
    double* stats_out;
};

template <int D>
__device__ void write_pack(float* pack, int k, const double* m, const double* W /*lower, row-major full DxD*/,
                           double c, double h, double ua, double ub) {
    using G = Geo<D>;
    float* p = pack + k * G::PACK;
    int idx = D;
    for (int j = 0; j < D; ++j) p[j] = (float)m[j];
    for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) p[idx++] = (float)W[i * D + j];
    p[idx++] = (float)c; p[idx++] = (float)h; p[idx++] = (float)ua; p[idx++] = (float)ub;
}

// Cholesky of SPD A (DxD, row-major) -> lower L (in place, upper zeroed).  Returns false if not SPD.
template <int D>
__device__ bool chol_lower(double* A) {
    for (int j = 0; j < D; ++j) {
        double s = A[j * D + j];
        for (int p = 0; p < j; ++p) s -= A[j * D + p] * A[j * D + p];
        if (!(s > 0.0)) return false;
        const double d = sqrt(s);
        A[j * D + j] = d;
        for (int i = j + 1; i < D; ++i) {
            double t = A[i * D + j];
            for (int p = 0; p < j; ++p) t -= A[i * D + p] * A[j * D + p];
            A[i * D + j] = t / d;
        }
        for (int i = 0; i < j; ++i) A[i * D + j] = 0.0;
    }
    return true;
}

// inverse of lower-triangular L -> Li (lower)
template <int D>
__device__ void tri_inv_lower(const double* L, double* Li) {
    for (int i = 0; i < D * D; ++i) Li[i] = 0.0;
    for (int j = 0; j < D; ++j) {
        Li[j * D + j] = 1.0 / L[j * D + j];
        for (int i = j + 1; i < D; ++i) {
            double s = 0.0;
            for (int p = j; p < i; ++p) s += L[i * D + p] * Li[p * D + j];
            Li[i * D + j] = -s / L[i * D + i];
        }
    }
}

// Shared tail: expected log-dets, E log pi, constants and the pack, given C_k's Cholesky-derived W.
template <int D>
__device__ void estep_constants(int k, int flavour, double alpha_k, double alpha_sum, double beta_k, double v_k,
                                double logdetP, double kap, double& c, double& h, double& ua, double& ub, double& elp) {
    const double LOG2 = 0.69314718055994530942, PI = 3.14159265358979323846;
    elp = digamma_d(alpha_k) - digamma_d(alpha_sum);
    double sdg = 0.0;
    if (flavour == VMP_GMM) {
        for (int i = 0; i < D; ++i) sdg += digamma_d(0.5 * (v_k + 1.0 + i));      // gmm.py:128-129
        // gmm.py:120-121: log det P replaced by 0 when det P <= 1e-20
        const double ld = (logdetP > log(1e-20)) ? logdetP : 0.0;
        const double eld = sdg + D * LOG2 + ld;
        c = elp + 0.5 * eld - 0.5 * (D / beta_k);
        h = 0.5; ua = 1.0; ub = 1.0;
    } else {
        for (int i = 0; i < D; ++i) sdg += digamma_d(0.5 * (v_k + i));            // smm.py:107-108
        const double eld = sdg + D * LOG2 + logdetP;                                // smm.py:102 (no guard)
        h = 0.5 * (D + kap);
        // smm.py:122-124 (note the precedence of line 124: ... - (0.5 (D+kappa) m - log kappa))
        c = lgamma(0.5 * (D + kap)) - lgamma(0.5 * kap) - 0.5 * D * log(kap * PI) + elp + 0.5 * eld
            - h * (D / beta_k) + log(kap);
        ua = D + kap;                                                               // smm.py:134-137
        ub = D / beta_k + kap;
    }
}

template <int D>
__global__ __launch_bounds__(256) void finalize_kernel(FinArgs a) {
    using G = Geo<D>;
    __shared__ double part[4][G::PF];
    __shared__ double st[G::SW];           // canonical: Nk, Wk, sx[D], sxx[D*D]
    __shared__ double nall[VMP_MAX_K];
    const int k = blockIdx.x, tid = threadIdx.x, K = a.K;

    if (a.src == 0) {
        const int f = tid & 63, g = tid >> 6;
        if (f < G::PF) {
            double s = 0.0;
            for (int b = g; b < a.nblk; b += 4) s += a.partials[((long long)b * K + k) * G::PF + f];
            part[g][f] = s;
        }
        for (int j = tid; j < K; j += 256) {
            double s = 0.0;
            for (int b = 0; b < a.nblk; ++b) s += a.partials[((long long)b * K + j) * G::PF + G::F];
            nall[j] = s;
        }
        __syncthreads();
        if (tid < G::PF) part[0][tid] = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
        __syncthreads();
        if (tid == 0) {
            st[0] = part[0][G::F];
            st[1] = part[0][0];
            for (int d = 0; d < D; ++d) st[2 + d] = part[0][1 + d];
            int p = 1 + D;
            for (int d = 0; d < D; ++d)
                for (int e = d; e < D; ++e) {
                    st[2 + D + d * D + e] = part[0][p];
                    st[2 + D + e * D + d] = part[0][p];
                    ++p;
                }
        }
    } else {
        for (int i = tid; i < G::SW; i += 256) st[i] = a.stats_in[(long long)k * G::SW + i];
        for (int j = tid; j < K; j += 256) nall[j] = a.stats_in[(long long)j * G::SW];
    }
    __syncthreads();
    if (a.stats_out)
        for (int i = tid; i < G::SW; i += 256) a.stats_out[(long long)k * G::SW + i] = st[i];
    if (!a.do_post || tid != 0) return;

    // ---------------- single-thread fp64 posterior for component k ----------------
    const bool smm = a.flavour == VMP_SMM;
    const double Nk = st[0], Wk = smm ? st[1] : st[0];
    const double* sx = st + 2;
    const double* sxx = st + 2 + D;
    const double beta0 = a.beta0[k], v0 = a.v0[k], alpha0 = a.alpha0[k];
    double xb[D], Sk[D * D], mk[D], Ck[D * D];
    // x_k: gmm.py:30-36 (NaN -> un-normalised when N_k == 0);  smm.py:32-38 (eps = 1e-20)
    const double den = smm ? (Wk + 1e-20) : Wk;
    const bool empty = (!smm) && !(Wk != 0.0);
    for (int d = 0; d < D; ++d) xb[d] = empty ? sx[d] : sx[d] / den;
    // S_k = sum_n w (x - x_k)(x - x_k)^T / W_k  from raw moments (gmm.py:39-46, smm.py:41-50)
    for (int d = 0; d < D; ++d)
        for (int e = 0; e < D; ++e) {
            const double cen = sxx[d * D + e] - xb[d] * sx[e] - sx[d] * xb[e] + Wk * xb[d] * xb[e];
            Sk[d * D + e] = empty ? cen : cen / den;
        }
    const double alpha_k = alpha0 + Nk;                              // gmm.py:49-51 / smm.py:53-55
    const double beta_k = beta0 + Wk;                                // gmm.py:54-56 / smm.py:58-60
    const double v_k = smm ? (v0 + Nk) : (v0 + Nk + 1.0);            // smm.py:73-76 / gmm.py:79-81 (+1 quirk)
    for (int d = 0; d < D; ++d) mk[d] = (beta0 * a.m0[k * D + d] + Wk * xb[d]) / beta_k;     // gmm.py:59-68
    const double cf = beta0 * Wk / beta_k;
    for (int d = 0; d < D; ++d)
        for (int e = 0; e < D; ++e) {
            const double q0d = xb[d] - a.m0[k * D + d], q0e = xb[e] - a.m0[k * D + e];
            Ck[d * D + e] = a.C0[(k * D + d) * D + e] + Wk * Sk[d * D + e] + cf * q0d * q0e;  // gmm.py:71-76
        }
    double asum = 0.0;
    for (int j = 0; j < K; ++j) asum += a.alpha0[j] + nall[j];
    if (a.alpha) a.alpha[k] = (float)alpha_k;
    if (a.beta) a.beta[k] = (float)beta_k;
    if (a.v) a.v[k] = (float)v_k;
    for (int d = 0; d < D; ++d) {
        if (a.m) a.m[k * D + d] = (float)mk[d];
        if (a.xbar) a.xbar[k * D + d] = (float)xb[d];
        for (int e = 0; e < D; ++e) {
            if (a.C) a.C[(k * D + d) * D + e] = (float)Ck[d * D + e];
            if (a.S) a.S[(k * D + d) * D + e] = (float)Sk[d * D + e];
        }
    }
    // P_k = inv(C_k) (gmm.py:260) is never formed: with C = Lc Lc^T,
    //   v (x-m)^T P (x-m) = || sqrt(v) Lc^{-1} (x-m) ||^2   and   log det P = -2 sum log diag Lc.
    double Lc[D * D], Li[D * D];
    for (int i = 0; i < D * D; ++i) Lc[i] = 0.5 * (Ck[i] + Ck[(i % D) * D + i / D]);
    double c = 0, h = 0.5, ua = 1, ub = 1, elp = 0;
    if (chol_lower<D>(Lc)) {
        tri_inv_lower<D>(Lc, Li);
        double ld = 0.0;
        for (int i = 0; i < D; ++i) ld += log(Lc[i * D + i]);
        const double sv = sqrt(v_k);
        for (int i = 0; i < D * D; ++i) Li[i] *= sv;
        const double kap = smm ? (double)a.kappa[k] : 0.0;
        estep_constants<D>(k, a.flavour, alpha_k, asum, beta_k, v_k, -2.0 * ld, kap, c, h, ua, ub, elp);
    } else {
        for (int i = 0; i < D * D; ++i) Li[i] = nan("");
        c = nan("");
    }
    if (a.pi) a.pi[k] = (float)exp(elp);
    if (a.pack) write_pack<D>(a.pack, k, mk, Li, c, h, ua, ub);
}

// E-step pack from explicit (alpha, beta, m, P, v): gmm.e_step / smm.e_step signature.
struct PackArgs {
    int K, flavour;
    const float *alpha, *beta, *m, *P, *v, *kappa;
    float *pack, *pi;
};

template <int D>
__global__ void pack_kernel(PackArgs a) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.K) return;
    double asum = 0.0;
    for (int j = 0; j < a.K; ++j) asum += a.alpha[j];
    const double v_k = a.v[k];
    // v P = R R^T (lower R);  q = || R^T d ||^2.  The kernel wants a LOWER-triangular W with q = ||W d||^2:
    // factor the reversed matrix J P J = U_r U_r^T ... simpler: upper-Cholesky via reversal permutation.
    double A[D * D], mk[D];
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            const int ri = D - 1 - i, rj = D - 1 - j;               // reversal: A = J (sym P) J
            A[i * D + j] = 0.5 * ((double)a.P[(k * D + ri) * D + rj] + (double)a.P[(k * D + rj) * D + ri]);
        }
    double c = 0, h = 0.5, ua = 1, ub = 1, elp = 0;
    double W[D * D];
    if (chol_lower<D>(A)) {
        // A = L L^T  =>  P = (J L J)(J L J)^T with J L J upper-triangular U; q = d^T U U^T d = ||U^T d||^2,
        // and U^T = J L^T J is LOWER triangular.
        double ld = 0.0;
        for (int i = 0; i < D; ++i) ld += log(A[i * D + i]);
        const double sv = sqrt(v_k);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) W[i * D + j] = sv * A[(D - 1 - j) * D + (D - 1 - i)];
        const double kap = a.flavour == VMP_SMM ? (double)a.kappa[k] : 0.0;
        estep_constants<D>(k, a.flavour, a.alpha[k], asum, a.beta[k], v_k, 2.0 * ld, kap, c, h, ua, ub, elp);
    } else {
        for (int i = 0; i < D * D; ++i) W[i] = nan("");
        c = nan("");
    }
    for (int d = 0; d < D; ++d) mk[d] = a.m[k * D + d];
    if (a.pi) a.pi[k] = (float)exp(elp);
    write_pack<D>(a.pack, k, mk, W, c, h, ua, ub);
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
struct Plan {
    int nw, blocks;
    size_t lds;
    long long ntiles;
};

Plan make_plan(long long N, int D, int K, int flavour, bool stats) {
    Plan p;
    const int nareas = flavour == VMP_SMM ? 3 : 1;
    const size_t wreg = (size_t)(D + 2 + nareas * K) * LS * sizeof(float);
    int nw = (int)((60 * 1024) / wreg);
    if (nw > MAX_NW) nw = MAX_NW;
    if (nw < 1) nw = 1;
    p.ntiles = (N + TR - 1) / TR;
    if ((long long)nw > p.ntiles) nw = (int)p.ntiles;
    long long blocks = (p.ntiles + nw - 1) / nw;
    if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
    p.nw = nw;
    p.blocks = (int)blocks;
    const int FTn = (1 + D + D * (D + 1) / 2 + 15) / 16, KTn = (K + 15) / 16;
    const size_t scratch = stats ? (size_t)(KTn <= 2 ? KTn : 4) * (FTn + 1) * 4 * WAVE * sizeof(double) : 0;
    p.lds = wreg * nw;
    if (scratch > p.lds) p.lds = scratch;
    return p;
}

template <int D, int KT>
int launch_pass_dk(const PassArgs& a, const Plan& p, int flavour, bool estep, bool stats, bool mask, hipStream_t s) {
    dim3 grid(p.blocks), block(p.nw * WAVE);
#define VMP_LAUNCH(FL, E, S, M) \
    hipLaunchKernelGGL((pass_kernel<D, KT, FL, E, S, M>), grid, block, p.lds, s, a)
    if (flavour == VMP_GMM) {
        if (estep && stats) VMP_LAUNCH(VMP_GMM, true, true, false);
        else if (estep && mask) VMP_LAUNCH(VMP_GMM, true, false, true);
        else if (estep) VMP_LAUNCH(VMP_GMM, true, false, false);
        else VMP_LAUNCH(VMP_GMM, false, true, false);
    } else {
        if (estep && stats) VMP_LAUNCH(VMP_SMM, true, true, false);
        else if (estep) VMP_LAUNCH(VMP_SMM, true, false, false);
        else VMP_LAUNCH(VMP_SMM, false, true, false);
    }
#undef VMP_LAUNCH
    return check_launch("pass_kernel");
}

template <int D>
int launch_pass_d(const PassArgs& a, const Plan& p, int flavour, bool estep, bool stats, bool mask, hipStream_t s) {
    const int KT = (a.K + 15) / 16;
    if (!stats || KT == 1) return launch_pass_dk<D, 1>(a, p, flavour, estep, stats, mask, s);
    if (KT == 2) return launch_pass_dk<D, 2>(a, p, flavour, estep, stats, mask, s);
    return launch_pass_dk<D, 4>(a, p, flavour, estep, stats, mask, s);
}

#define VMP_DISPATCH_D(D, CALL)            \
    switch (D) {                            \
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

int check_dims(int64_t N, int D, int K) {
    if (N <= 0) { set_error("N must be positive (got %lld)", (long long)N); return VMP_E_BADARG; }
    if (D < 1 || D > VMP_MAX_D) { set_error("D=%d outside compiled range 1..%d", D, VMP_MAX_D); return VMP_E_DIM; }
    if (K < 1 || K > VMP_MAX_K) { set_error("K=%d outside compiled range 1..%d", K, VMP_MAX_K); return VMP_E_DIM; }
    return 0;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int run_pass(PassArgs a, int D, int flavour, bool estep, bool stats, bool mask, hipStream_t s) {
    Plan p = make_plan(a.N, D, a.K, flavour, stats);
    a.ntiles = p.ntiles;
    int rc = -1;
    VMP_DISPATCH_D(D, rc = launch_pass_d<DD>(a, p, flavour, estep, stats, mask, s));
    return rc;
}

int run_finalize(FinArgs f, int D, hipStream_t s) {
    int rc = -1;
    VMP_DISPATCH_D(D, {
        hipLaunchKernelGGL((finalize_kernel<DD>), dim3(f.K), dim3(256), 0, s, f);
        rc = check_launch("finalize_kernel");
    });
    return rc;
}

}  // namespace

// =========================================================================================================
// C ABI
// =========================================================================================================
extern "C" {

int vmp_mix_pack_words(int D) { return pack_words(D); }
int vmp_mix_stats_words(int D) { return stats_words(D); }

size_t vmp_mix_workspace_bytes(int64_t N, int D, int K) {
    (void)N;
    return (size_t)MAX_BLOCKS * K * partial_words(D) * sizeof(double);
}

int vmp_mix_stats(const float* x, const float* r, const float* u, int64_t N, int D, int K, double* stats, void* ws,
                  size_t ws_bytes, void* stream) {
    int rc = check_dims(N, D, K);
    if (rc) return rc;
    if (!x || !r || !stats || !ws) { set_error("vmp_mix_stats: null pointer"); return VMP_E_BADARG; }
    if (ws_bytes < vmp_mix_workspace_bytes(N, D, K)) { set_error("vmp_mix_stats: workspace too small"); return VMP_E_WS; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int flavour = u ? VMP_SMM : VMP_GMM;
    PassArgs a{};
    a.x = x; a.r_in = r; a.u_in = u; a.N = N; a.K = K; a.partials = static_cast<double*>(ws);
    a.vec_ok = aligned16(x) && aligned16(r) && (!u || aligned16(u));
    rc = run_pass(a, D, flavour, false, true, false, s);
    if (rc) return rc;
    FinArgs f{};
    f.partials = a.partials; f.nblk = make_plan(N, D, K, flavour, true).blocks; f.K = K; f.flavour = flavour;
    f.src = 0; f.do_post = 0; f.stats_out = stats;
    return run_finalize(f, D, s);
}

int vmp_mix_finalize(const double* stats, int D, int K, int flavour, const float* alpha0, const float* beta0,
                     const float* m0, const float* C0, const float* v0, const float* kappa, float* alpha, float* beta,
                     float* m, float* C, float* v, float* xbar, float* S, float* pi, float* pack, void* stream) {
    int rc = check_dims(1, D, K);
    if (rc) return rc;
    if (!stats || !alpha0 || !beta0 || !m0 || !C0 || !v0) { set_error("vmp_mix_finalize: null pointer"); return VMP_E_BADARG; }
    if (flavour == VMP_SMM && !kappa) { set_error("vmp_mix_finalize: SMM needs kappa"); return VMP_E_BADARG; }
    FinArgs f{};
    f.stats_in = stats; f.nblk = 0; f.K = K; f.flavour = flavour; f.src = 1; f.do_post = 1;
    f.alpha0 = alpha0; f.beta0 = beta0; f.m0 = m0; f.C0 = C0; f.v0 = v0; f.kappa = kappa;
    f.alpha = alpha; f.beta = beta; f.m = m; f.C = C; f.v = v; f.xbar = xbar; f.S = S; f.pi = pi; f.pack = pack;
    return run_finalize(f, D, static_cast<hipStream_t>(stream));
}

int vmp_mix_pack_from_params(int D, int K, int flavour, const float* alpha, const float* beta, const float* m,
                             const float* P, const float* v, const float* kappa, float* pack, float* pi, void* stream) {
    int rc = check_dims(1, D, K);
    if (rc) return rc;
    if (!alpha || !beta || !m || !P || !v || !pack) { set_error("vmp_mix_pack_from_params: null pointer"); return VMP_E_BADARG; }
    if (flavour == VMP_SMM && !kappa) { set_error("vmp_mix_pack_from_params: SMM needs kappa"); return VMP_E_BADARG; }
    PackArgs a{K, flavour, alpha, beta, m, P, v, kappa, pack, pi};
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = -1;
    VMP_DISPATCH_D(D, {
        hipLaunchKernelGGL((pack_kernel<DD>), dim3((K + 63) / 64), dim3(64), 0, s, a);
        rc = check_launch("pack_kernel");
    });
    return rc;
}

int vmp_mix_estep(const float* x, int64_t N, int D, int K, int flavour, const float* pack, const uint8_t* miss_mask,
                  float* r_out, float* u_out, float* logr_out, double* stats_out, void* ws, size_t ws_bytes,
                  void* stream) {
    int rc = check_dims(N, D, K);
    if (rc) return rc;
    if (!x || !pack || !r_out) { set_error("vmp_mix_estep: null pointer"); return VMP_E_BADARG; }
    if (flavour == VMP_SMM && !u_out) { set_error("vmp_mix_estep: SMM needs u_out"); return VMP_E_BADARG; }
    if (flavour != VMP_GMM && flavour != VMP_SMM) { set_error("vmp_mix_estep: bad flavour %d", flavour); return VMP_E_BADARG; }
    if (miss_mask && (flavour != VMP_GMM || stats_out)) {
        set_error("vmp_mix_estep: miss_mask only with GMM flavour and without fused stats");
        return VMP_E_BADARG;
    }
    if (stats_out && (!ws || ws_bytes < vmp_mix_workspace_bytes(N, D, K))) {
        set_error("vmp_mix_estep: workspace too small for fused stats");
        return VMP_E_WS;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    PassArgs a{};
    a.x = x; a.mask = miss_mask; a.pack = pack; a.r_out = r_out; a.u_out = u_out; a.logr_out = logr_out;
    a.N = N; a.K = K; a.partials = static_cast<double*>(ws);
    a.vec_ok = aligned16(x) && aligned16(r_out) && (!u_out || aligned16(u_out)) && (!logr_out || aligned16(logr_out));
    rc = run_pass(a, D, flavour, true, stats_out != nullptr, miss_mask != nullptr, s);
    if (rc || !stats_out) return rc;
    FinArgs f{};
    f.partials = a.partials; f.nblk = make_plan(N, D, K, flavour, true).blocks; f.K = K; f.flavour = flavour;
    f.src = 0; f.do_post = 0; f.stats_out = stats_out;
    return run_finalize(f, D, s);
}

int vmp_mix_estep_fused(const float* x, int64_t N, int D, int K, int flavour, const float* pack, float* r_out,
                        float* u_out, float* logr_out, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_dims(N, D, K);
    if (rc) return rc;
    if (!x || !pack || !r_out || !ws) { set_error("vmp_mix_estep_fused: null pointer"); return VMP_E_BADARG; }
    if (flavour != VMP_GMM && flavour != VMP_SMM) { set_error("vmp_mix_estep_fused: bad flavour %d", flavour); return VMP_E_BADARG; }
    if (flavour == VMP_SMM && !u_out) { set_error("vmp_mix_estep_fused: SMM needs u_out"); return VMP_E_BADARG; }
    if (ws_bytes < vmp_mix_workspace_bytes(N, D, K)) { set_error("vmp_mix_estep_fused: workspace too small"); return VMP_E_WS; }
    PassArgs a{};
    a.x = x; a.pack = pack; a.r_out = r_out; a.u_out = u_out; a.logr_out = logr_out;
    a.N = N; a.K = K; a.partials = static_cast<double*>(ws);
    a.vec_ok = aligned16(x) && aligned16(r_out) && (!u_out || aligned16(u_out)) && (!logr_out || aligned16(logr_out));
    return run_pass(a, D, flavour, true, true, false, static_cast<hipStream_t>(stream));
}

int vmp_mix_stats_ws(const float* x, const float* r, const float* u, int64_t N, int D, int K, void* ws,
                     size_t ws_bytes, void* stream) {
    int rc = check_dims(N, D, K);
    if (rc) return rc;
    if (!x || !r || !ws) { set_error("vmp_mix_stats_ws: null pointer"); return VMP_E_BADARG; }
    if (ws_bytes < vmp_mix_workspace_bytes(N, D, K)) { set_error("vmp_mix_stats_ws: workspace too small"); return VMP_E_WS; }
    PassArgs a{};
    a.x = x; a.r_in = r; a.u_in = u; a.N = N; a.K = K; a.partials = static_cast<double*>(ws);
    a.vec_ok = aligned16(x) && aligned16(r) && (!u || aligned16(u));
    return run_pass(a, D, u ? VMP_SMM : VMP_GMM, false, true, false, static_cast<hipStream_t>(stream));
}

int vmp_mix_finalize_ws(const void* ws, int64_t N, int D, int K, int flavour, const float* alpha0, const float* beta0,
                        const float* m0, const float* C0, const float* v0, const float* kappa, float* alpha,
                        float* beta, float* m, float* C, float* v, float* xbar, float* S, float* pi, float* pack,
                        double* stats_out, void* stream) {
    int rc = check_dims(N, D, K);
    if (rc) return rc;
    if (!ws || !alpha0 || !beta0 || !m0 || !C0 || !v0) { set_error("vmp_mix_finalize_ws: null pointer"); return VMP_E_BADARG; }
    if (flavour == VMP_SMM && !kappa) { set_error("vmp_mix_finalize_ws: SMM needs kappa"); return VMP_E_BADARG; }
    FinArgs f{};
    f.partials = static_cast<const double*>(ws);
    f.nblk = make_plan(N, D, K, flavour, true).blocks;
    f.K = K; f.flavour = flavour; f.src = 0; f.do_post = 1;
    f.alpha0 = alpha0; f.beta0 = beta0; f.m0 = m0; f.C0 = C0; f.v0 = v0; f.kappa = kappa;
    f.alpha = alpha; f.beta = beta; f.m = m; f.C = C; f.v = v; f.xbar = xbar; f.S = S; f.pi = pi; f.pack = pack;
    f.stats_out = stats_out;
    return run_finalize(f, D, static_cast<hipStream_t>(stream));
}

}  // extern "C"
