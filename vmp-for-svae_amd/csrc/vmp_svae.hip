// T2: the SVAE E-step (reference models/svae.py:14-119) fused with the regulariser of the ELBO
// (models/svae.py:229-252), forward and backward, for gfx950.
//
// Cell arithmetic (validated against the reference graph): for datapoint n, component k
//   Pt = diag(p_n) + P_k,  ht = h_n + h_k,  Lt = chol(Pt),  a = Lt^-1 ht,  ld = sum_i log Lt_ii
//   c_nk = bias_k + 1/2 |a|^2 - ld,            log_z_nk = c_nk - logsumexp_k c_nk         (svae.py:50-92)
//   x_nks = Lt^-T (a + eps_nks)                                                            (svae.py:95-119)
//   T'_nk = mean_s[ log N(x_s; phi~_nk) - log N(x_s; theta_k) - E log pi_k ]
//         = -(L/2) log 2pi + ld - 1/(2S) sum_s |eps_s|^2 + 1/(2S) sum_s |W_k (x_s - m_k)|^2 - kappa_k
//   (Student-t theta, svae.py:265-322 / student_t.py:31-37:  ... + 1/(2S) sum_s (nu_k + L) log1p(|W_k (x_s - m_k)|^2 / nu_k) - kappa_k)
// so that the reference's regulariser is sum_nk r_nk (T'_nk + log_z_nk) (svae.py:245-252).  One 8x8 Cholesky
// per cell replaces the reference's 7 LU + 4 Cholesky factorisations.
//
// Work decomposition: one (n,k) cell per lane.  A wave tile is RPT = 64/K whole rows (RPT*K lanes active) so the
// softmax over k and the per-row gradient sums never leave the wave; the lane's component k = lane % K is the
// same for every tile, so the component's parameters stay resident in its VGPRs for the whole kernel.  The
// noise tile eps (layout (cell, L, S)) is brought in with coalesced vector loads and staged in LDS with an odd
// cell stride (conflict-free per-lane reads); the samples x are written back into the same LDS slots and leave
// with coalesced stores in the reference's (cell, S, L) layout.
#include "vmp_common.h"
#include <stdlib.h>

using namespace vmp;

namespace {

constexpr int SV_NW = 4;            // waves per block (backward); forward: as many as the LDS noise tiles allow
constexpr int SV_FWD_MAX_NW = 8;
constexpr int SV_MAX_BLOCKS = 2048;
constexpr float LOG_2PI = 1.8378770664093454836f;

struct EFwdArgs {
    const float* eta1;      // (N,L)   encoder eta1
    const float* eta2d;     // (N,L)   encoder eta2 diagonal (negative)
    const float* hk;        // (K,L)   recognition eta1_k
    const float* Pk;        // (K,L,L) recognition precision (symmetric)
    const float* bias;      // (K)     B_k + log pi_k
    const float* noise;     // (N,K,L,S)
    const float* mk;        // (K,L)   E[mu_k] of theta
    const float* Wk;        // (K,L,L) lower-triangular W with W^T W = E[Sigma_k]^-1 (GMM) / Sigma_k^-1 (SMM)
    const float* kappa;     // (K)     the x-independent part of log p(x, z=k | theta)
    const float* nu;        // (K)     Student-t degrees of freedom, or NULL for the Gaussian theta
    float* x;               // (N,K,S,L)
    float* lz;              // (N,K)
    float* Tp;              // (N,K)
    long long N;
    int K, S, vec_ok;
};

template <int L>
struct SvGeo {
    static constexpr int TRI = L * (L + 1) / 2;
};

// lower-triangular packed index (row-major), i >= j
__host__ __device__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// Cholesky of the cell matrix (lower, packed).  On return Lm holds the factor with the DIAGONAL REPLACED BY ITS
// RECIPROCAL rd_j = 1/Lt_jj (every later use multiplies by it), and half_logdet = sum_j log Lt_jj.
template <int L>
__device__ __forceinline__ void cell_cholesky(float (&Lm)[SvGeo<L>::TRI], float& half_logdet) {
    float prod_log = 0.f;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        float s = Lm[tri(j, j)];
#pragma unroll
        for (int p = 0; p < j; ++p) s = fmaf(-Lm[tri(j, p)], Lm[tri(j, p)], s);
        const float rd = __builtin_amdgcn_rsqf(s);
        prod_log += __logf(s);
#pragma unroll
        for (int i = j + 1; i < L; ++i) {
            float t = Lm[tri(i, j)];
#pragma unroll
            for (int p = 0; p < j; ++p) t = fmaf(-Lm[tri(i, p)], Lm[tri(j, p)], t);
            Lm[tri(i, j)] = t * rd;
        }
        Lm[tri(j, j)] = rd;
    }
    half_logdet = 0.5f * prod_log;
}

// v <- Lt^-1 v   (forward substitution; diagonal of Lm holds reciprocals)
template <int L>
__device__ __forceinline__ void solve_lower(const float (&Lm)[SvGeo<L>::TRI], float (&v)[L]) {
#pragma unroll
    for (int i = 0; i < L; ++i) {
        float t = v[i];
#pragma unroll
        for (int p = 0; p < i; ++p) t = fmaf(-Lm[tri(i, p)], v[p], t);
        v[i] = t * Lm[tri(i, i)];
    }
}

// v <- Lt^-T v   (back substitution)
template <int L>
__device__ __forceinline__ void solve_lower_t(const float (&Lm)[SvGeo<L>::TRI], float (&v)[L]) {
#pragma unroll
    for (int i = L - 1; i >= 0; --i) {
        float t = v[i];
#pragma unroll
        for (int p = i + 1; p < L; ++p) t = fmaf(-Lm[tri(p, i)], v[p], t);
        v[i] = t * Lm[tri(i, i)];
    }
}

// sum / max over the K lanes of this lane's row, through a 64-float LDS scratch
__device__ __forceinline__ float row_sum(float v, float* scr, int lane, int rbase, int K) {
    scr[lane] = v;
    __builtin_amdgcn_wave_barrier();
    float s = 0.f;
    for (int j = 0; j < K; ++j) s += scr[rbase + j];
    __builtin_amdgcn_wave_barrier();
    return s;
}
__device__ __forceinline__ float row_max(float v, float* scr, int lane, int rbase, int K) {
    scr[lane] = v;
    __builtin_amdgcn_wave_barrier();
    float m = -INFINITY;
    for (int j = 0; j < K; ++j) m = fmaxf(m, scr[rbase + j]);
    __builtin_amdgcn_wave_barrier();
    return m;
}

template <int L>
__global__ __launch_bounds__(SV_FWD_MAX_NW * WAVE) void svae_estep_fwd_kernel(EFwdArgs a) {
    constexpr int TRI = SvGeo<L>::TRI;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int K = a.K, S = a.S;
    const int LSn = L * S, CSTR = LSn | 1;                 // odd LDS stride per cell
    const int RPT = WAVE / K, CT = RPT * K;                // rows / cells per wave tile
    float* et = smem + wave * (WAVE * CSTR + WAVE);        // noise tile, later the samples
    float* scr = et + WAVE * CSTR;
    const bool lane_on = lane < CT;
    const int r = lane / K, k = lane - r * K, rbase = lane_on ? r * K : 0;

    // resident component parameters
    float Pl[TRI], hkk[L], mkk[L], Wt[TRI];                // Wt: lower triangle of W_k, packed
    float biask = 0.f, kappak = 0.f, nuk = 0.f;
    const bool student = a.nu != nullptr;
    // unconditional loads from clamped indices + value selects (a load under a per-element condition costs a
    // branch and a full wait per element)
    const int kc = lane_on ? k : 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const float hv = a.hk[kc * L + i], mv = a.mk[kc * L + i];
        hkk[i] = lane_on ? hv : 0.f;
        mkk[i] = lane_on ? mv : 0.f;
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const float pvv = a.Pk[(kc * L + i) * L + j], wv = a.Wk[(kc * L + i) * L + j];
            Pl[tri(i, j)] = lane_on ? pvv : (i == j ? 1.f : 0.f);
            Wt[tri(i, j)] = lane_on ? wv : 0.f;
        }
    }
    {
        const float bv = a.bias[kc], kv = a.kappa[kc], nv = *(student ? a.nu + kc : a.bias);
        biask = lane_on ? bv : 0.f; kappak = lane_on ? kv : 0.f; nuk = (student && lane_on) ? nv : 1.f;
    }
    const float inv_nu = 1.0f / nuk;

    const long long ntiles = (a.N + RPT - 1) / RPT;
    const float invLS = 1.0f / (float)LSn, invS = 1.0f / (float)S;
    for (long long t = (long long)blockIdx.x * nw + wave; t < ntiles; t += (long long)gridDim.x * nw) {
        const long long row = t * RPT + r;
        const bool on = lane_on && row < a.N;
        const long long rows_here = (a.N - t * RPT) < RPT ? (a.N - t * RPT) : RPT;
        const int tot = (int)rows_here * K * LSn;          // floats of noise / samples in this tile
        // ---- stage the noise tile: coalesced global reads, padded per-cell layout in LDS
        {
            const float* __restrict__ g = a.noise + t * CT * LSn;
            if (a.vec_ok && (LSn & 3) == 0) {
                // 5 vector loads in flight per lane before the first LDS write (a one-load-at-a-time copy loop
                // exposes the full memory latency 20 times per tile)
                constexpr int UN = 10;
                for (int e0 = 4 * lane; e0 < tot; e0 += UN * 4 * WAVE) {
                    float4 v[UN];
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        const int e = e0 + u * 4 * WAVE;
                        v[u] = (e < tot) ? *reinterpret_cast<const float4*>(g + e) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        const int e = e0 + u * 4 * WAVE;
                        if (e < tot) {
                            const int c = (int)(((float)e + 0.5f) * invLS);
                            const int j = e - c * LSn;
                            float* d = et + c * CSTR + j;
                            d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                        }
                    }
                }
            } else {
                for (int e = lane; e < tot; e += WAVE) {
                    const int c = (int)(((float)e + 0.5f) * invLS);
                    et[c * CSTR + (e - c * LSn)] = g[e];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- cell factorisation
        float Lm[TRI], av[L];
#pragma unroll
        for (int i = 0; i < TRI; ++i) Lm[i] = Pl[i];
        const long long rowc = on ? row : 0;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const float e1v = a.eta1[rowc * L + i], e2v = a.eta2d[rowc * L + i];
            const float e1 = on ? e1v : 0.f;
            const float e2 = on ? e2v : -0.5f;
            Lm[tri(i, i)] = fmaf(-2.f, e2, Lm[tri(i, i)]);
            av[i] = e1 + hkk[i];
        }
        float ld;
        cell_cholesky<L>(Lm, ld);
        solve_lower<L>(Lm, av);
        float aa = 0.f;
#pragma unroll
        for (int i = 0; i < L; ++i) aa = fmaf(av[i], av[i], aa);
        const float c = on ? (biask + 0.5f * aa - ld) : -INFINITY;
        const float mx = row_max(c, scr, lane, rbase, K);
        const float ex = on ? __expf(c - mx) : 0.f;
        const float se = row_sum(ex, scr, lane, rbase, K);
        const float lz = c - mx - __logf(se);

        // ---- samples and the per-cell regulariser term
        float eps2 = 0.f, qth = 0.f;
        float* cell = et + lane * CSTR;
        for (int s = 0; s < S; ++s) {
            float z[L];
#pragma unroll
            for (int i = 0; i < L; ++i) {
                const float e = lane_on ? cell[i * S + s] : 0.f;
                eps2 = fmaf(e, e, eps2);
                z[i] = av[i] + e;
            }
            solve_lower_t<L>(Lm, z);                       // z is now x_s
            float d[L];
#pragma unroll
            for (int i = 0; i < L; ++i) d[i] = z[i] - mkk[i];
            float del2 = 0.f;
#pragma unroll
            for (int i = 0; i < L; ++i) {
                float y = 0.f;
#pragma unroll
                for (int j = 0; j <= i; ++j) y = fmaf(Wt[tri(i, j)], d[j], y);
                del2 = fmaf(y, y, del2);
            }
            qth += student ? (nuk + (float)L) * log1pf(del2 * inv_nu) : del2;
            if (lane_on) {
#pragma unroll
                for (int i = 0; i < L; ++i) cell[i * S + s] = z[i];
            }
        }
        if (on) {
            a.lz[row * K + k] = lz;
            a.Tp[row * K + k] = -0.5f * L * LOG_2PI + ld - 0.5f * invS * eps2 + 0.5f * invS * qth - kappak;
        }
        __builtin_amdgcn_wave_barrier();

        // ---- samples out: (cell, S, L) layout, coalesced
        {
            float* __restrict__ g = a.x + t * CT * LSn;
            if (a.vec_ok && (L & 3) == 0) {
                for (int o = 4 * lane; o < tot; o += 4 * WAVE) {
                    const int c2 = (int)(((float)o + 0.5f) * invLS);
                    const int rem = o - c2 * LSn;
                    const int s = rem / L, l = rem - s * L;
                    const float* src = et + c2 * CSTR + l * S + s;
                    float4 v;
                    v.x = src[0]; v.y = src[S]; v.z = src[2 * S]; v.w = src[3 * S];
                    *reinterpret_cast<float4*>(g + o) = v;
                }
            } else {
                for (int o = lane; o < tot; o += WAVE) {
                    const int c2 = (int)(((float)o + 0.5f) * invLS);
                    const int rem = o - c2 * LSn;
                    const int s = rem / L, l = rem - s * L;
                    g[o] = et[c2 * CSTR + l * S + s];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}


// ---------------------------------------------------------------------------------------------------------
// Forward, sample-parallel form (the fast path): lane = (cell, pair of samples).  The SP = ceil(S/2) lanes of a
// cell each repeat the cell's small factorisation (cheap, and it keeps every lane busy) and then own two samples,
// so that a lane's dependent chain is one Cholesky + two back-substitutions instead of one Cholesky + S of them,
// and there are SP times more independent chains in flight.  A block covers RPB whole rows (RPB*K cells): the
// softmax over k and the per-cell sums of the regulariser go through LDS.  Noise is read straight from global
// memory (issued before the factorisation, so its latency hides behind it); a cell's SP lanes write its S samples
// as one contiguous (S x L) block, i.e. the wave's stores are fully coalesced without staging.
// ---------------------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(512) void svae_estep_fwd2_kernel(EFwdArgs a, int RPB, int SP) {
    constexpr int TRI = SvGeo<L>::TRI;
    constexpr int PSTR = TRI | 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = a.K, S = a.S;
    const int CB = RPB * K;                                 // cells per block tile
    float* pk_lds = smem;                                   // [K][PSTR]
    float* c_lds = smem + K * PSTR;                         // [CB]   c_nk of the tile
    float* e_lds = c_lds + CB;                              // [CB*SP] eps^2 partial sums
    float* q_lds = e_lds + CB * SP;                         // [CB*SP] theta-term partial sums
    const int tid = threadIdx.x;
    const int cib = tid / SP, pair = tid - cib * SP;        // cell in block, sample pair
    const bool lane_on = cib < CB;
    const int r = lane_on ? cib / K : 0, k = lane_on ? cib - r * K : 0;
    const int s0 = 2 * pair;
    const bool two = s0 + 1 < S;

    for (int e = tid; e < K * TRI; e += blockDim.x) {
        const int kk = e / TRI, idx = e - kk * TRI;
        int i = 0;
        while (tri(i + 1, 0) <= idx) ++i;
        pk_lds[kk * PSTR + idx] = a.Pk[(kk * L + i) * L + (idx - tri(i, 0))];
    }
    float hkk[L], mkk[L], Wt[TRI];
    const bool student = a.nu != nullptr;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        hkk[i] = lane_on ? a.hk[k * L + i] : 0.f;
        mkk[i] = lane_on ? a.mk[k * L + i] : 0.f;
#pragma unroll
        for (int j = 0; j <= i; ++j) Wt[tri(i, j)] = lane_on ? a.Wk[(k * L + i) * L + j] : 0.f;
    }
    const float biask = lane_on ? a.bias[k] : 0.f, kappak = lane_on ? a.kappa[k] : 0.f;
    const float nuk = (student && lane_on) ? a.nu[k] : 1.f, inv_nu = 1.0f / nuk;
    const float invS = 1.0f / (float)S;
    __syncthreads();

    const long long ntiles = (a.N + RPB - 1) / RPB;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long row = t * RPB + r;
        const bool on = lane_on && row < a.N;
        const long long cell = row * K + k;
        // noise for this lane's two samples: issued first, consumed after the factorisation
        float e0[L], e1[L];
        {
            const float* __restrict__ nz = a.noise + cell * (L * S) + s0;
#pragma unroll
            for (int i = 0; i < L; ++i) {
                e0[i] = on ? nz[i * S] : 0.f;
                e1[i] = (on && two) ? nz[i * S + 1] : 0.f;
            }
        }
        float Lm[TRI], av[L];
#pragma unroll
        for (int i = 0; i < TRI; ++i) Lm[i] = lane_on ? pk_lds[k * PSTR + i] : 0.f;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const float x1 = on ? a.eta1[row * L + i] : 0.f;
            const float x2 = on ? a.eta2d[row * L + i] : -0.5f;
            Lm[tri(i, i)] = fmaf(-2.f, x2, Lm[tri(i, i)]);
            av[i] = x1 + hkk[i];
        }
        float ld;
        cell_cholesky<L>(Lm, ld);
        solve_lower<L>(Lm, av);
        float aa = 0.f;
#pragma unroll
        for (int i = 0; i < L; ++i) aa = fmaf(av[i], av[i], aa);
        const float c = on ? (biask + 0.5f * aa - ld) : -INFINITY;
        if (lane_on && pair == 0) c_lds[cib] = c;

        // two samples: z = a + eps, x = Lt^-T z (two independent chains), theta term
        float z0[L], z1[L];
        float eps2 = 0.f;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            eps2 = fmaf(e0[i], e0[i], fmaf(e1[i], e1[i], eps2));
            z0[i] = av[i] + e0[i];
            z1[i] = av[i] + e1[i];
        }
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
            float t0 = z0[i], t1 = z1[i];
#pragma unroll
            for (int p = i + 1; p < L; ++p) {
                t0 = fmaf(-Lm[tri(p, i)], z0[p], t0);
                t1 = fmaf(-Lm[tri(p, i)], z1[p], t1);
            }
            z0[i] = t0 * Lm[tri(i, i)];
            z1[i] = t1 * Lm[tri(i, i)];
        }
        if (on) {
            float* __restrict__ xo = a.x + (cell * S + s0) * L;
            if ((L & 3) == 0 && a.vec_ok) {
#pragma unroll
                for (int q = 0; q < L / 4; ++q) reinterpret_cast<float4*>(xo)[q] = make_float4(z0[4 * q], z0[4 * q + 1], z0[4 * q + 2], z0[4 * q + 3]);
                if (two) {
#pragma unroll
                    for (int q = 0; q < L / 4; ++q) reinterpret_cast<float4*>(xo + L)[q] = make_float4(z1[4 * q], z1[4 * q + 1], z1[4 * q + 2], z1[4 * q + 3]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < L; ++i) { xo[i] = z0[i]; if (two) xo[L + i] = z1[i]; }
            }
        }
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int i = 0; i < L; ++i) { z0[i] -= mkk[i]; z1[i] -= mkk[i]; }
#pragma unroll
        for (int i = 0; i < L; ++i) {
            float y0 = 0.f, y1 = 0.f;
#pragma unroll
            for (int j = 0; j <= i; ++j) { y0 = fmaf(Wt[tri(i, j)], z0[j], y0); y1 = fmaf(Wt[tri(i, j)], z1[j], y1); }
            d0 = fmaf(y0, y0, d0);
            d1 = fmaf(y1, y1, d1);
        }
        float qth;
        if (student) qth = (nuk + (float)L) * (log1pf(d0 * inv_nu) + (two ? log1pf(d1 * inv_nu) : 0.f));
        else qth = d0 + (two ? d1 : 0.f);
        if (lane_on) { e_lds[tid] = eps2; q_lds[tid] = qth; }
        __syncthreads();
        if (on && pair == 0) {
            float mx = -INFINITY;
            for (int j = 0; j < K; ++j) mx = fmaxf(mx, c_lds[r * K + j]);
            float se = 0.f;
            for (int j = 0; j < K; ++j) se += __expf(c_lds[r * K + j] - mx);
            float es = 0.f, qs = 0.f;
            for (int p = 0; p < SP; ++p) { es += e_lds[tid + p]; qs += q_lds[tid + p]; }
            a.lz[cell] = c - mx - __logf(se);
            a.Tp[cell] = -0.5f * L * LOG_2PI + ld - 0.5f * invS * es + 0.5f * invS * qs - kappak;
        }
        __syncthreads();
    }
}

// =========================================================================================================
// backward
// =========================================================================================================
struct EBwdArgs {
    const float* eta1;
    const float* eta2d;
    const float* hk;
    const float* Pk;
    const float* bias;
    const float* mk;
    const float* Wk;
    const float* nu;        // (K) or NULL
    const float* x;         // (N,K,S,L) samples from the forward pass
    const float* lz;        // (N,K)
    const float* Gx;        // (N,K,S,L) dLoss/dx  (from the decoder)
    const float* Glz;       // (N,K)     dLoss/dlog_z
    const float* GT;        // (N,K)     dLoss/dT'
    float* g_eta1;          // (N,L)
    float* g_eta2d;         // (N,L)
    float* partials;        // (nblk, K, 2(L+TRI+1)): g_hk | g_Pk (lower, symmetric gradient) | g_bias | g_mk | g_Wk (lower) | g_kappa
    long long N;
    int K, S, vec_ok;
};

template <int L>
__global__ __launch_bounds__(SV_NW * WAVE) void svae_estep_bwd_kernel(EBwdArgs a) {
    constexpr int TRI = SvGeo<L>::TRI;
    constexpr int PW = 2 * (L + TRI + 1);
    constexpr int TH = L + TRI + 1;                         // offset of the theta-side sums
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int K = a.K, S = a.S;
    const int LSn = L * S;
    const int RPT = WAVE / K, CT = RPT * K;
    const int PSTR = TRI | 1;
    float* pk_lds = smem;                                   // [K][PSTR]  lower triangle of P_k
    float* scr = smem + K * PSTR + wave * WAVE;
    const int PWa = (a.nu != nullptr) ? PW : TH;            // accumulator rows in use (theta-side sums only for Student-t)
    float* red = smem + K * PSTR + nw * WAVE;               // block reduction scratch [PWa][64]
    float* accl = red + PWa * WAVE + wave * (PWa * WAVE);   // this wave's accumulators [PWa][64], lane-private columns
    const bool lane_on = lane < CT;
    const int r = lane / K, k = lane - r * K, rbase = lane_on ? r * K : 0;

    for (int e = threadIdx.x; e < K * TRI; e += blockDim.x) {
        const int kk = e / TRI, idx = e - kk * TRI;
        int i = 0;
        while (tri(i + 1, 0) <= idx) ++i;
        const int j = idx - tri(i, 0);
        pk_lds[kk * PSTR + idx] = a.Pk[(kk * L + i) * L + j];
    }
    __syncthreads();

    float hkk[L], mkk[L], Wt[TRI];
    const bool student = a.nu != nullptr;
    const int kc = lane_on ? k : 0;                         // unconditional loads from clamped indices + selects
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const float hv = a.hk[kc * L + i], mv = a.mk[kc * L + i];
        hkk[i] = lane_on ? hv : 0.f;
        mkk[i] = lane_on ? mv : 0.f;
#pragma unroll
        for (int j = 0; j <= i; ++j) { const float wv = a.Wk[(kc * L + i) * L + j]; Wt[tri(i, j)] = lane_on ? wv : 0.f; }
    }
    const float nuv = *(student ? a.nu + kc : a.bias);
    const float nuk = (student && lane_on) ? nuv : 1.f;
    for (int i = 0; i < PWa; ++i) accl[i * WAVE + lane] = 0.f;  // sums over this lane's cells (fixed k)

    const long long ntiles = (a.N + RPT - 1) / RPT;
    const float invS = 1.0f / (float)S;
    for (long long t = (long long)blockIdx.x * nw + wave; t < ntiles; t += (long long)gridDim.x * nw) {
        const long long row = t * RPT + r;
        const bool on = lane_on && row < a.N;
        const long long rowc = on ? row : 0;
        const long long cellid = rowc * K + kc;             // clamped: always a valid cell

        float Lm[TRI], av[L], mu[L];
#pragma unroll
        for (int i = 0; i < TRI; ++i) Lm[i] = lane_on ? pk_lds[k * PSTR + i] : 0.f;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const float e1v = a.eta1[rowc * L + i], e2v = a.eta2d[rowc * L + i];
            const float e1 = on ? e1v : 0.f;
            const float e2 = on ? e2v : -0.5f;
            Lm[tri(i, i)] = fmaf(-2.f, e2, lane_on ? Lm[tri(i, i)] : 0.f);
            av[i] = e1 + hkk[i];
        }
        float ld;
        cell_cholesky<L>(Lm, ld);
        solve_lower<L>(Lm, av);
#pragma unroll
        for (int i = 0; i < L; ++i) mu[i] = av[i];
        solve_lower_t<L>(Lm, mu);                           // mu~ = Pt^-1 ht

        const float glzv = a.Glz[cellid], gTv = a.GT[cellid], lzv = a.lz[cellid];
        const float glz = on ? glzv : 0.f;
        const float gT = on ? gTv : 0.f;
        const float rnk = on ? __expf(lzv) : 0.f;
        const float gsum = row_sum(glz, scr, lane, rbase, K);
        const float Gc = glz - rnk * gsum;                  // through the log-sum-exp normalisation
        const float Gld = gT - Gc;                          // T' has +ld, c has -ld

        float Wsum[L], M[TRI];
#pragma unroll
        for (int i = 0; i < L; ++i) Wsum[i] = 0.f;
#pragma unroll
        for (int i = 0; i < TRI; ++i) M[i] = 0.f;
        const float gts = gT * invS;
        const float* __restrict__ xc = a.x + cellid * LSn;
        const float* __restrict__ gc = a.Gx + cellid * LSn;
        float nxs[L], ngx[L];                               // next sample's rows, in flight while this one is used
        // (one branch around the whole group of row loads: skipping the tail prefetches matters, this kernel is
        //  bound by the per-lane row loads - an unconditional clamped version measured 27% slower)
        auto load_rows = [&](int s2, float (&xo)[L], float (&go)[L]) {
            const bool live = on && s2 < S;
            if (live) {
                if ((L & 3) == 0 && a.vec_ok) {
#pragma unroll
                    for (int q = 0; q < L / 4; ++q) {
                        const float4 v = reinterpret_cast<const float4*>(xc + s2 * L)[q];
                        const float4 w = reinterpret_cast<const float4*>(gc + s2 * L)[q];
                        xo[4 * q] = v.x; xo[4 * q + 1] = v.y; xo[4 * q + 2] = v.z; xo[4 * q + 3] = v.w;
                        go[4 * q] = w.x; go[4 * q + 1] = w.y; go[4 * q + 2] = w.z; go[4 * q + 3] = w.w;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < L; ++i) { xo[i] = xc[s2 * L + i]; go[i] = gc[s2 * L + i]; }
                }
            }
#pragma unroll
            for (int i = 0; i < L; ++i) { xo[i] = live ? xo[i] : 0.f; go[i] = live ? go[i] : 0.f; }
        };
        float nxs2[L], ngx2[L];                             // ... and the one after it
        load_rows(0, nxs, ngx);
        load_rows(1, nxs2, ngx2);
        for (int s = 0; s < S; ++s) {
            float xs[L], gx[L];
#pragma unroll
            for (int i = 0; i < L; ++i) { xs[i] = nxs[i]; gx[i] = ngx[i]; nxs[i] = nxs2[i]; ngx[i] = ngx2[i]; }
            load_rows(s + 2, nxs2, ngx2);
            // d/dx of the theta term of T':  (1/S) c_s W^T W (x - m),  c_s = 1 (Gaussian) or (nu+L)/(nu+delta^2)
            float d[L], y[L];
#pragma unroll
            for (int i = 0; i < L; ++i) d[i] = xs[i] - mkk[i];
            float del2 = 0.f;
#pragma unroll
            for (int i = 0; i < L; ++i) {
                float yy = 0.f;
#pragma unroll
                for (int j = 0; j <= i; ++j) yy = fmaf(Wt[tri(i, j)], d[j], yy);
                y[i] = yy;
                del2 = fmaf(yy, yy, del2);
            }
            const float gc = student ? gts * (nuk + (float)L) / (nuk + del2) : gts;
#pragma unroll
            for (int i = 0; i < L; ++i) {
                const float gy = gc * y[i];
#pragma unroll
                for (int j = 0; j <= i; ++j) gx[j] = fmaf(Wt[tri(i, j)], gy, gx[j]);
            }
            if (student && on) {
                // theta is trainable in the SMM model: d/dm = -(the x-gradient of the theta term), d/dW = c y d^T
                float tx[L];
#pragma unroll
                for (int j = 0; j < L; ++j) tx[j] = 0.f;
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    const float gy = gc * y[i];
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        tx[j] = fmaf(Wt[tri(i, j)], gy, tx[j]);
                        accl[(TH + L + tri(i, j)) * WAVE + lane] += gy * d[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < L; ++j) accl[(TH + j) * WAVE + lane] -= tx[j];
            }
            solve_lower<L>(Lm, gx);                         // w_s = Lt^-1 gx_s
#pragma unroll
            for (int i = 0; i < L; ++i) {
                Wsum[i] += gx[i];
                const float e = xs[i] - mu[i];              // e_s = Lt^-T eps_s
#pragma unroll
                for (int j = 0; j <= i; ++j) M[tri(i, j)] = fmaf(e, gx[j], M[tri(i, j)]);
            }
        }
        // ---- assemble dLoss/dht and dLoss/dPt (symmetric, lower triangle)
        float V[L];
#pragma unroll
        for (int i = 0; i < L; ++i) V[i] = Wsum[i];
        solve_lower_t<L>(Lm, V);                            // V = Pt^-1 sum_s gx_s
        float gh[L];
#pragma unroll
        for (int i = 0; i < L; ++i) gh[i] = fmaf(Gc, mu[i], V[i]);

        // Cholesky adjoint with G_L = -tril(M):  A = Lt^T G_L (lower part), B = Phi(A), C = B + B^T + Gld I
        // real diagonal of Lt is 1/Lm[tri(j,j)]
        float Cs[TRI];                                      // symmetric C, lower triangle
        float dg[L];
#pragma unroll
        for (int i = 0; i < L; ++i) dg[i] = 1.0f / Lm[tri(i, i)];
#pragma unroll
        for (int i = 0; i < L; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                float s2 = 0.f;
#pragma unroll
                for (int p = i; p < L; ++p) {
                    const float lpi = (p == i) ? dg[i] : Lm[tri(p, i)];
                    s2 = fmaf(lpi, -M[tri(p, j)], s2);
                }
                Cs[tri(i, j)] = (i == j) ? (s2 + Gld) : s2;   // Phi halves the diagonal, B + B^T doubles it back
            }
        // Y = Lt^-1 (lower, explicit);  g = 1/2 Y^T C Y
        float Y[TRI];
#pragma unroll
        for (int j = 0; j < L; ++j) {
            Y[tri(j, j)] = Lm[tri(j, j)];
#pragma unroll
            for (int i = j + 1; i < L; ++i) {
                float s2 = 0.f;
#pragma unroll
                for (int p = j; p < i; ++p) s2 = fmaf(Lm[tri(i, p)], Y[tri(p, j)], s2);
                Y[tri(i, j)] = -s2 * Lm[tri(i, i)];
            }
        }
        float gP[TRI];
#pragma unroll
        for (int i = 0; i < TRI; ++i) gP[i] = 0.f;
        // Z = C Y column by column (Z[:,j] needs Y[p][j], p >= j), then gP[i][j] = 1/2 sum_{p>=i} Y[p][i] Z[p][j]
#pragma unroll
        for (int j = 0; j < L; ++j) {
            float Zc[L];
#pragma unroll
            for (int q = 0; q < L; ++q) {
                float s2 = 0.f;
#pragma unroll
                for (int p = j; p < L; ++p) {
                    const float cqp = (q >= p) ? Cs[tri(q, p)] : Cs[tri(p, q)];
                    s2 = fmaf(cqp, Y[tri(p, j)], s2);
                }
                Zc[q] = s2;
            }
#pragma unroll
            for (int i = j; i < L; ++i) {
                float s2 = 0.f;
#pragma unroll
                for (int p = i; p < L; ++p) s2 = fmaf(Y[tri(p, i)], Zc[p], s2);
                gP[tri(i, j)] = 0.5f * s2;
            }
        }
        // rank-one terms:  - sym(V mu^T) - 1/2 Gc mu mu^T
#pragma unroll
        for (int i = 0; i < L; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j)
                gP[tri(i, j)] += -0.5f * (V[i] * mu[j] + mu[i] * V[j]) - 0.5f * Gc * mu[i] * mu[j];

        // ---- per-component sums (registers) and per-row sums (through LDS)
        if (on) {
#pragma unroll
            for (int i = 0; i < L; ++i) accl[i * WAVE + lane] += gh[i];
#pragma unroll
            for (int i = 0; i < TRI; ++i) accl[(L + i) * WAVE + lane] += gP[i];
            accl[(L + TRI) * WAVE + lane] += Gc;
            if (student) accl[(TH + L + TRI) * WAVE + lane] -= gT;   // T' has -kappa_k
        }
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const float s1 = row_sum(on ? gh[i] : 0.f, scr, lane, rbase, K);
            const float s2 = row_sum(on ? gP[tri(i, i)] : 0.f, scr, lane, rbase, K);
            if (on && k == 0) {
                a.g_eta1[row * L + i] = s1;
                a.g_eta2d[row * L + i] = -2.f * s2;          // p = -2 eta2d
            }
        }
    }

    // ---- block reduction of the per-component sums: lanes with equal k, all waves, fixed order
    __syncthreads();
    for (int w = 0; w < nw; ++w) {
        if (wave == w) {
            for (int i = 0; i < PWa; ++i) red[i * WAVE + lane] = (w == 0 ? 0.f : red[i * WAVE + lane]) + accl[i * WAVE + lane];
        }
        __syncthreads();
    }
    float* out = a.partials + (long long)blockIdx.x * K * PW;
    for (int e = threadIdx.x; e < K * PW; e += blockDim.x) {
        const int kk = e / PW, f = e - kk * PW;
        float s2 = 0.f;
        if (f < PWa)
            for (int rr = 0; rr < RPT; ++rr) s2 += red[f * WAVE + rr * K + kk];
        out[e] = s2;
    }
}


// ---------------------------------------------------------------------------------------------------------
// subsample_x (reference svae.py:122-151): z_ns ~ Cat(exp log_z_n), x_samples[n,s,:] = x[n, z_ns, s, :].
// The categorical draw is the inverse CDF of a supplied uniform (or a supplied index, for parity tests).
// ---------------------------------------------------------------------------------------------------------
// Large-S form of the forward kernel (evaluation runs use S=100, experiments.py:283): the cell's L*S noise block no
// longer fits the per-wave LDS tile, so the samples are processed SC at a time.  Same lane mapping and arithmetic
// order per sample as svae_estep_fwd_kernel; eps^2 / q_theta accumulate across chunks in sample order.
template <int L>
__global__ __launch_bounds__(SV_FWD_MAX_NW * WAVE) void svae_estep_fwd_chunked_kernel(EFwdArgs a, int SC) {
    constexpr int TRI = SvGeo<L>::TRI;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int K = a.K, S = a.S;
    const int LSn = L * S, CSTR = (L * SC) | 1;
    const int RPT = WAVE / K, CT = RPT * K;
    float* et = smem + wave * (WAVE * CSTR + WAVE);
    float* scr = et + WAVE * CSTR;
    const bool lane_on = lane < CT;
    const int r = lane / K, k = lane - r * K, rbase = lane_on ? r * K : 0;
    float Pl[TRI], hkk[L], mkk[L], Wt[TRI];
    const bool student = a.nu != nullptr;
    const int kc = lane_on ? k : 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const float hv = a.hk[kc * L + i], mv = a.mk[kc * L + i];
        hkk[i] = lane_on ? hv : 0.f;
        mkk[i] = lane_on ? mv : 0.f;
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const float pvv = a.Pk[(kc * L + i) * L + j], wv = a.Wk[(kc * L + i) * L + j];
            Pl[tri(i, j)] = lane_on ? pvv : (i == j ? 1.f : 0.f);
            Wt[tri(i, j)] = lane_on ? wv : 0.f;
        }
    }
    const float bv = a.bias[kc], kv = a.kappa[kc], nv = *(student ? a.nu + kc : a.bias);
    const float biask = lane_on ? bv : 0.f, kappak = lane_on ? kv : 0.f, nuk = (student && lane_on) ? nv : 1.f;
    const float inv_nu = 1.0f / nuk, invS = 1.0f / (float)S;

    const long long ntiles = (a.N + RPT - 1) / RPT;
    for (long long t = (long long)blockIdx.x * nw + wave; t < ntiles; t += (long long)gridDim.x * nw) {
        const long long row = t * RPT + r;
        const bool on = lane_on && row < a.N;
        const long long rows_here = (a.N - t * RPT) < RPT ? (a.N - t * RPT) : RPT;
        const int cells = (int)rows_here * K;
        float Lm[TRI], av[L];
#pragma unroll
        for (int i = 0; i < TRI; ++i) Lm[i] = Pl[i];
        const long long rowc = on ? row : 0;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const float e1v = a.eta1[rowc * L + i], e2v = a.eta2d[rowc * L + i];
            const float e1 = on ? e1v : 0.f;
            const float e2 = on ? e2v : -0.5f;
            Lm[tri(i, i)] = fmaf(-2.f, e2, Lm[tri(i, i)]);
            av[i] = e1 + hkk[i];
        }
        float ld;
        cell_cholesky<L>(Lm, ld);
        solve_lower<L>(Lm, av);
        float aa = 0.f;
#pragma unroll
        for (int i = 0; i < L; ++i) aa = fmaf(av[i], av[i], aa);
        const float c = on ? (biask + 0.5f * aa - ld) : -INFINITY;
        const float mx = row_max(c, scr, lane, rbase, K);
        const float ex = on ? __expf(c - mx) : 0.f;
        const float se = row_sum(ex, scr, lane, rbase, K);
        const float lz = c - mx - __logf(se);

        float eps2 = 0.f, qth = 0.f;
        float* cell = et + lane * CSTR;
        const float* __restrict__ gin = a.noise + t * CT * LSn;
        float* __restrict__ gout = a.x + t * CT * LSn;
        for (int c0 = 0; c0 < S; c0 += SC) {
            const int sc = (S - c0) < SC ? (S - c0) : SC;
            const int per = L * sc, tot = cells * per;
            for (int e = lane; e < tot; e += WAVE) {       // noise (cell, L, S) -> LDS (cell, L, sc)
                const int c2 = e / per, rem = e - c2 * per;
                const int i = rem / sc, s = rem - i * sc;
                et[c2 * CSTR + i * sc + s] = gin[c2 * LSn + i * S + c0 + s];
            }
            __builtin_amdgcn_wave_barrier();
            for (int s = 0; s < sc; ++s) {
                float z[L];
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    const float e = lane_on ? cell[i * sc + s] : 0.f;
                    eps2 = fmaf(e, e, eps2);
                    z[i] = av[i] + e;
                }
                solve_lower_t<L>(Lm, z);
                float d[L];
#pragma unroll
                for (int i = 0; i < L; ++i) d[i] = z[i] - mkk[i];
                float del2 = 0.f;
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    float y = 0.f;
#pragma unroll
                    for (int j = 0; j <= i; ++j) y = fmaf(Wt[tri(i, j)], d[j], y);
                    del2 = fmaf(y, y, del2);
                }
                qth += student ? (nuk + (float)L) * log1pf(del2 * inv_nu) : del2;
                if (lane_on) {
#pragma unroll
                    for (int i = 0; i < L; ++i) cell[i * sc + s] = z[i];
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int o = lane; o < tot; o += WAVE) {       // LDS (cell, L, sc) -> x (cell, S, L)
                const int c2 = o / per, rem = o - c2 * per;
                const int s = rem / L, l = rem - s * L;
                gout[c2 * LSn + (c0 + s) * L + l] = et[c2 * CSTR + l * sc + s];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (on) {
            a.lz[row * K + k] = lz;
            a.Tp[row * K + k] = -0.5f * L * LOG_2PI + ld - 0.5f * invS * eps2 + 0.5f * invS * qth - kappak;
        }
    }
}

struct SubArgs {
    const float* x;         // (N,K,S,L)
    const float* lz;        // (N,K)
    const float* u;         // (N,S_out) uniforms in [0,1) or NULL
    const long long* z;     // (N,S_out) indices or NULL
    float* out;             // (N,S_out,L)
    long long* z_out;       // (N,S_out) chosen component (may be NULL)
    long long N;
    int K, S, L, S_out;
};

__global__ __launch_bounds__(256) void subsample_kernel(SubArgs a) {
    const long long tot = a.N * a.S_out;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < tot; g += (long long)gridDim.x * blockDim.x) {
        const long long n = g / a.S_out;
        const int s = (int)(g - n * a.S_out);
        int zk;
        if (a.z) {
            zk = (int)a.z[g];
        } else {
            const float uu = a.u[g];
            float cum = 0.f;
            zk = a.K - 1;
            for (int k = 0; k < a.K; ++k) {
                cum += __expf(a.lz[n * a.K + k]);
                if (uu < cum) { zk = k; break; }
            }
        }
        if (a.z_out) a.z_out[g] = zk;
        const float* src = a.x + ((n * a.K + zk) * a.S + s) * a.L;
        float* dst = a.out + g * a.L;
        for (int l = 0; l < a.L; ++l) dst[l] = src[l];
    }
}

int check_sv(long long N, int K, int L, int S) {
    if (N <= 0 || S <= 0) { set_error("N and S must be positive"); return VMP_E_BADARG; }
    if (L < 1 || L > VMP_MAX_D) { set_error("L=%d outside compiled range 1..%d", L, VMP_MAX_D); return VMP_E_DIM; }
    if (K < 1 || K > VMP_MAX_K) { set_error("K=%d outside compiled range 1..%d", K, VMP_MAX_K); return VMP_E_DIM; }
    if (S > (1 << 16)) { set_error("S=%d too large", S); return VMP_E_DIM; }
    return 0;
}

int sv_blocks(long long N, int K) {
    const int RPT = WAVE / K;
    const long long ntiles = (N + RPT - 1) / RPT;
    long long b = (ntiles + SV_NW - 1) / SV_NW;
    static const int cap = getenv("VMP_SV_BLOCKS") ? atoi(getenv("VMP_SV_BLOCKS")) : 1024;
    if (b > cap) b = cap;
    if (b > SV_MAX_BLOCKS) b = SV_MAX_BLOCKS;
    return (int)b;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#define VMP_DISPATCH_L(Lv, CALL)           \
    switch (Lv) {                           \
        case 1: { constexpr int LL = 1; CALL; } break; \
        case 2: { constexpr int LL = 2; CALL; } break; \
        case 3: { constexpr int LL = 3; CALL; } break; \
        case 4: { constexpr int LL = 4; CALL; } break; \
        case 5: { constexpr int LL = 5; CALL; } break; \
        case 6: { constexpr int LL = 6; CALL; } break; \
        case 7: { constexpr int LL = 7; CALL; } break; \
        case 8: { constexpr int LL = 8; CALL; } break; \
        default: break;                     \
    }

}  // namespace

extern "C" {

int vmp_svae_bwd_partial_words(int L) { return 2 * (L + L * (L + 1) / 2 + 1); }

size_t vmp_svae_workspace_bytes(int64_t N, int K, int L) {
    (void)N;
    return (size_t)SV_MAX_BLOCKS * K * vmp_svae_bwd_partial_words(L) * sizeof(float);
}

int vmp_svae_bwd_blocks(int64_t N, int K) { return sv_blocks(N, K); }

int vmp_svae_estep_fwd(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                       const float* noise, const float* mk, const float* Wk, const float* kappa, const float* nu,
                       int64_t N, int K, int L, int S, float* x, float* lz, float* Tp, void* stream) {
    int rc = check_sv(N, K, L, S);
    if (rc) return rc;
    if (!eta1 || !eta2d || !hk || !Pk || !bias || !noise || !mk || !Wk || !kappa || !x || !lz || !Tp) {
        set_error("vmp_svae_estep_fwd: null pointer");
        return VMP_E_BADARG;
    }
    EFwdArgs a{eta1, eta2d, hk, Pk, bias, noise, mk, Wk, kappa, nu, x, lz, Tp, N, K, S, 0};
    a.vec_ok = al16(noise) && al16(x);
    {
        const int SP = (S + 1) / 2;
        static const int use_v2 = getenv("VMP_SV_FWD_V2") ? 1 : 0;   // measured slower than the staged form at C3 (4.1 vs 3.6 ms)
        if (use_v2 && K * SP <= 512) {
            int RPB = 512 / (K * SP);
            if (RPB > 8) RPB = 8;
            const int threads = ((RPB * K * SP + WAVE - 1) / WAVE) * WAVE;
            const int CB = RPB * K;
            const size_t lds2 = (size_t)(K * ((L * (L + 1) / 2) | 1) + CB + 2 * CB * SP + 2 * WAVE) * sizeof(float);
            long long blocks2 = (N + RPB - 1) / RPB;
            if (blocks2 > 256 * 6) blocks2 = 256 * 6;
            rc = -1;
            VMP_DISPATCH_L(L, {
                hipLaunchKernelGGL((svae_estep_fwd2_kernel<LL>), dim3((int)blocks2), dim3(threads), lds2, static_cast<hipStream_t>(stream), a, RPB, SP);
                rc = check_launch("svae_estep_fwd2_kernel");
            });
            return rc;
        }
    }
    if ((size_t)(L * S | 1) * WAVE * sizeof(float) > 36 * 1024) {
        // the cell's noise block does not fit the LDS tile: process the samples SC at a time
        int SC = (32 * 1024 / (int)(WAVE * sizeof(float))) / L;   // L*SC*64*4 B <= 32 KiB per wave
        const size_t pw = (size_t)(WAVE * ((L * SC) | 1) + WAVE) * sizeof(float);
        const int nwc = 4;
        const int RPTc = WAVE / K;
        long long bl = ((N + RPTc - 1) / RPTc + nwc - 1) / nwc;
        if (bl > 1024) bl = 1024;
        rc = -1;
        VMP_DISPATCH_L(L, {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(svae_estep_fwd_chunked_kernel<LL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(pw * nwc));
            hipLaunchKernelGGL((svae_estep_fwd_chunked_kernel<LL>), dim3((int)bl), dim3(nwc * WAVE), pw * nwc, static_cast<hipStream_t>(stream), a, SC);
            rc = check_launch("svae_estep_fwd_chunked_kernel");
        });
        return rc;
    }
    const size_t per_wave = (size_t)(WAVE * (L * S | 1) + WAVE) * sizeof(float);
    int nw = (int)((150 * 1024) / per_wave);                 // one block per CU, as many waves as 160 KiB of LDS hold
    if (nw > SV_FWD_MAX_NW) nw = SV_FWD_MAX_NW;
    if (nw < 1) nw = 1;
    const size_t lds = per_wave * nw;
    const int RPT = WAVE / K;
    long long blocks = ((N + RPT - 1) / RPT + nw - 1) / nw;
    if (blocks > 256) blocks = 256;
    rc = -1;
    VMP_DISPATCH_L(L, {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(svae_estep_fwd_kernel<LL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((svae_estep_fwd_kernel<LL>), dim3((int)blocks), dim3(nw * WAVE), lds, static_cast<hipStream_t>(stream), a);
        rc = check_launch("svae_estep_fwd_kernel");
    });
    return rc;
}

int vmp_svae_estep_bwd(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                       const float* mk, const float* Wk, const float* nu, const float* x, const float* lz, const float* Gx,
                       const float* Glz, const float* GT, int64_t N, int K, int L, int S, float* g_eta1, float* g_eta2d,
                       float* partials, size_t partial_bytes, void* stream) {
    int rc = check_sv(N, K, L, S);
    if (rc) return rc;
    if (!eta1 || !eta2d || !hk || !Pk || !bias || !mk || !Wk || !x || !lz || !Gx || !Glz || !GT || !g_eta1 || !g_eta2d || !partials) {
        set_error("vmp_svae_estep_bwd: null pointer");
        return VMP_E_BADARG;
    }
    const int blocks = sv_blocks(N, K);
    const int PW = vmp_svae_bwd_partial_words(L);
    if (partial_bytes < (size_t)blocks * K * PW * sizeof(float)) { set_error("vmp_svae_estep_bwd: partials buffer too small"); return VMP_E_WS; }
    EBwdArgs a{eta1, eta2d, hk, Pk, bias, mk, Wk, nu, x, lz, Gx, Glz, GT, g_eta1, g_eta2d, partials, N, K, S, 0};
    a.vec_ok = al16(x) && al16(Gx);
    const int PWa = nu ? PW : PW / 2;
    const size_t lds = (size_t)(K * ((L * (L + 1) / 2) | 1) + SV_NW * WAVE + PWa * WAVE + SV_NW * PWa * WAVE) * sizeof(float);
    rc = -1;
    VMP_DISPATCH_L(L, {
        hipLaunchKernelGGL((svae_estep_bwd_kernel<LL>), dim3(blocks), dim3(SV_NW * WAVE), lds, static_cast<hipStream_t>(stream), a);
        rc = check_launch("svae_estep_bwd_kernel");
    });
    return rc;
}

int vmp_svae_subsample(const float* x, const float* lz, const float* u, const int64_t* z, int64_t N, int K, int S, int L,
                       int S_out, float* out, int64_t* z_out, void* stream) {
    if (!x || !lz || (!u && !z) || !out || N <= 0 || K <= 0 || S <= 0 || L <= 0 || S_out <= 0 || S_out > S) {
        set_error("vmp_svae_subsample: bad argument");
        return VMP_E_BADARG;
    }
    SubArgs a{x, lz, u, reinterpret_cast<const long long*>(z), out, reinterpret_cast<long long*>(z_out), N, K, S, L, S_out};
    long long blocks = (N * S_out + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(subsample_kernel, dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("subsample_kernel");
}

}  // extern "C"
