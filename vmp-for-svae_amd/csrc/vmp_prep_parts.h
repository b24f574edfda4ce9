// Device bodies of the K-sized parameter maps (vmp_prep.hip: recognition unpacking forward / backward, theta packing), shared with
// the launches that run them beside other work (round 6: vmp_step.hip step_final_kernel).  See vmp_prep.hip's header comment.
#pragma once
#include "vmp_common.h"
#include "vmp_step_parts.h"

namespace vmp {

constexpr int PREP_THREADS = 64;     // K <= VMP_MAX_K = 64

// The bodies below are ONE wave of work.  WS = false: the wave is the whole block and synchronises with the block barrier;
// WS = true: the body runs in wave 0 of a larger block whose other waves have left (round 6: the K-sized maps as extra blocks of the
// encoder's forward launch) - the LDS unit executes a wave's accesses in order, only the compiler must keep the program order.
template <bool WS>
__device__ __forceinline__ void prep_sync() {
    if (WS) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

__device__ __forceinline__ double softplus_d(double x) { return x > 0.0 ? x + log1p(exp(-x)) : log1p(exp(x)); }

__device__ inline double digamma_dd(double x) {
    double r = 0.0;
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x
           - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760))))));
}

// L_k = tril(raw) with softplus on the diagonal (svae.py:347-352), rounded to fp32 as the tensors the reference holds
template <int L>
__device__ __forceinline__ void load_Lk(const float* __restrict__ raw, double (&Lm)[L][L]) {
#pragma unroll
    for (int i = 0; i < L; ++i)
#pragma unroll
        for (int j = 0; j < L; ++j) {
            double v = 0.0;
            if (j < i) v = (double)raw[i * L + j];
            else if (j == i) v = (double)(float)softplus_d((double)raw[i * L + i]);
            Lm[i][j] = v;
        }
}

// One lane group's share of the fixed-order fp64 reduction of the E-step backward kernel's partial rows (svae_bwd_reduce_kernel below):
// group bg adds the rows b = bg, bg + 16, .. as four independent chains (their loads are in flight together), combined in a fixed order.
constexpr int RED_GROUPS = 16;
__device__ __forceinline__ double red_group_sum(const float* __restrict__ p, const size_t bs, const int nblk, const int bg) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = bg;
    for (; b + 3 * RED_GROUPS < nblk; b += 4 * RED_GROUPS) {
        s0 += (double)p[(size_t)b * bs]; s1 += (double)p[(size_t)(b + RED_GROUPS) * bs];
        s2 += (double)p[(size_t)(b + 2 * RED_GROUPS) * bs]; s3 += (double)p[(size_t)(b + 3 * RED_GROUPS) * bs];
    }
    for (; b < nblk; b += RED_GROUPS) s0 += (double)p[(size_t)b * bs];
    return (s0 + s1) + (s2 + s3);
}
struct PhiArgs {
    const float* mu;      // (K,L)  'phi_gmm/mu_k' (used as eta1, svae.py:345)
    const float* Lraw;    // (K,L,L)
    const float* piraw;   // (K)
    const float* g_hk;    // bwd: (K,L)
    const float* g_P;     // bwd: (K,L,L) gradient w.r.t. the full matrix P = L L^T
    const float* g_bias;  // bwd: (K)
    float* Lk;            // fwd out (K,L,L)
    float* P;             // fwd out (K,L,L)
    float* bias;          // fwd out (K): B_k + log softmax(piraw)_k
    float* g_mu;          // bwd out
    float* g_Lraw;        // bwd out
    float* g_piraw;       // bwd out
    int K, L;
    double* logpi_out;    // fwd out, optional: log softmax(piraw)_k in fp64 (K) - lets a backward block run without reading piraw_j, j != k
    const double* logpi;  // bwd (RED form): the forward's logpi_out
};

// Adam applied to component k's phi_gmm elements right where their gradients are formed (round 6, step_final_kernel): the three
// tensors mu_k, L_k, log_pi_k with their slots; NULL = gradients only.
struct PhiAdam {
    float* p[3];                 // NULL: no update (data-parallel step: Adam follows the all-reduce)
    float* m[3];
    float* v[3];
    double* gx[3];               // optional: the gradients also as doubles (the packed exchange buffer of a data-parallel step)
    float lr_t, b1, b2, c1, c2, eps;
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
// r[idx] of a register array with a lane-dependent index (selects: no scratch memory)
template <int L>
__device__ __forceinline__ double pick(const double (&r)[L], int idx) {
    double v = r[0];
#pragma unroll
    for (int q = 1; q < L; ++q) v = idx == q ? r[q] : v;
    return v;
}

// One 64-lane block per component; lane (i, j) = (lane / L, lane % L) owns element (i, j) of the L x L matrices, lanes
// j < K the K-sized softmax terms.  The transcendental work (softplus, log, exp - sequences of dozens of fp64
// instructions each) is spread over the lanes; only the two triangular solves are serial (every lane runs them on
// operands broadcast from LDS).  (One thread per component did all of it serially: 11-12 us per launch at L = 8.)
// RED (backward only, round 6; a block of 64 * RED_GROUPS threads): the upstream gradients (g_hk, g_P, g_bias) are not read from
// tensors but summed here from the E-step backward kernel's `nblk` partial rows - the work of svae_bwd_reduce_kernel for component k
// (and the g_bias column of all components), value for value in its order and rounded to fp32 as the tensors were; log softmax(pi)_k
// comes from the forward pass (a.logpi), so that the block reads nothing of another component's parameters - with `ad` it may
// update its own in place (Adam) while the other blocks run.  Waves 1.. only take part in the reduction.
template <int L, bool BWD, bool RED = false, bool WS = false>
__device__ __forceinline__ void phi_prep_body(const PhiArgs& a, const int k, const float* __restrict__ partials = nullptr, const int nblk = 0,
                                              const PhiAdam* ad = nullptr) {
    static_assert(L * L <= PREP_THREADS, "one lane per matrix element");
    static_assert(!RED || BWD, "RED is a backward form");
    static_assert(!(RED && WS), "the RED form synchronises its whole block");
    const int lane = threadIdx.x, K = a.K;
    __shared__ double Ls[L][L + 1];
    __shared__ double Gm[L][L + 1];
    __shared__ double inv_d[L];
    __shared__ float ghs[L];
    __shared__ float gbs[PREP_THREADS];
    if (RED) {
        constexpr int TRI = L * (L + 1) / 2, TH = L + TRI + 1, PW = 2 * TH;
        static_assert(TH <= 64, "one lane per partial word");
        __shared__ double rp[2][RED_GROUPS][64];
        const int eg = threadIdx.x & 63, bg = threadIdx.x >> 6;
        const size_t bs = (size_t)K * PW;
        rp[0][bg][eg] = eg < L + TRI ? red_group_sum(partials + (size_t)k * PW + eg, bs, nblk, bg) : 0.0;
        rp[1][bg][eg] = eg < K ? red_group_sum(partials + (size_t)eg * PW + (L + TRI), bs, nblk, bg) : 0.0;
        __syncthreads();
        if (threadIdx.x >= PREP_THREADS) {
            __syncthreads();                               // (the one block barrier of the body below)
            return;
        }
        double s0 = rp[0][0][lane], s1 = rp[1][0][lane];
        for (int g2 = 1; g2 < RED_GROUPS; ++g2) { s0 += rp[0][g2][lane]; s1 += rp[1][g2][lane]; }
        const float v = (float)s0;
        if (lane < L) ghs[lane] = v;
        else if (lane < L + TRI) {
            int idx = lane - L, ii = 0;
            while ((ii + 1) * (ii + 2) / 2 <= idx) ++ii;
            const int jj = idx - ii * (ii + 1) / 2;
            Gm[ii][jj] = (double)v;
            Gm[jj][ii] = (double)v;                        // symmetric gradient: both triangles carry the packed lower value
        }
        gbs[lane] = (float)s1;
    }
    const int i = lane / L, j = lane % L;
    const bool in = lane < L * L;
    const float* raw = a.Lraw + (size_t)k * L * L;          // (no restrict: with `ad` these elements are written below)
    double lij = 0.0;
    if (in) {
        if (j < i) lij = (double)raw[i * L + j];
        else if (j == i) lij = (double)(float)softplus_d((double)raw[i * L + i]);
        Ls[i][j] = lij;
        if (i == j) inv_d[i] = 1.0 / lij;
        if (BWD && !RED) Gm[i][j] = (double)a.g_P[(size_t)k * L * L + i * L + j];
    }
    double logpi;
    if (RED) logpi = a.logpi[k];
    else {
        const double pr = lane < K ? (double)a.piraw[lane] : -1e300;
        const double mx = wave_max_d(pr);
        const double se = wave_sum_d(lane < K ? exp(pr - mx) : 0.0);
        logpi = (double)a.piraw[k] - (mx + log(se));
    }
    const double ld = wave_sum_d((in && i == j) ? log(lij) : 0.0);           // log det L
    prep_sync<WS>();
    double s[L];                                        // s = L^-1 h
    double q2 = 0.0;
#pragma unroll
    for (int r = 0; r < L; ++r) {
        double t = (double)a.mu[k * L + r];
#pragma unroll
        for (int c = 0; c < r; ++c) t -= Ls[r][c] * s[c];
        s[r] = t * inv_d[r];
        q2 += s[r] * s[r];
    }
    if (!BWD) {
        if (in) {
            a.Lk[(size_t)k * L * L + lane] = (float)lij;
            double p = 0.0;
            const int m = i < j ? i : j;
#pragma unroll
            for (int q = 0; q < L; ++q) p += q <= m ? Ls[i][q] * Ls[j][q] : 0.0;
            a.P[(size_t)k * L * L + lane] = (float)p;
        }
        if (lane == 0) {
            a.bias[k] = (float)(-0.5 * q2 + ld + logpi);
            if (a.logpi_out) a.logpi_out[k] = logpi;
        }
        return;
    }
    // ---- backward
    double u[L];                                        // u = L^-T s = P^-1 h
#pragma unroll
    for (int r = L - 1; r >= 0; --r) {
        double t = s[r];
#pragma unroll
        for (int c = r + 1; c < L; ++c) t -= Ls[c][r] * u[c];
        u[r] = t * inv_d[r];
    }
    const double s_gb = wave_sum_d(lane < K ? (double)(RED ? gbs[lane] : a.g_bias[lane]) : 0.0);
    const double gb = (double)(RED ? gbs[k] : a.g_bias[k]);
    if (lane < L) {
        const float g = (float)((double)(RED ? ghs[lane] : a.g_hk[k * L + lane]) - gb * pick<L>(u, lane));
        a.g_mu[k * L + lane] = g;
        if (ad && ad->gx[0]) ad->gx[0][k * L + lane] = (double)g;
        if (ad && ad->p[0]) adam_update(ad->p[0], ad->m[0], ad->v[0], (unsigned)(k * L + lane), g, ad->lr_t, ad->b1, ad->b2, ad->c1, ad->c2, ad->eps);
    }
    if (lane == 0) {
        const float g = (float)(gb - exp(logpi) * s_gb);
        a.g_piraw[k] = g;
        if (ad && ad->gx[2]) ad->gx[2][k] = (double)g;
        if (ad && ad->p[2]) adam_update(ad->p[2], ad->m[2], ad->v[2], (unsigned)k, g, ad->lr_t, ad->b1, ad->b2, ad->c1, ad->c2, ad->eps);
    }
    if (in) {
        double g = 0.0;
        if (j <= i) {
#pragma unroll
            for (int q = 0; q < L; ++q) g += q >= j ? (Gm[i][q] + Gm[q][i]) * Ls[q][j] : 0.0;    // (G + G^T) L
            g += gb * pick<L>(u, i) * pick<L>(s, j);
            if (i == j) {
                g += gb * inv_d[i];
                const double r = (double)raw[i * L + i];
                g *= 1.0 / (1.0 + exp(-r));             // softplus'
            }
        }
        const float gf = (float)g;
        a.g_Lraw[(size_t)k * L * L + lane] = gf;
        if (ad && ad->gx[1]) ad->gx[1][(size_t)k * L * L + lane] = (double)gf;
        if (ad && ad->p[1]) adam_update(ad->p[1], ad->m[1], ad->v[1], (unsigned)(k * L * L + lane), gf, ad->lr_t, ad->b1, ad->b2, ad->c1, ad->c2, ad->eps);
    }
}

struct ThetaArgs {
    const float *alpha, *A, *b, *beta, *vhat;     // natural NIW / Dirichlet parameters
    float* m;        // (K,L)
    float* W;        // (K,L,L) lower, W^T W = E[Sigma]^-1
    float* kappa;    // (K)
    int K, L;
};

template <int L, bool WS = false>
__device__ __forceinline__ void theta_pack_body(const ThetaArgs& a, const int k) {
    static_assert(L * L <= PREP_THREADS, "one lane per matrix element");
    const int lane = threadIdx.x, K = a.K;
    __shared__ double Cs[L][L + 1];                     // sym(C) / nu, overwritten by its Cholesky factor (lower)
    __shared__ double inv_d[L];
    const int i = lane / L, j = lane % L;
    const bool in = lane < L * L;
    const double asum = wave_sum_d(lane < K ? (double)a.alpha[lane] + 1.0 : 0.0);            // dirichlet.natural_to_standard
    const double dg = lane < 2 ? digamma_dd(lane == 0 ? (double)a.alpha[k] + 1.0 : asum) : 0.0;
    const double elp = __shfl(dg, 0) - __shfl(dg, 1);
    const double beta = (double)a.beta[k], nu = (double)a.vhat[k] - (double)(L + 2);         // niw.natural_to_standard
    const double inv_nu = 1.0 / nu;
    if (in) {
        const double bi = (double)a.b[k * L + i], bj = (double)a.b[k * L + j];
        const double cij = (double)a.A[((size_t)k * L + i) * L + j] - bi * (bj / beta);
        const double cji = (double)a.A[((size_t)k * L + j) * L + i] - bj * (bi / beta);
        Cs[i][j] = 0.5 * (cij + cji) * inv_nu;          // E[Sigma] = sym(C) / nu  (niw.expected_values)
    }
    prep_sync<WS>();
    // Cholesky: column c; lanes r = lane < L take the rows r >= c; every lane also forms the pivot itself
#pragma unroll
    for (int c = 0; c < L; ++c) {
        const int r = lane < L ? lane : c;
        double t = Cs[r][c], d = Cs[c][c];
#pragma unroll
        for (int q = 0; q < c; ++q) {
            t -= Cs[r][q] * Cs[c][q];
            d -= Cs[c][q] * Cs[c][q];
        }
        const double sd = sqrt(d);
        prep_sync<WS>();
        if (lane < L && lane >= c) Cs[lane][c] = lane == c ? sd : t / sd;
        if (lane == c) inv_d[c] = 1.0 / sd;
        prep_sync<WS>();
    }
    // W = Lc^-1: lane c < L solves for column c
    double wdiag = 1.0;
    if (lane < L) {
        const int c = lane;
        double w[L];
#pragma unroll
        for (int r = 0; r < L; ++r) {
            double t = r == c ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < r; ++q) t -= Cs[r][q] * w[q];
            w[r] = r < c ? 0.0 : t * inv_d[r];
            if (r == c) wdiag = w[r];
            a.W[((size_t)k * L + r) * L + c] = (float)w[r];
        }
        a.m[k * L + lane] = (float)((double)a.b[k * L + lane] / beta);
    }
    const double lw = wave_sum_d(lane < L ? log(wdiag) : 0.0);
    if (lane == 0) a.kappa[k] = (float)(-0.5 * L * 1.8378770664093454836 + elp + lw);
}

}  // namespace vmp
