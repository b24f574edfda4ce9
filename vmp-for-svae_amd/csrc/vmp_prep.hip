// K-sized parameter maps of the SVAE training step (reference models/svae.py:342-358 unpack_recognition_gmm,
// :205-214 the theta side of compute_elbo via distributions/niw.py:8-43 + dirichlet.py:8-22, :154-176 m_step and
// :376-403 update_gmm_params) as single-launch kernels: one 64-lane block per mixture component, fp64 inside, fp32 in/out.
// At the reference's real operating point (minibatches of 64-100 rows) the step is bound by the NUMBER of launches:
// these maps were ~120 tiny torch kernels (batched Cholesky / triangular solves with host-side error checks, tril,
// softplus, digamma ...) and their autograd; here they are 4 launches.
#include "vmp_common.h"

using namespace vmp;

namespace {

constexpr int PREP_THREADS = 64;     // K <= VMP_MAX_K = 64
// L is a template parameter everywhere: with run-time loop bounds the per-thread matrices live in scratch memory and
// every element access is a dependent round trip (measured: 50 us per launch instead of a few).
#define PREP_DISPATCH_L(L, CALL)                                                     \
    switch (L) {                                                                     \
        case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; \
        case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; default: CALL(8); break; \
    }

__device__ __forceinline__ double softplus_d(double x) { return x > 0.0 ? x + log1p(exp(-x)) : log1p(exp(x)); }

__device__ double digamma_dd(double x) {
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
template <int L, bool BWD>
__device__ __forceinline__ void phi_prep_body(const PhiArgs& a, const int k) {
    static_assert(L * L <= PREP_THREADS, "one lane per matrix element");
    const int lane = threadIdx.x, K = a.K;
    __shared__ double Ls[L][L + 1];
    __shared__ double Gm[L][L + 1];
    __shared__ double inv_d[L];
    const int i = lane / L, j = lane % L;
    const bool in = lane < L * L;
    const float* __restrict__ raw = a.Lraw + (size_t)k * L * L;
    double lij = 0.0;
    if (in) {
        if (j < i) lij = (double)raw[i * L + j];
        else if (j == i) lij = (double)(float)softplus_d((double)raw[i * L + i]);
        Ls[i][j] = lij;
        if (i == j) inv_d[i] = 1.0 / lij;
        if (BWD) Gm[i][j] = (double)a.g_P[(size_t)k * L * L + i * L + j];
    }
    const double pr = lane < K ? (double)a.piraw[lane] : -1e300;
    const double mx = wave_max_d(pr);
    const double se = wave_sum_d(lane < K ? exp(pr - mx) : 0.0);
    const double logpi = (double)a.piraw[k] - (mx + log(se));
    const double ld = wave_sum_d((in && i == j) ? log(lij) : 0.0);           // log det L
    __syncthreads();
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
        if (lane == 0) a.bias[k] = (float)(-0.5 * q2 + ld + logpi);
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
    const double s_gb = wave_sum_d(lane < K ? (double)a.g_bias[lane] : 0.0);
    const double gb = (double)a.g_bias[k];
    if (lane < L) a.g_mu[k * L + lane] = (float)((double)a.g_hk[k * L + lane] - gb * pick<L>(u, lane));
    if (lane == 0) a.g_piraw[k] = (float)(gb - exp(logpi) * s_gb);
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
        a.g_Lraw[(size_t)k * L * L + lane] = (float)g;
    }
}

struct ThetaArgs {
    const float *alpha, *A, *b, *beta, *vhat;     // natural NIW / Dirichlet parameters
    float* m;        // (K,L)
    float* W;        // (K,L,L) lower, W^T W = E[Sigma]^-1
    float* kappa;    // (K)
    int K, L;
};

// Block per component, lane (i, j) per matrix element as in phi_prep_kernel: the two digammas run side by side in lanes 0
// and 1, the Cholesky factor is built column by column in LDS (rows in parallel), the columns of its inverse in parallel.
template <int L, bool BWD>
__global__ __launch_bounds__(PREP_THREADS) void phi_prep_kernel(PhiArgs a) { phi_prep_body<L, BWD>(a, blockIdx.x); }

template <int L>
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
    __syncthreads();
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
        __syncthreads();
        if (lane < L && lane >= c) Cs[lane][c] = lane == c ? sd : t / sd;
        if (lane == c) inv_d[c] = 1.0 / sd;
        __syncthreads();
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

template <int L>
__global__ __launch_bounds__(PREP_THREADS) void theta_pack_kernel(ThetaArgs a) { theta_pack_body<L>(a, blockIdx.x); }

// Both K-sized forward maps of a training step in ONE launch: blocks [0, K) unpack the recognition GMM, blocks [K, 2K) pack
// theta - they are independent, and at minibatch sizes a launch costs as much as either of them.
template <int L>
__global__ __launch_bounds__(PREP_THREADS) void prep_both_kernel(PhiArgs a, ThetaArgs t) {
    if ((int)blockIdx.x < a.K) phi_prep_body<L, false>(a, blockIdx.x);
    else theta_pack_body<L>(t, blockIdx.x - a.K);
}

struct CviArgs {
    const double* stats;                           // (K, 2+L+L*L): [Nk | Wk | sx | sxx]
    const float *p_alpha, *p_A, *p_b, *p_beta, *p_vhat;   // prior (natural)
    float *t_alpha, *t_A, *t_b, *t_beta, *t_vhat;         // theta (natural), updated in place
    float *s_alpha, *s_A, *s_b, *s_beta, *s_vhat;         // theta* out (may be NULL)
    const float* rho_dev;                          // step size on the device (NULL: use rho)
    float rho;
    int K, L;
};

__device__ __forceinline__ void cvi_one(float* __restrict__ t, float* __restrict__ s, float star, float rho, size_t i) {
    if (s) s[i] = star;
    t[i] = t[i] * (1.0f - rho) + rho * star;       // update_gmm_params: theta <- (1-rho) theta + rho theta*
}

// element f (< L*L + L + 3) of component k; st = that component's raw moments [Nk | Wk | sx | sxx]
__device__ __forceinline__ void cvi_element(const CviArgs& a, const double* __restrict__ st, int k, int f, float rho) {
    const int L = a.L;
    const float Nk = (float)st[0];
    if (f < L * L) {
        const size_t i = (size_t)k * L * L + f;
        cvi_one(a.t_A, a.s_A, a.p_A[i] + (float)st[2 + L + f], rho, i);
    } else if (f < L * L + L) {
        const int d = f - L * L;
        const size_t i = (size_t)k * L + d;
        cvi_one(a.t_b, a.s_b, a.p_b[i] + (float)st[2 + d], rho, i);
    } else if (f == L * L + L) {
        cvi_one(a.t_alpha, a.s_alpha, a.p_alpha[k] + Nk, rho, k);
    } else if (f == L * L + L + 1) {
        cvi_one(a.t_beta, a.s_beta, a.p_beta[k] + Nk, rho, k);
    } else {
        cvi_one(a.t_vhat, a.s_vhat, a.p_vhat[k] + Nk + 1.0f, rho, k);   // the +1 of gmm.update_vk (gmm.py:81)
    }
}

__global__ __launch_bounds__(256) void cvi_kernel(CviArgs a) {
    const int K = a.K, L = a.L, SW = 2 + L + L * L;
    const float rho = a.rho_dev ? *a.rho_dev : a.rho;
    const int per = L * L + L + 3;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < K * per; e += gridDim.x * blockDim.x) {
        const int k = e / per, f = e - k * per;
        cvi_element(a, a.stats + (size_t)k * SW, k, f, rho);
    }
}

// M-step moments of a small batch AND the CVI update in one launch (the single-process training step at minibatch sizes:
// the moments of the whole minibatch are in hand, no all-reduce between the two): block k sums component k's moments
// (small_stats_component, vmp_common.h), publishes them, and updates component k of theta from the copy in LDS.
__global__ __launch_bounds__(SMALL_STATS_GROUPS * 80) void stats_cvi_kernel(SmallStatsArgs sa, CviArgs a) {
    __shared__ double part[SMALL_STATS_GROUPS][80];
    __shared__ double st[80];
    const int k = blockIdx.x, L = a.L, SW = 2 + L + L * L, i = threadIdx.x % 80;
    const double t = small_stats_component(sa, k, part);
    if (threadIdx.x < 80 && i < SW) {
        st[i] = t;
        sa.stats[(long long)k * SW + i] = t;
    }
    __syncthreads();
    const float rho = a.rho_dev ? *a.rho_dev : a.rho;
    for (int f = threadIdx.x; f < L * L + L + 3; f += blockDim.x) cvi_element(a, st, k, f, rho);
}

// M-step moments from the fused E-step forward kernel's per-block partials AND the CVI update in one launch (round 6): the forward
// kernel (csrc/vmp_svae.hip, epilogue) leaves (nblk, 16, 48) fp64 partial sums of r_nk [x | 1 | x x^T lower-packed | 0 0 0] over its
// blocks' rows; block k adds component k's partials in a fixed order (MOM_GROUPS interleaved chains, then the chains in order),
// expands them to the raw-moment row [N_k | W_k = N_k | sum r x | sum r x x^T] (svae.m_step -> gmm.update_Nk/xk/Sk,
// svae.py:154-176, gmm.py:25-46, in the natural-parameter form of SURVEY appendix A.6) and - when theta is given - updates
// component k of theta from the copy in LDS (svae.update_gmm_params, svae.py:376-403).
constexpr int MOM_F = 48, MOM_GROUPS = 16, MOM_L = 8;
struct MomArgs { const double* mom; double* stats; int nblk, K; };
__global__ __launch_bounds__(64 * MOM_GROUPS) void mom_cvi_kernel(MomArgs m, CviArgs a, int do_cvi) {
    __shared__ double part[MOM_GROUPS][64];
    __shared__ double st[2 + MOM_L + MOM_L * MOM_L];
    constexpr int L = MOM_L, SW = 2 + L + L * L;
    const int k = blockIdx.x, f = threadIdx.x & 63, bg = threadIdx.x >> 6;
    // group bg adds the partials b = bg, bg + G, ..: four independent chains (loads in flight together), combined in a fixed order
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (f < MOM_F) {
        const double* __restrict__ p = m.mom + (size_t)k * MOM_F + f;
        const size_t bs = (size_t)m.K * MOM_F;
        int b = bg;
        for (; b + 3 * MOM_GROUPS < m.nblk; b += 4 * MOM_GROUPS) {
            s0 += p[(size_t)b * bs]; s1 += p[(size_t)(b + MOM_GROUPS) * bs];
            s2 += p[(size_t)(b + 2 * MOM_GROUPS) * bs]; s3 += p[(size_t)(b + 3 * MOM_GROUPS) * bs];
        }
        for (; b < m.nblk; b += MOM_GROUPS) s0 += p[(size_t)b * bs];
    }
    double s = (s0 + s1) + (s2 + s3);
    part[bg][f] = s;
    __syncthreads();
    if (bg == 0) {
        for (int g2 = 1; g2 < MOM_GROUPS; ++g2) s += part[g2][f];
        part[0][f] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < SW; e += blockDim.x) {
        double v;
        if (e < 2) v = part[0][L];                          // N_k; W_k = N_k (no per-row scale in the SVAE M-step)
        else if (e < 2 + L) v = part[0][e - 2];
        else {
            const int i = (e - 2 - L) / L, j = (e - 2 - L) - i * L;
            const int hi = i > j ? i : j, lo = i > j ? j : i;
            v = part[0][L + 1 + hi * (hi + 1) / 2 + lo];
        }
        st[e] = v;
        if (m.stats) m.stats[(size_t)k * SW + e] = v;
    }
    __syncthreads();
    if (do_cvi) {
        const float rho = a.rho_dev ? *a.rho_dev : a.rho;
        for (int e = threadIdx.x; e < L * L + L + 3; e += blockDim.x) cvi_element(a, st, k, e, rho);
    }
}

// Reduction of the fused E-step backward kernel's per-block partial sums (vmp_svae_estep_bwd: (nblk, K, PW) with
// PW = 2 (L + TRI + 1): [g_hk | g_Pk lower | g_bias | g_mk | g_Wk lower | g_kappa]) in a fixed order in fp64, unpacked
// into the K-sized gradient tensors: g_P symmetric (both triangles carry the packed lower value), g_W lower.
struct RedArgs {
    const float* partials;
    float *g_hk, *g_P, *g_bias, *g_mk, *g_W, *g_kappa;     // theta-side outputs may be NULL
    int nblk, K, L;
};

// A block sums 64 consecutive (k, f) elements: lane group bg = tid / 64 takes the rows b = bg, bg + 16, .. of the partials
// (64 consecutive floats per row: coalesced, independent loads), the 16 group sums are then added in a fixed order.
// (One thread per element walking all rows was a chain of nblk dependent strided loads: 390 us at nblk = 1024.)
constexpr int RED_GROUPS = 16;
__global__ __launch_bounds__(64 * RED_GROUPS) void svae_bwd_reduce_kernel(RedArgs a) {
    const int L = a.L, TRI = L * (L + 1) / 2, TH = L + TRI + 1, PW = 2 * TH;
    const int half = a.g_mk ? PW : TH;
    __shared__ double part[RED_GROUPS][64];
    const int eg = threadIdx.x & 63, bg = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + eg;
    const bool live = e < a.K * half;
    const int k = live ? e / half : 0, f = live ? e - k * half : 0;
    // four independent chains per thread (their loads are in flight together: one chain of nblk / 16 dependent loads was most of
    // this launch at 512 blocks), combined in a fixed order
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (live) {
        const float* __restrict__ p = a.partials + (size_t)k * PW + f;
        const size_t bs = (size_t)a.K * PW;
        int b = bg;
        for (; b + 3 * RED_GROUPS < a.nblk; b += 4 * RED_GROUPS) {
            s0 += (double)p[(size_t)b * bs]; s1 += (double)p[(size_t)(b + RED_GROUPS) * bs];
            s2 += (double)p[(size_t)(b + 2 * RED_GROUPS) * bs]; s3 += (double)p[(size_t)(b + 3 * RED_GROUPS) * bs];
        }
        for (; b < a.nblk; b += RED_GROUPS) s0 += (double)p[(size_t)b * bs];
    }
    double s = (s0 + s1) + (s2 + s3);
    part[bg][eg] = s;
    __syncthreads();
    if (bg == 0 && live) {
        for (int g2 = 1; g2 < RED_GROUPS; ++g2) s += part[g2][eg];
        const float v = (float)s;
        const int g = f < TH ? f : f - TH;                  // position inside the phi-side / theta-side group
        const bool th = f >= TH;
        if (g < L) {
            (th ? a.g_mk : a.g_hk)[k * L + g] = v;
        } else if (g < L + TRI) {
            int idx = g - L, i = 0;
            while ((i + 1) * (i + 2) / 2 <= idx) ++i;
            const int j = idx - i * (i + 1) / 2;
            float* M = (th ? a.g_W : a.g_P) + (size_t)k * L * L;
            M[i * L + j] = v;
            if (i != j) M[j * L + i] = th ? 0.f : v;
        } else {
            (th ? a.g_kappa : a.g_bias)[k] = v;
        }
    }
}

int prep_check(const char* what, int K, int L) {
    if (K < 1 || K > VMP_MAX_K || L < 1 || L > VMP_MAX_D) {
        set_error("%s: K=%d, L=%d outside the compiled range (K <= %d, L <= %d)", what, K, L, VMP_MAX_K, VMP_MAX_D);
        return VMP_E_DIM;
    }
    return 0;
}

}  // namespace

extern "C" {

int vmp_svae_phi_prep_fwd(const float* mu_k, const float* L_raw, const float* pi_raw, int K, int L, float* Lk, float* P,
                          float* bias, void* stream) {
    if (int e = prep_check("vmp_svae_phi_prep_fwd", K, L)) return e;
    if (!mu_k || !L_raw || !pi_raw || !Lk || !P || !bias) { set_error("vmp_svae_phi_prep_fwd: NULL argument"); return VMP_E_BADARG; }
    PhiArgs a{};
    a.mu = mu_k; a.Lraw = L_raw; a.piraw = pi_raw; a.Lk = Lk; a.P = P; a.bias = bias; a.K = K; a.L = L;
#define PREP_CALL(LL) hipLaunchKernelGGL((phi_prep_kernel<LL, false>), dim3(K), dim3(PREP_THREADS), 0, static_cast<hipStream_t>(stream), a)
    PREP_DISPATCH_L(L, PREP_CALL)
#undef PREP_CALL
    return check_launch("vmp_svae_phi_prep_fwd");
}

int vmp_svae_phi_prep_bwd(const float* mu_k, const float* L_raw, const float* pi_raw, const float* g_hk, const float* g_P,
                          const float* g_bias, int K, int L, float* g_mu, float* g_Lraw, float* g_piraw, void* stream) {
    if (int e = prep_check("vmp_svae_phi_prep_bwd", K, L)) return e;
    if (!mu_k || !L_raw || !pi_raw || !g_hk || !g_P || !g_bias || !g_mu || !g_Lraw || !g_piraw) {
        set_error("vmp_svae_phi_prep_bwd: NULL argument");
        return VMP_E_BADARG;
    }
    PhiArgs a{};
    a.mu = mu_k; a.Lraw = L_raw; a.piraw = pi_raw; a.g_hk = g_hk; a.g_P = g_P; a.g_bias = g_bias;
    a.g_mu = g_mu; a.g_Lraw = g_Lraw; a.g_piraw = g_piraw; a.K = K; a.L = L;
#define PREP_CALL(LL) hipLaunchKernelGGL((phi_prep_kernel<LL, true>), dim3(K), dim3(PREP_THREADS), 0, static_cast<hipStream_t>(stream), a)
    PREP_DISPATCH_L(L, PREP_CALL)
#undef PREP_CALL
    return check_launch("vmp_svae_phi_prep_bwd");
}

int vmp_svae_bwd_reduce(const float* partials, int nblk, int K, int L, float* g_hk, float* g_P, float* g_bias, float* g_mk,
                        float* g_W, float* g_kappa, void* stream) {
    if (int e = prep_check("vmp_svae_bwd_reduce", K, L)) return e;
    if (!partials || nblk < 1 || !g_hk || !g_P || !g_bias || ((!g_mk) != (!g_W)) || ((!g_mk) != (!g_kappa))) {
        set_error("vmp_svae_bwd_reduce: bad argument");
        return VMP_E_BADARG;
    }
    RedArgs a{partials, g_hk, g_P, g_bias, g_mk, g_W, g_kappa, nblk, K, L};
    const int n = K * 2 * (L + L * (L + 1) / 2 + 1);
    hipLaunchKernelGGL(svae_bwd_reduce_kernel, dim3((n + 63) / 64), dim3(64 * RED_GROUPS), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("vmp_svae_bwd_reduce");
}

int vmp_svae_theta_pack(const float* alpha, const float* A, const float* b, const float* beta, const float* v_hat, int K,
                        int L, float* m, float* W, float* kappa, void* stream) {
    if (int e = prep_check("vmp_svae_theta_pack", K, L)) return e;
    if (!alpha || !A || !b || !beta || !v_hat || !m || !W || !kappa) { set_error("vmp_svae_theta_pack: NULL argument"); return VMP_E_BADARG; }
    ThetaArgs a{alpha, A, b, beta, v_hat, m, W, kappa, K, L};
#define PREP_CALL(LL) hipLaunchKernelGGL((theta_pack_kernel<LL>), dim3(K), dim3(PREP_THREADS), 0, static_cast<hipStream_t>(stream), a)
    PREP_DISPATCH_L(L, PREP_CALL)
#undef PREP_CALL
    return check_launch("vmp_svae_theta_pack");
}

int vmp_svae_prep_fwd(const float* mu_k, const float* L_raw, const float* pi_raw, const float* alpha, const float* A,
                      const float* b, const float* beta, const float* v_hat, int K, int L, float* Lk, float* P, float* bias,
                      float* m, float* W, float* kappa, void* stream) {
    if (int e = prep_check("vmp_svae_prep_fwd", K, L)) return e;
    if (!mu_k || !L_raw || !pi_raw || !Lk || !P || !bias || !alpha || !A || !b || !beta || !v_hat || !m || !W || !kappa) {
        set_error("vmp_svae_prep_fwd: NULL argument");
        return VMP_E_BADARG;
    }
    PhiArgs a{};
    a.mu = mu_k; a.Lraw = L_raw; a.piraw = pi_raw; a.Lk = Lk; a.P = P; a.bias = bias; a.K = K; a.L = L;
    ThetaArgs t{alpha, A, b, beta, v_hat, m, W, kappa, K, L};
#define PREP_CALL(LL) hipLaunchKernelGGL((prep_both_kernel<LL>), dim3(2 * K), dim3(PREP_THREADS), 0, static_cast<hipStream_t>(stream), a, t)
    PREP_DISPATCH_L(L, PREP_CALL)
#undef PREP_CALL
    return check_launch("vmp_svae_prep_fwd");
}

int vmp_svae_stats_cvi(const float* x_samples, const float* r, int64_t N, const float* p_alpha, const float* p_A, const float* p_b,
                       const float* p_beta, const float* p_vhat, float* t_alpha, float* t_A, float* t_b, float* t_beta,
                       float* t_vhat, float* s_alpha, float* s_A, float* s_b, float* s_beta, float* s_vhat,
                       const float* rho_dev, float rho, int K, int L, double* stats_out, void* stream) {
    if (int e = prep_check("vmp_svae_stats_cvi", K, L)) return e;
    if (N < 0 || N > SMALL_STATS_MAX_N) {
        set_error("vmp_svae_stats_cvi: N = %lld outside 0..%d (larger batches: vmp_mix_stats + vmp_svae_cvi_update)", (long long)N,
                  SMALL_STATS_MAX_N);
        return VMP_E_DIM;
    }
    if ((N > 0 && (!x_samples || !r)) || !stats_out || !p_alpha || !p_A || !p_b || !p_beta || !p_vhat || !t_alpha || !t_A || !t_b ||
        !t_beta || !t_vhat) {
        set_error("vmp_svae_stats_cvi: NULL argument");
        return VMP_E_BADARG;
    }
    SmallStatsArgs sa{x_samples, r, nullptr, stats_out, (int)N, L, K};
    CviArgs a{stats_out, p_alpha, p_A, p_b, p_beta, p_vhat, t_alpha, t_A, t_b, t_beta, t_vhat,
              s_alpha, s_A, s_b, s_beta, s_vhat, rho_dev, rho, K, L};
    hipLaunchKernelGGL(stats_cvi_kernel, dim3(K), dim3(SMALL_STATS_GROUPS * 80), 0, static_cast<hipStream_t>(stream), sa, a);
    return check_launch("vmp_svae_stats_cvi");
}

int vmp_svae_mom_cvi(const double* mom, int nblk, const float* p_alpha, const float* p_A, const float* p_b, const float* p_beta,
                     const float* p_vhat, float* t_alpha, float* t_A, float* t_b, float* t_beta, float* t_vhat,
                     float* s_alpha, float* s_A, float* s_b, float* s_beta, float* s_vhat, const float* rho_dev, float rho, int K,
                     int L, double* stats_out, void* stream) {
    if (int e = prep_check("vmp_svae_mom_cvi", K, L)) return e;
    if (K != 16 || L != MOM_L) { set_error("vmp_svae_mom_cvi: the in-kernel moments exist for K = 16, L = 8 (vmp_svae_fwd_mom_blocks)"); return VMP_E_DIM; }
    const bool cvi = t_alpha != nullptr;
    if (!mom || nblk < 1 || (!cvi && !stats_out) || (cvi && (!p_alpha || !p_A || !p_b || !p_beta || !p_vhat || !t_A || !t_b || !t_beta || !t_vhat))) {
        set_error("vmp_svae_mom_cvi: bad argument");
        return VMP_E_BADARG;
    }
    MomArgs m{mom, stats_out, nblk, K};
    CviArgs a{stats_out, p_alpha, p_A, p_b, p_beta, p_vhat, t_alpha, t_A, t_b, t_beta, t_vhat,
              s_alpha, s_A, s_b, s_beta, s_vhat, rho_dev, rho, K, L};
    hipLaunchKernelGGL(mom_cvi_kernel, dim3(K), dim3(64 * MOM_GROUPS), 0, static_cast<hipStream_t>(stream), m, a, cvi ? 1 : 0);
    return check_launch("vmp_svae_mom_cvi");
}

int vmp_svae_cvi_update(const double* stats, const float* p_alpha, const float* p_A, const float* p_b, const float* p_beta,
                        const float* p_vhat, float* t_alpha, float* t_A, float* t_b, float* t_beta, float* t_vhat,
                        float* s_alpha, float* s_A, float* s_b, float* s_beta, float* s_vhat, const float* rho_dev,
                        float rho, int K, int L, void* stream) {
    if (int e = prep_check("vmp_svae_cvi_update", K, L)) return e;
    if (!stats || !p_alpha || !p_A || !p_b || !p_beta || !p_vhat || !t_alpha || !t_A || !t_b || !t_beta || !t_vhat) {
        set_error("vmp_svae_cvi_update: NULL argument");
        return VMP_E_BADARG;
    }
    CviArgs a{stats, p_alpha, p_A, p_b, p_beta, p_vhat, t_alpha, t_A, t_b, t_beta, t_vhat,
              s_alpha, s_A, s_b, s_beta, s_vhat, rho_dev, rho, K, L};
    const int n = K * (L * L + L + 3);
    hipLaunchKernelGGL(cvi_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("vmp_svae_cvi_update");
}

}  // extern "C"
