// K-sized parameter maps of the SVAE training step (reference models/svae.py:342-358 unpack_recognition_gmm,
// :205-214 the theta side of compute_elbo via distributions/niw.py:8-43 + dirichlet.py:8-22, :154-176 m_step and
// :376-403 update_gmm_params) as single-launch kernels: one 64-lane block per mixture component, fp64 inside, fp32 in/out.
// At the reference's real operating point (minibatches of 64-100 rows) the step is bound by the NUMBER of launches:
// these maps were ~120 tiny torch kernels (batched Cholesky / triangular solves with host-side error checks, tril,
// softplus, digamma ...) and their autograd; here they are 4 launches.
#include "vmp_common.h"
#include "vmp_step_parts.h"
#include "vmp_prep_parts.h"

using namespace vmp;

namespace {

// L is a template parameter everywhere: with run-time loop bounds the per-thread matrices live in scratch memory and
// every element access is a dependent round trip (measured: 50 us per launch instead of a few).
#define PREP_DISPATCH_L(L, CALL)                                                     \
    switch (L) {                                                                     \
        case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; \
        case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; default: CALL(8); break; \
    }

// Block per component, lane (i, j) per matrix element as in phi_prep_kernel: the two digammas run side by side in lanes 0
// and 1, the Cholesky factor is built column by column in LDS (rows in parallel), the columns of its inverse in parallel.
template <int L, bool BWD>
__global__ __launch_bounds__(PREP_THREADS) void phi_prep_kernel(PhiArgs a) { phi_prep_body<L, BWD>(a, blockIdx.x); }
template <int L>
__global__ __launch_bounds__(64 * RED_GROUPS) void bwd_reduce_prep_kernel(PhiArgs a, const float* partials, int nblk) {
    phi_prep_body<L, true, true>(a, blockIdx.x, partials, nblk);
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
    stats_cvi_body(sa, a, blockIdx.x, part, st);
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
__global__ __launch_bounds__(64 * RED_GROUPS) void svae_bwd_reduce_kernel(RedArgs a) {
    const int L = a.L, TRI = L * (L + 1) / 2, TH = L + TRI + 1, PW = 2 * TH;
    const int half = a.g_mk ? PW : TH;
    __shared__ double part[RED_GROUPS][64];
    const int eg = threadIdx.x & 63, bg = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + eg;
    const bool live = e < a.K * half;
    const int k = live ? e / half : 0, f = live ? e - k * half : 0;
    // four independent chains per thread (red_group_sum: one chain of nblk / 16 dependent loads was most of this launch at 512 blocks)
    double s = live ? red_group_sum(a.partials + (size_t)k * PW + f, (size_t)a.K * PW, a.nblk, bg) : 0.0;
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

// round 6: vmp_svae_bwd_reduce (phi side) + vmp_svae_phi_prep_bwd in one launch - block k sums component k's partials and runs the
// backward of the recognition unpacking on them (the minibatch step runs the same body inside vmp_svae_step_final)
int vmp_svae_bwd_reduce_prep(const float* partials, int nblk, const float* mu_k, const float* L_raw, const float* pi_raw,
                             const double* logpi, int K, int L, float* g_mu, float* g_Lraw, float* g_piraw, void* stream) {
    if (int e = prep_check("vmp_svae_bwd_reduce_prep", K, L)) return e;
    if (!partials || !mu_k || !L_raw || !pi_raw || !logpi || !g_mu || !g_Lraw || !g_piraw || nblk < 1) {
        set_error("vmp_svae_bwd_reduce_prep: NULL argument or nblk < 1");
        return VMP_E_BADARG;
    }
    PhiArgs a{};
    a.mu = mu_k; a.Lraw = L_raw; a.piraw = pi_raw; a.g_mu = g_mu; a.g_Lraw = g_Lraw; a.g_piraw = g_piraw; a.K = K; a.L = L; a.logpi = logpi;
#define PREP_CALL(LL) hipLaunchKernelGGL((bwd_reduce_prep_kernel<LL>), dim3(K), dim3(64 * RED_GROUPS), 0, static_cast<hipStream_t>(stream), a, partials, nblk)
    PREP_DISPATCH_L(L, PREP_CALL)
#undef PREP_CALL
    return check_launch("vmp_svae_bwd_reduce_prep");
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

int vmp_svae_prep_fwd2(const float* mu_k, const float* L_raw, const float* pi_raw, const float* alpha, const float* A,
                       const float* b, const float* beta, const float* v_hat, int K, int L, float* Lk, float* P, float* bias,
                       float* m, float* W, float* kappa, double* logpi, void* stream) {
    if (int e = prep_check("vmp_svae_prep_fwd", K, L)) return e;
    if (!mu_k || !L_raw || !pi_raw || !Lk || !P || !bias || !alpha || !A || !b || !beta || !v_hat || !m || !W || !kappa) {
        set_error("vmp_svae_prep_fwd: NULL argument");
        return VMP_E_BADARG;
    }
    PhiArgs a{};
    a.mu = mu_k; a.Lraw = L_raw; a.piraw = pi_raw; a.Lk = Lk; a.P = P; a.bias = bias; a.K = K; a.L = L; a.logpi_out = logpi;
    ThetaArgs t{alpha, A, b, beta, v_hat, m, W, kappa, K, L};
#define PREP_CALL(LL) hipLaunchKernelGGL((prep_both_kernel<LL>), dim3(2 * K), dim3(PREP_THREADS), 0, static_cast<hipStream_t>(stream), a, t)
    PREP_DISPATCH_L(L, PREP_CALL)
#undef PREP_CALL
    return check_launch("vmp_svae_prep_fwd");
}

int vmp_svae_prep_fwd(const float* mu_k, const float* L_raw, const float* pi_raw, const float* alpha, const float* A,
                      const float* b, const float* beta, const float* v_hat, int K, int L, float* Lk, float* P, float* bias,
                      float* m, float* W, float* kappa, void* stream) {
    return vmp_svae_prep_fwd2(mu_k, L_raw, pi_raw, alpha, A, b, beta, v_hat, K, L, Lk, P, bias, m, W, kappa, nullptr, stream);
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
