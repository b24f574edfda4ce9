// Pieces of the training step's K- and parameter-sized tail that more than one launch uses (round 6: the minibatch step runs them
// inside ONE launch, vmp_step.hip step_final_kernel; the stand-alone launches in vmp_decoder.hip / vmp_prep.hip / vmp_step.hip are the
// same device functions, so that the fused launch reproduces them bit for bit).
#pragma once
#include "vmp_common.h"

namespace vmp {

// ---- tf.train.AdamOptimizer's update of one element (TF 1.3: epsilon outside the bias correction; lr_t carries the correction)
__device__ __forceinline__ void adam_update(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const unsigned i,
                                            const float gi, const float lr_t, const float b1, const float b2, const float c1, const float c2,
                                            const float eps) {
    // every operation rounded on its own: no contraction into FMAs, which the compiler picks differently from kernel to kernel (HIP's
    // __fmul_rn / __fadd_rn are plain operators and contract like them) - the update is the same bits wherever this is inlined, and
    // what a numpy fp32 restatement computes
#pragma clang fp contract(off)
    const float mi = m[i] * b1 + c1 * gi;
    const float vi = v[i] * b2 + c2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * (mi / (__fsqrt_rn(vi) + eps));
}

// ---- reduction of the fused MLP backward kernel's per-block parameter partials ((blocks, PW) fp32): 64 consecutive parameters per
// block of 64 * DEC_RED_GROUPS threads; lane group bg = tid / 64 sums the block rows b = bg, bg + 16, .. (independent coalesced loads
// instead of one chain of `blocks` dependent ones), the 16 group sums are added in a fixed order.  Returns the sum of parameter
// i = blk * 64 + (tid & 63) in the threads of group 0 (others: garbage); the sigmoid(bs2) factor of d/d bs2 log1p(exp(bs2)) included.
constexpr int DEC_RED_GROUPS = 16;
struct DecRedArgs {
    const float* part;
    const float* bs2;
    float* out;
    int blocks, PW, obs2, Dy;
};
__device__ __forceinline__ double dec_reduce_sum(const DecRedArgs& r, const int blk, double (*part)[64]) {
    const int eg = threadIdx.x & 63, bg = threadIdx.x >> 6;
    const int i = blk * 64 + eg;
    double s = 0.0;
    if (i < r.PW && bg < DEC_RED_GROUPS)
        for (int b = bg; b < r.blocks; b += DEC_RED_GROUPS) s += (double)r.part[(size_t)b * r.PW + i];
    if (bg < DEC_RED_GROUPS) part[bg][eg] = s;
    __syncthreads();
    if (bg != 0 || i >= r.PW) return 0.0;
    for (int g2 = 1; g2 < DEC_RED_GROUPS; ++g2) s += part[g2][eg];
    if (i >= r.obs2) s *= 1.0 / (1.0 + exp(-(double)r.bs2[i - r.obs2]));
    return s;
}

// ---- svae.m_step + update_gmm_params from raw moments (reference svae.py:154-176, 376-403)
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
#pragma clang fp contract(off)
    if (s) s[i] = star;
    t[i] = t[i] * (1.0f - rho) + rho * star;       // update_gmm_params: theta <- (1-rho) theta + rho theta* (no contraction: the same bits in every kernel)
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

// M-step moments of a small batch AND the CVI update of component k by one block of >= SMALL_STATS_GROUPS * 80 threads
// (small_stats_component, vmp_common.h): sums component k's moments, publishes them, updates theta_k from the copy in LDS.
// (do_cvi = false: the moments only - a data-parallel step sums them over the ranks first)
__device__ __forceinline__ void stats_cvi_body(const SmallStatsArgs& sa, const CviArgs& a, const int k, double (*part)[80], double* st,
                                               const bool do_cvi = true) {
    const int L = a.L, SW = 2 + L + L * L, i = threadIdx.x % 80;
    const double t = small_stats_component(sa, k, part);
    if (threadIdx.x < 80 && i < SW) {
        st[i] = t;
        sa.stats[(long long)k * SW + i] = t;
    }
    if (!do_cvi) return;
    __syncthreads();
    const float rho = a.rho_dev ? *a.rho_dev : a.rho;
    for (int f = threadIdx.x; f < L * L + L + 3; f += blockDim.x) cvi_element(a, st, k, f, rho);
}

}  // namespace vmp
