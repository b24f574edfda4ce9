// Scalar tail of the SVAE ELBO as a device function (see vmp_step.hip): shared by the stand-alone launch
// (vmp_svae_elbo_tail) and the launch that runs it beside the decoder's partial reduction (vmp_decoder_elbo).
#pragma once
#include "vmp_common.h"

namespace vmp {

constexpr int TAIL_MAX_BLOCKS = 1024;
constexpr int TAIL_MAX_WAVES = 16;      // block sizes up to 1024 threads

struct TailArgs {
    const float* lz;      // (NK)
    const float* Tp;      // (NK)
    const float* ll;      // (NK, S) per-sample reconstruction sums of the decoder kernel
    float* g_lz;          // (NK)  sigma * d elbo / d log z
    float* g_Tp;          // (NK)  sigma * d elbo / d T'
    float* r;             // (NK)  exp(log z)
    float* scal;          // [elbo, rec, reg]
    double* part;         // (blocks, 2)
    long long NK;
    int S;
    float sigma;
    double cst;           // N Dy / 2 log(2 pi)
};

__device__ __forceinline__ double tail_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one cell: r = exp(log z), the two gradient seeds and the cell's terms of the two fp64 sums (hs = 1 / (2 S), A = sum_s ll_s)
__device__ __forceinline__ void tail_cell(const float lz, const float tp, const float A, const float hs, const float sigma, float& r,
                                          float& g_lz, float& g_Tp, double& wa, double& rg) {
    // (no contraction here: the same roundings in every kernel this is inlined into, whatever the compiler would pick there)
#pragma clang fp contract(off)
    r = expf(lz);
    const float w = hs * r;
    wa = fma((double)w, (double)A, wa);
    rg = fma((double)r, (double)(tp + lz), rg);
    const float t1 = r * (tp + lz + 1.0f);
    g_lz = -sigma * fmaf(w, A, t1);
    g_Tp = -sigma * r;
}

// block tb of ntb (any block size that is a multiple of 64, <= 1024)
__device__ __forceinline__ void elbo_tail_body(const TailArgs& a, const unsigned tb, const unsigned ntb) {
    __shared__ double sm[2][TAIL_MAX_WAVES];
    const float hs = 0.5f / (float)a.S;
#ifndef VMP_TAIL_LL2
#define VMP_TAIL_LL2 1
#endif
    const bool ll2 = VMP_TAIL_LL2 && (a.S & 1) == 0 && (reinterpret_cast<uintptr_t>(a.ll) & 7) == 0;
    double wa = 0.0, rg = 0.0;
    for (long long c = (long long)tb * blockDim.x + threadIdx.x; c < a.NK; c += (long long)ntb * blockDim.x) {
        const float lz = a.lz[c], tp = a.Tp[c];
        const float* __restrict__ lr = a.ll + c * a.S;
        float A = 0.f;
        if (ll2) {                                                  // even S, 8-byte aligned: half the load instructions, same order
            const float2* __restrict__ l2 = reinterpret_cast<const float2*>(lr);
            for (int s = 0; s < (a.S >> 1); ++s) { const float2 v = l2[s]; A += v.x; A += v.y; }
        } else {
            for (int s = 0; s < a.S; ++s) A += lr[s];
        }
        float r, glz, gtp;
        tail_cell(lz, tp, A, hs, a.sigma, r, glz, gtp, wa, rg);
        a.r[c] = r;
        a.g_lz[c] = glz;
        a.g_Tp[c] = gtp;
    }
    wa = tail_wave_sum(wa);
    rg = tail_wave_sum(rg);
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE, nwv = blockDim.x / WAVE;
    if (lane == 0) { sm[0][wave] = wa; sm[1][wave] = rg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s0 = 0.0, s1 = 0.0;
        for (int i = 0; i < nwv; ++i) { s0 += sm[0][i]; s1 += sm[1][i]; }
        a.part[2 * tb] = s0;
        a.part[2 * tb + 1] = s1;
        if (ntb == 1) {
            // a batch of ONE tail block (N K <= the block size: the reference's minibatches): this block holds the whole sums - the
            // three scalars are written here and the host skips the final launch (one graph node less per training step).  The same
            // arithmetic as elbo_final_body on one partial pair.
            const double rec = -s0 - a.cst;
            a.scal[0] = (float)(rec - s1);
            a.scal[1] = (float)rec;
            a.scal[2] = (float)s1;
        }
    }
}

// The per-block partials are summed by a SEPARATE single-wave launch (elbo_final_body), i.e. behind a kernel boundary.
// A one-launch form - last block by ticket, partials published with device-scope atomics - was in use for a few hours: it
// passed 5 000 repetitions of a determinism stress and then produced ONE wrong ELBO (1.2 % off: a block's partial of the
// PREVIOUS launch) in the middle of a long test session.  The stress could not see it (same inputs every repetition: stale
// partials equal fresh ones).  One more 2 us launch is the price of not depending on cross-XCD visibility inside a kernel.
__device__ __forceinline__ void elbo_final_body(const TailArgs& a, const unsigned ntb) {
    const int lane = threadIdx.x;                                   // one wave
    double s0 = 0.0, s1 = 0.0;
    for (unsigned b = lane; b < ntb; b += WAVE) { s0 += a.part[2 * b]; s1 += a.part[2 * b + 1]; }   // fixed assignment: deterministic
    s0 = tail_wave_sum(s0);
    s1 = tail_wave_sum(s1);
    if (lane == 0) {
        const double rec = -s0 - a.cst;
        a.scal[0] = (float)(rec - s1);
        a.scal[1] = (float)rec;
        a.scal[2] = (float)s1;
    }
}

inline size_t tail_workspace_bytes() { return (size_t)TAIL_MAX_BLOCKS * 2 * sizeof(double); }

// fills a TailArgs and returns the number of blocks of `threads` threads to run it on
inline unsigned tail_setup(TailArgs& a, const float* log_z, const float* T_prime, const float* ll, long long N, int K, int S, int Dy,
                           float sigma, float* scalars, float* g_log_z, float* g_T_prime, float* r, void* ws, int threads) {
    a.lz = log_z; a.Tp = T_prime; a.ll = ll; a.g_lz = g_log_z; a.g_Tp = g_T_prime; a.r = r; a.scal = scalars;
    a.part = static_cast<double*>(ws);
    a.NK = N * K; a.S = S; a.sigma = sigma;
    a.cst = (double)N * Dy * 0.5 * 1.8378770664093453;             // log(2 pi)
    long long blocks = (a.NK + threads - 1) / threads;
    if (blocks > TAIL_MAX_BLOCKS) blocks = TAIL_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace vmp
