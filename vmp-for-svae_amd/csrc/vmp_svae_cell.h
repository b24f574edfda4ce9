// Shared by the T2 translation units (vmp_svae.hip: forward, generic backward, sub-sampling, C ABI; vmp_svae_ring.hip: the
// LDS-ring backward kernels): per-cell linear algebra of the SVAE E-step (reference models/svae.py:14-119 and its autodiff)
// and the argument block of the backward kernels.  See the header comment of vmp_svae.hip for the cell arithmetic.
#pragma once
#include "vmp_common.h"

namespace vmp {
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
    // minibatch form with the ELBO's scalar tail inside (vmp_svae_estep_bwd_tail; Glz / GT unused): dLoss/dlog_z and dLoss/dT' are
    // formed per cell from (lz, Tp, ll) exactly as elbo_tail_body does (vmp_tail.h)
    const float* Tp;        // (N,K)
    const float* ll;        // (N,K,S) per-sample reconstruction sums of the decoder kernel
    float* r_out;           // (N,K) exp(log z)
    double* tail_part;      // (tiles, 2) per-tile terms of the ELBO's two fp64 sums
    float sigma;
#ifdef VMP_DEBUG_TS
    long long* dbg_t;       // exploration builds only (tools/build_variant.sh ts -DVMP_DEBUG_TS): stage time stamps of block 0, wave 0
#endif
};
}  // namespace vmp

namespace {
using namespace vmp;

constexpr int SV_AST = 65;          // row stride of the generic backward kernel's LDS accumulators (see there)
constexpr int SV_NW = 4;            // waves per block (backward); forward: as many as the LDS noise tiles allow
constexpr int SV_FWD_MAX_NW = 8;
constexpr int SV_MAX_BLOCKS = 2048;
constexpr float LOG_2PI = 1.8378770664093454836f;

template <int L>
struct SvGeo {
    static constexpr int TRI = L * (L + 1) / 2;
};

// lower-triangular packed index (row-major), i >= j
__host__ __device__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// Cholesky of the cell matrix (lower, packed).  On return Lm holds the factor with the DIAGONAL REPLACED BY ITS
// RECIPROCAL rd_j = 1/Lt_jj (every later use multiplies by it), and half_logdet = sum_j log Lt_jj.
// Right-looking form: once column j is final, every remaining entry takes its update -L_ij L_qj at once.  Each entry still
// receives its terms in the order p = 0, 1, 2, ... (bitwise the same result as the left-looking loop), but the updates
// of one column are mutually independent, so the dependency chain is L steps deep instead of ~L^2/2 - with two waves
// per SIMD that is what the VALU waits on (SQ_WAIT_INST_ANY 30 % of the backward kernel in round 1).
template <int L>
__device__ __forceinline__ void cell_cholesky(float (&Lm)[SvGeo<L>::TRI], float& half_logdet) {
    float prod_log = 0.f;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const float s = Lm[tri(j, j)];
        const float rd = __builtin_amdgcn_rsqf(s);
        prod_log += __logf(s);
#pragma unroll
        for (int i = j + 1; i < L; ++i) Lm[tri(i, j)] *= rd;
        Lm[tri(j, j)] = rd;
#pragma unroll
        for (int i = j + 1; i < L; ++i)
#pragma unroll
            for (int q = j + 1; q <= i; ++q) Lm[tri(i, q)] = fmaf(-Lm[tri(i, j)], Lm[tri(q, j)], Lm[tri(i, q)]);
    }
    half_logdet = 0.5f * prod_log;
}

// v <- Lt^-1 v   (forward substitution, column-oriented: v_j final -> all v_i, i > j, updated independently; the terms
// reach every v_i in the same order p = 0, 1, ... as in the row-oriented loop; diagonal of Lm holds reciprocals)
template <int L>
__device__ __forceinline__ void solve_lower(const float (&Lm)[SvGeo<L>::TRI], float (&v)[L]) {
#pragma unroll
    for (int j = 0; j < L; ++j) {
        v[j] *= Lm[tri(j, j)];
#pragma unroll
        for (int i = j + 1; i < L; ++i) v[i] = fmaf(-Lm[tri(i, j)], v[j], v[i]);
    }
}

// v <- Lt^-T v   (back substitution, column-oriented; terms reach v_i in the order p = L-1, L-2, ...: the row-oriented
// loop summed p = i+1, ..., L-1, so the rounding differs in the last bits)
template <int L>
__device__ __forceinline__ void solve_lower_t(const float (&Lm)[SvGeo<L>::TRI], float (&v)[L]) {
#pragma unroll
    for (int j = L - 1; j >= 0; --j) {
        v[j] *= Lm[tri(j, j)];
#pragma unroll
        for (int i = 0; i < j; ++i) v[i] = fmaf(-Lm[tri(j, i)], v[j], v[i]);
    }
}

// sum / max over the K lanes of this lane's row, through a 64-float LDS scratch
__device__ __forceinline__ float row_sum(float v, float* scr, int lane, int rbase, int K) {
    if (K == 16) return row16_sum(v);            // a row = one 16-lane DPP row: 4 rotations instead of 17 LDS accesses + 16 adds
    scr[lane] = v;
    __builtin_amdgcn_wave_barrier();
    float s = 0.f;
#pragma unroll 4                                     // reads issued ahead of the (in-order) adds: one LDS latency per 4 values, not per value
    for (int j = 0; j < K; ++j) s += scr[rbase + j];
    __builtin_amdgcn_wave_barrier();
    return s;
}
__device__ __forceinline__ float row_max(float v, float* scr, int lane, int rbase, int K) {
    if (K == 16) return row16_max(v);
    scr[lane] = v;
    __builtin_amdgcn_wave_barrier();
    float m = -INFINITY;
#pragma unroll 4
    for (int j = 0; j < K; ++j) m = fmaxf(m, scr[rbase + j]);
    __builtin_amdgcn_wave_barrier();
    return m;
}

// =========================================================================================================
// backward
// =========================================================================================================
#ifdef VMP_DEBUG_TS
#define SV_TS(i) do { if (a.dbg_t && blockIdx.x == 0 && threadIdx.x == 0) { a.dbg_t[i] = clock64(); a.dbg_t[32 + (i)] = wall_clock64(); } } while (0)
#define SV_USE(v) asm volatile("" :: "v"(v))
#else
#define SV_TS(i) do { } while (0)
#define SV_USE(v) do { } while (0)
#endif

constexpr int SVR_NW = 8;
template <int L> constexpr int svr_stage_floats() { return 2 * WAVE * 2 * L; }       // x pair + dx pair of 64 cells


}  // namespace

namespace vmp {
// LDS-ring backward (vmp_svae_ring.hip): returns -2 when the shape is not covered (the caller takes the generic kernel),
// otherwise the launch status.  nblk_abi = number of partial rows the ABI sized the buffer for (rows the ring grid does not
// write are zeroed by the kernel).
int svae_bwd_ring_launch(const EBwdArgs& a, int L, int nblk_abi, void* stream);
int svae_bwd1_launch(const EBwdArgs& a, int L, int ntiles, int P, bool tail, void* stream);      // vmp_svae_mini.hip: one block per tile, one wave per sample pair
}
