// Expected diagonal-Gaussian reconstruction term of the ELBO (reference models/vae.py:201-250, weights branch
// :233-248) for gfx950:   A_nk = sum_{s,d} [ (y_nd - mu_nksd)^2 / var_nksd + log(var_nksd + 1e-8) ]
// (the reference's einsum 'nksd,nk->' then contracts A with the responsibilities; that N x K contraction and the
// constants are K-cheap and stay on the host side).  Pure streaming: the decoder outputs (N,K,S,Dy) x 2 are read
// once in the forward pass; the backward pass reads them again and writes the two gradients.
// Lane <-> one sample row (cell, s): consecutive lanes read consecutive Dy-float rows, i.e. fully coalesced.
#include "vmp_common.h"

using namespace vmp;

namespace {

struct LLArgs {
    const float* y;        // (N,Dy)
    const float* mean;     // (N,K,S,Dy)
    const float* var;      // (N,K,S,Dy)
    const float* gA;       // (N,K)   backward only
    float* A;              // (N,K)   forward only
    float* gmean;          // (N,K,S,Dy)
    float* gvar;           // (N,K,S,Dy)
    long long cells;       // N*K
    int K, S, Dy, vec_ok;
    float eps;             // log(var + eps): 1e-8 in the weights branch (vae.py:240), 0 in the plain-VAE branch (vae.py:225)
};

template <bool BWD>
__global__ __launch_bounds__(256) void loglike_kernel(LLArgs a) {
    __shared__ float scr[4][WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int S = a.S, Dy = a.Dy;
    const int SL = S < WAVE ? S : WAVE;           // lanes per cell
    const int CPT = WAVE / SL;                    // cells per wave tile
    const int c_in = lane / SL, sub = lane - c_in * SL;
    const bool lane_on = c_in < CPT;
    const long long ntiles = (a.cells + CPT - 1) / CPT;
    for (long long t = (long long)blockIdx.x * nw + wave; t < ntiles; t += (long long)gridDim.x * nw) {
        const long long cell = t * CPT + c_in;
        const bool on = lane_on && cell < a.cells;
        const long long n = on ? cell / a.K : 0;
        const float* __restrict__ yr = a.y + n * Dy;
        const float g = (BWD && on) ? a.gA[cell] : 0.f;
        float acc = 0.f;
        if (on) {
            for (int s = sub; s < S; s += SL) {
                const long long base = (cell * S + s) * Dy;
                if (a.vec_ok && (Dy & 3) == 0) {
                    for (int d = 0; d < Dy; d += 4) {
                        const float4 m = *reinterpret_cast<const float4*>(a.mean + base + d);
                        const float4 v = *reinterpret_cast<const float4*>(a.var + base + d);
                        const float4 yy = *reinterpret_cast<const float4*>(yr + d);
                        const float mm[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w}, y4[4] = {yy.x, yy.y, yy.z, yy.w};
                        float gm[4], gv[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float df = y4[q] - mm[q], iv = 1.0f / vv[q];
                            if (BWD) {
                                gm[q] = g * (-2.f * df * iv);
                                gv[q] = g * (1.0f / (vv[q] + a.eps) - df * df * iv * iv);
                            } else {
                                acc += df * df * iv + logf(vv[q] + a.eps);
                            }
                        }
                        if (BWD) {
                            *reinterpret_cast<float4*>(a.gmean + base + d) = make_float4(gm[0], gm[1], gm[2], gm[3]);
                            *reinterpret_cast<float4*>(a.gvar + base + d) = make_float4(gv[0], gv[1], gv[2], gv[3]);
                        }
                    }
                } else {
                    for (int d = 0; d < Dy; ++d) {
                        const float m = a.mean[base + d], v = a.var[base + d];
                        const float df = yr[d] - m, iv = 1.0f / v;
                        if (BWD) {
                            a.gmean[base + d] = g * (-2.f * df * iv);
                            a.gvar[base + d] = g * (1.0f / (v + a.eps) - df * df * iv * iv);
                        } else {
                            acc += df * df * iv + logf(v + a.eps);
                        }
                    }
                }
            }
        }
        if (!BWD) {
            scr[wave][lane] = acc;
            __builtin_amdgcn_wave_barrier();
            if (on && sub == 0) {
                float s2 = 0.f;
                for (int j = 0; j < SL; ++j) s2 += scr[wave][c_in * SL + j];
                a.A[cell] = s2;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Evaluation metrics, per-cell part (reference losses.py:9-38 weighted_mse, :83-145 diagonal_gaussian_logprob):
//   mse_nk = mean_s sum_d (y_nd - mean_nksd)^2
//   lse_nk = log( 1/S sum_s exp( lw_nk(s) - 1/2 sum_d m_nd [ (y-mean)^2/var + log var + log 2pi ] ) )
// (m = 1, or the missing-data mask).  The K-cheap contractions over k (sum_k r_nk mse_nk; log-sum-exp over k) and
// the means over n stay on the host side.  Same lane mapping as the reconstruction kernel above.
// ---------------------------------------------------------------------------------------------------------
struct EvArgs {
    const float* y;
    const float* mean;
    const float* var;       // may be NULL (mse only)
    const float* logw;      // (N,K) or (N,K,S) or NULL
    const uint8_t* mask;    // (N,Dy) or NULL
    float* mse;             // (N,K) or NULL
    float* lse;             // (N,K) or NULL
    long long cells;
    int K, S, Dy, logw_per_sample, mask_mse;
};

__global__ __launch_bounds__(256) void eval_kernel(EvArgs a) {
    __shared__ float s_mx[4][WAVE], s_se[4][WAVE], s_sq[4][WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int S = a.S, Dy = a.Dy;
    const int SL = S < WAVE ? S : WAVE;
    const int CPT = WAVE / SL;
    const int c_in = lane / SL, sub = lane - c_in * SL;
    const bool lane_on = c_in < CPT;
    const long long ntiles = (a.cells + CPT - 1) / CPT;
    const float LOG2PI = 1.8378770664093454836f;
    for (long long t = (long long)blockIdx.x * nw + wave; t < ntiles; t += (long long)gridDim.x * nw) {
        const long long cell = t * CPT + c_in;
        const bool on = lane_on && cell < a.cells;
        const long long cc = on ? cell : 0;
        const long long n = cc / a.K;
        const float* __restrict__ yr = a.y + n * Dy;
        float mx = -INFINITY, se = 0.f, sq = 0.f;
        if (on) {
            for (int s = sub; s < S; s += SL) {
                const long long base = (cc * S + s) * Dy;
                float q = 0.f, lp = 0.f;
                for (int d = 0; d < Dy; ++d) {
                    const float df = yr[d] - a.mean[base + d];
                    const float m = a.mask ? (a.mask[n * Dy + d] ? 1.f : 0.f) : 1.f;
                    q = fmaf(a.mask_mse ? m * df : df, df, q);
                    if (a.var) {
                        const float v = a.var[base + d];
                        const float term = df * df / v + logf(v) + LOG2PI;
                        lp = fmaf(-0.5f * m, term, lp);
                    }
                }
                sq += q;
                if (a.var) {
                    if (a.logw) lp += a.logw_per_sample ? a.logw[cc * S + s] : a.logw[cc];
                    const float nm = fmaxf(mx, lp);                 // online log-sum-exp
                    se = se * __expf(mx - nm) + __expf(lp - nm);
                    mx = nm;
                }
            }
        }
        s_mx[wave][lane] = mx; s_se[wave][lane] = se; s_sq[wave][lane] = sq;
        __builtin_amdgcn_wave_barrier();
        if (on && sub == 0) {
            float M = -INFINITY, Q = 0.f;
            for (int j = 0; j < SL; ++j) { M = fmaxf(M, s_mx[wave][c_in * SL + j]); Q += s_sq[wave][c_in * SL + j]; }
            if (a.mse) a.mse[cell] = Q / (float)S;
            if (a.lse) {
                float E = 0.f;
                for (int j = 0; j < SL; ++j) {
                    const float mj = s_mx[wave][c_in * SL + j];
                    if (mj > -INFINITY) E += s_se[wave][c_in * SL + j] * __expf(mj - M);
                }
                a.lse[cell] = M + logf(E) - logf((float)S);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------
// Bernoulli decoder (reference models/vae.py:175-198 expected_bernoulli_loglike, losses.py:41-80
// bernoulli_logprob): per sample row (n,k,s)   rows_nks = sum_d m_nd * ( -log(1 + exp(-logit_nksd * y_nd)) ),
// y in {-1,+1}, m = 1 or the missing-data mask; evaluated as -softplus(-logit*y) (the reference's naive form
// overflows for logit*y << 0).  One wave per row when D >= 32 (lanes stride over d, coalesced), else one lane per row.
// ---------------------------------------------------------------------------------------------------------
struct BernArgs {
    const float* y;          // (N,D)
    const float* logits;     // (R,D), R = N*K*S
    const uint8_t* mask;     // (N,D) or NULL
    const float* grow;       // (R) backward
    float* rows;             // (R) forward
    float* glogits;          // (R,D) backward
    long long R;
    int KS, D;
};

__device__ __forceinline__ float neg_softplus(float z) {           // -log(1 + exp(z))
    return -(fmaxf(z, 0.f) + log1p_f(__expf(-fabsf(z))));
}

template <bool BWD>
__global__ __launch_bounds__(256) void bern_kernel(BernArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int D = a.D;
    if (D >= 32) {
        for (long long r = (long long)blockIdx.x * nw + wave; r < a.R; r += (long long)gridDim.x * nw) {
            const long long n = r / a.KS;
            const float* __restrict__ lg = a.logits + r * D;
            const float* __restrict__ yr = a.y + n * D;
            const float g = BWD ? a.grow[r] : 0.f;
            float acc = 0.f;
            for (int d = lane; d < D; d += WAVE) {
                const float m = a.mask ? (a.mask[n * D + d] ? 1.f : 0.f) : 1.f;
                const float z = -lg[d] * yr[d];
                if (BWD) a.glogits[r * D + d] = g * m * yr[d] / (1.0f + __expf(-z));     // y * sigmoid(-logit*y)
                else acc += m * neg_softplus(z);
            }
            if (!BWD) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
                if (lane == 0) a.rows[r] = acc;
            }
        }
    } else {
        for (long long r = ((long long)blockIdx.x * nw + wave) * WAVE + lane; r < a.R; r += (long long)gridDim.x * nw * WAVE) {
            const long long n = r / a.KS;
            const float g = BWD ? a.grow[r] : 0.f;
            float acc = 0.f;
            for (int d = 0; d < D; ++d) {
                const float m = a.mask ? (a.mask[n * D + d] ? 1.f : 0.f) : 1.f;
                const float yv = a.y[n * D + d];
                const float z = -a.logits[r * D + d] * yv;
                if (BWD) a.glogits[r * D + d] = g * m * yv / (1.0f + __expf(-z));
                else acc += m * neg_softplus(z);
            }
            if (!BWD) a.rows[r] = acc;
        }
    }
}

int bern_launch(BernArgs a, bool bwd, hipStream_t s) {
    long long units = a.D >= 32 ? a.R : (a.R + WAVE - 1) / WAVE;
    long long blocks = (units + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    if (bwd) hipLaunchKernelGGL((bern_kernel<true>), dim3((int)blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((bern_kernel<false>), dim3((int)blocks), dim3(256), 0, s, a);
    return check_launch("bern_kernel");
}

bool al16b(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int ll_launch(LLArgs a, bool bwd, hipStream_t s) {
    const int SL = a.S < WAVE ? a.S : WAVE, CPT = WAVE / SL;
    long long ntiles = (a.cells + CPT - 1) / CPT;
    long long blocks = (ntiles + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (bwd) hipLaunchKernelGGL((loglike_kernel<true>), dim3((int)blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((loglike_kernel<false>), dim3((int)blocks), dim3(256), 0, s, a);
    return check_launch("loglike_kernel");
}

}  // namespace

extern "C" {

int vmp_diag_gauss_loglike_fwd(const float* y, const float* mean, const float* var, int64_t N, int K, int S, int Dy,
                               float eps, float* A, void* stream) {
    if (!y || !mean || !var || !A || N <= 0 || K <= 0 || S <= 0 || Dy <= 0) { set_error("vmp_diag_gauss_loglike_fwd: bad argument"); return VMP_E_BADARG; }
    LLArgs a{y, mean, var, nullptr, A, nullptr, nullptr, (long long)N * K, K, S, Dy, 0, eps};
    a.vec_ok = al16b(y) && al16b(mean) && al16b(var);
    return ll_launch(a, false, static_cast<hipStream_t>(stream));
}

int vmp_diag_gauss_loglike_bwd(const float* y, const float* mean, const float* var, const float* gA, int64_t N, int K,
                               int S, int Dy, float eps, float* gmean, float* gvar, void* stream) {
    if (!y || !mean || !var || !gA || !gmean || !gvar || N <= 0 || K <= 0 || S <= 0 || Dy <= 0) { set_error("vmp_diag_gauss_loglike_bwd: bad argument"); return VMP_E_BADARG; }
    LLArgs a{y, mean, var, gA, nullptr, gmean, gvar, (long long)N * K, K, S, Dy, 0, eps};
    a.vec_ok = al16b(y) && al16b(mean) && al16b(var) && al16b(gmean) && al16b(gvar);
    return ll_launch(a, true, static_cast<hipStream_t>(stream));
}

int vmp_eval_cell_metrics(const float* y, const float* mean, const float* var, const float* logw, int logw_per_sample,
                          const uint8_t* mask, int mask_mse, int64_t N, int K, int S, int Dy, float* mse, float* lse,
                          void* stream) {
    if (!y || !mean || N <= 0 || K <= 0 || S <= 0 || Dy <= 0 || (!mse && !lse) || (lse && !var)) {
        set_error("vmp_eval_cell_metrics: bad argument");
        return VMP_E_BADARG;
    }
    EvArgs a{y, mean, var, logw, mask, mse, lse, (long long)N * K, K, S, Dy, logw_per_sample, (mask && mask_mse) ? 1 : 0};
    const int SL = S < WAVE ? S : WAVE, CPT = WAVE / SL;
    long long blocks = (((long long)N * K + CPT - 1) / CPT + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(eval_kernel, dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("eval_kernel");
}

int vmp_bernoulli_rows_fwd(const float* y, const float* logits, const uint8_t* mask, int64_t N, int K, int S, int D,
                           float* rows, void* stream) {
    if (!y || !logits || !rows || N <= 0 || K <= 0 || S <= 0 || D <= 0) { set_error("vmp_bernoulli_rows_fwd: bad argument"); return VMP_E_BADARG; }
    BernArgs a{y, logits, mask, nullptr, rows, nullptr, (long long)N * K * S, K * S, D};
    return bern_launch(a, false, static_cast<hipStream_t>(stream));
}

int vmp_bernoulli_rows_bwd(const float* y, const float* logits, const uint8_t* mask, const float* g_rows, int64_t N, int K,
                           int S, int D, float* g_logits, void* stream) {
    if (!y || !logits || !g_rows || !g_logits || N <= 0 || K <= 0 || S <= 0 || D <= 0) { set_error("vmp_bernoulli_rows_bwd: bad argument"); return VMP_E_BADARG; }
    BernArgs a{y, logits, mask, g_rows, nullptr, g_logits, (long long)N * K * S, K * S, D};
    return bern_launch(a, true, static_cast<hipStream_t>(stream));
}

}  // extern "C"
