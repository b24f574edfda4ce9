// Fused decoder MLP + expected diagonal-Gaussian log-likelihood for gfx950 (reference models/vae.py:75-128
// make_nnet with layerspecs [(U,tanh),(U,tanh),(Dy,'standard')] - experiments.py:140 - followed by the per-row part
// of vae.expected_diagonal_gaussian_loglike, vae.py:233-248):
//
//   h0 = tanh(x W0 + b0);  h1 = tanh(h0 W1 + b1);  [raw1 | raw2] = h1 W2 + b2
//   mean = raw1 + x Ws + bs1;  var = softplus(raw2) + log1p(exp(bs2))
//   ll_row = sum_d [ (y_nd - mean_d)^2 / var_d + log(var_d + 1e-8) ]             row = (n, k, s)
//
// The reference evaluates this on N*K*S rows with three tiny GEMMs whose (rows x U) activations go through memory
// five times forward and as often backward.  Here one wave owns 16 rows at a time and the activations never leave
// its registers:
//   * every layer is a chain of v_mfma_f32_16x16x4_f32 (exact fp32) with M = output unit, N = data row,
//     K = input unit.  The accumulator layout of layer l (lane (g,c), register v  <->  unit 16t+4g+v, row c) IS the
//     B-operand layout of layer l+1 when its k-steps are enumerated as (t, v) - the contraction order is free - so
//     the layers chain without any transpose or LDS round trip; the weights are pre-permuted into that order once
//     per block ("operand images" in LDS, read as conflict-free ds_read_b128).
//   * the backward pass recomputes the forward (cheaper than storing 2 x rows x U floats), back-propagates with the
//     transposed operand images, and accumulates ALL weight gradients in MFMA accumulators with the data row as the
//     contraction index; the two operands of those products need (unit,row) -> (row,unit) transposes, done 16x16 at
//     a time through a per-wave LDS scratch.  Per-wave accumulators are reduced in a fixed order: waves of a block
//     through LDS, blocks through a workspace and a second tiny kernel in fp64 (deterministic, no atomics).
// MFMA work per 16 rows at U=50: 80 (fwd) / 270 (bwd incl. recompute) instructions of 32 cycles (k-steps that would only
// multiply zero padding are skipped at compile time).  On gfx950 the fp32 MFMA runs at the fp32 vector rate and its time
// ADDS to the VALU time of the SIMD (tools/ubench/mfma_cover.hip), so the bound of both kernels is MFMA + VALU issue.
#include "vmp_common.h"

using namespace vmp;

namespace {

constexpr int FWD_THREADS = 256;       // forward: 2 blocks per CU
constexpr int BWD_THREADS = 512;       // backward: 1 block per CU (8 waves share one set of operand images)
constexpr int BWD_WAVES = BWD_THREADS / WAVE;
constexpr int TS = 20;                 // row stride (floats) of a 16x16 transpose scratch block
constexpr int TBLK = 16 * TS;

struct DecArgs {
    const float* x;        // (R, L)  rows = N*K*S
    const float* y;        // (N, Dy)
    const float* gA;       // (N, K)  backward: upstream gradient of A_nk = sum_s ll_row
    const float* gmean;    // (R, Dy) backward, gradient-input mode: upstream gradients of the two head outputs
    const float* gvar;     // (R, Dy)
    const float *W0, *b0, *W1, *b1, *W2, *b2, *Ws, *bs1, *bs2;
    float* ll;             // (R)      forward (nullable)
    float* mean;           // (R, Dy)  forward (nullable)
    float* var;            // (R, Dy)  forward (nullable)
    float* dx;             // (R, L)   backward
    float* part;           // (blocks, PW) backward: per-block partial parameter gradients
    unsigned R;            // rows
    unsigned K, S;
    int L, Dy, U;
};

struct DecGeo {
    int oW0, ob0, oW1, ob1, oW2, ob2, oWs, obs1, obs2, PW;
};
__host__ __device__ inline DecGeo dec_geo(int L, int U, int Dy) {
    DecGeo q;
    q.oW0 = 0;
    q.ob0 = q.oW0 + L * U;
    q.oW1 = q.ob0 + U;
    q.ob1 = q.oW1 + U * U;
    q.oW2 = q.ob1 + U;
    q.ob2 = q.oW2 + U * 2 * Dy;
    q.oWs = q.ob2 + 2 * Dy;
    q.obs1 = q.oWs + L * Dy;
    q.obs2 = q.obs1 + Dy;
    q.PW = q.obs2 + Dy;
    return q;
}

// LDS layout (floats / dwords).  fp32 operand image entry for MFMA operand `e`, lane l, sub-step v:  (e*64 + l)*4 + v.
// The two U x U products of a tile (layer 1 forward, dh0 backward) run on the XDL pipe as 3-term bf16 splits
// (vmp_common.h): their images hold, per (output tile t', k-block kb, term), one 16x32 bf16 A operand = 4 dwords per lane,
// k-slot j of lane group g  <->  hidden unit 16 (2 kb + (j >> 2)) + 4 g + (j & 3) - the two accumulator tiles 2kb, 2kb+1
// of the producing layer, so the layers still chain in registers.
template <int UT>
struct Img {
    static constexpr int KB = (UT + 1) / 2;             // k-blocks of 32 hidden units
    static constexpr int F0 = 0;                        // [t'][lane][kk<2 (pad 4)]      layer 0
    static constexpr int F1 = F0 + UT * 256;            // [t'][kb][term][lane][4]        layer 1 (bf16 x 3)
    static constexpr int F2 = F1 + UT * KB * 3 * 256;   // [t][lane][v]                   output layer
    static constexpr int F2S = F2 + UT * 256;           // [lane][kk<2 (pad 4)]           shortcut into the output layer
    static constexpr int BIAS0 = F2S + 256;             // 16*UT   b0 zero-padded
    static constexpr int BIAS1 = BIAS0 + 16 * UT;       // 16*UT   b1
    static constexpr int BIASO = BIAS1 + 16 * UT;       // 16      output bias in slot order
    static constexpr int SP2 = BIASO + 16;              // 8       log1p(exp(bs2_d))
    static constexpr int SG2 = SP2 + 8;                 // 8       sigmoid(bs2_d)
    static constexpr int FWD_END = SG2 + 8;
    static constexpr int B1 = FWD_END;                  // [t'][lane][v]      dh1 = W2 . dO
    static constexpr int B2 = B1 + UT * 256;            // [t'][kb][term][lane][4]   dh0 = W1 . dh1pre (bf16 x 3)
    static constexpr int B3 = B2 + UT * KB * 3 * 256;   // [t][lane][v]       dx  = W0 . dh0pre
    static constexpr int B3S = B3 + UT * 256;           // [lane][v]          dx += Ws . dO(mean slots)
    static constexpr int BWD_END = B3S + 256;
    // per-wave transpose scratch, two regions that are re-used through the tile:
    //   P: (h, m) bf16 terms of h0 [term][t][half][unit][8 rows] (UT*256 dwords), later dh0pre as fp32 blocks (UT*TBLK)
    //   Q: h1 and dO as fp32 blocks ((UT+1)*TBLK), later the (h, m) bf16 terms of dh1pre (UT*256 dwords)
    static constexpr int TW = UT * 256;
    static constexpr int PSZ = TW > UT * TBLK ? TW : UT * TBLK;
    static constexpr int QSZ = TW > (UT + 1) * TBLK ? TW : (UT + 1) * TBLK;
    static constexpr int SCR = BWD_END;
    static constexpr int BWD_TOTAL = SCR + BWD_WAVES * (PSZ + QSZ);
};

// output slot m (0..15) of the last layer: lane group g = m>>2 owns slots 4g..4g+3 = (mean d0, mean d1, var d0, var d1)
// with d0 = 2g, d1 = 2g+1 - so that a lane finds mean and variance of the same output dimension in its own registers.
__device__ __forceinline__ int slot_d(int m) { return 2 * (m >> 2) + (m & 1); }
__device__ __forceinline__ int slot_ty(int m) { return (m >> 1) & 1; }

__device__ __forceinline__ float rcp_f(float v) { return __builtin_amdgcn_rcpf(v); }
// log(1 + exp(v)), absolute error ~1e-7 (v_exp / v_log based); series for tiny exp(-|v|) keeps it relatively accurate
__device__ __forceinline__ float softplus_f(float v) {
    const float t = __expf(-fabsf(v));
    const float l = t < 1e-3f ? t * (1.0f - t * (0.5f - t * 0.33333334f)) : __logf(1.0f + t);
    return fmaxf(v, 0.f) + l;
}
__device__ __forceinline__ float sigmoid_f(float v) { return rcp_f(1.0f + __expf(-v)); }

// tanh as the clamped rational x P(x^2) / Q(x^2) of Eigen's generic_fast_tanh_float - the implementation behind
// tf.tanh in the reference's TensorFlow - max relative error 3.5e-7, |tanh| <= 1; two values per v_pk_* instruction.
__device__ __forceinline__ v2f tanh2(v2f x) {
    const float lim = 7.90531110763549805f;
    x[0] = __builtin_amdgcn_fmed3f(x[0], -lim, lim);
    x[1] = __builtin_amdgcn_fmed3f(x[1], -lim, lim);
    const v2f x2 = x * x;
    v2f p = x2 * -2.76076847742355e-16f + 2.00018790482477e-13f;
    p = p * x2 + -8.60467152213735e-11f;
    p = p * x2 + 5.12229709037114e-08f;
    p = p * x2 + 1.48572235717979e-05f;
    p = p * x2 + 6.37261928875436e-04f;
    p = p * x2 + 4.89352455891786e-03f;
    p = p * x;
    v2f q = x2 * 1.19825839466702e-06f + 1.18534705686654e-04f;
    q = q * x2 + 2.26843463243900e-03f;
    q = q * x2 + 4.89352518554385e-03f;
    v2f r;
    r[0] = rcp_f(q[0]);
    r[1] = rcp_f(q[1]);
    return p * r;
}
__device__ __forceinline__ f32x4 tanh4(f32x4 z) {
    const v2f a = tanh2(v2f{z[0], z[1]}), b = tanh2(v2f{z[2], z[3]});
    return f32x4{a[0], a[1], b[0], b[1]};
}

template <int UT, bool BWD, int THREADS>
__device__ void fill_images(float* __restrict__ sm, const DecArgs& a) {
    using I = Img<UT>;
    const int L = a.L, U = a.U, Dy = a.Dy;
    const int tid = threadIdx.x;
    // F0: A[i = unit 16t'+c][k = dim 4kk+g] = W0[dim][unit]
    for (int i = tid; i < UT * 256; i += THREADS) {
        const int v = i & 3, l = (i >> 2) & 63, tp = i >> 8, g = l >> 4, c = l & 15;
        const int dim = 4 * v + g, unit = 16 * tp + c;
        sm[I::F0 + i] = (v < 2 && dim < L && unit < U) ? a.W0[dim * U + unit] : 0.f;
    }
    // F1 (bf16 x 3): A[i = out 16t'+c][k-slot 8g+j of block kb = in 16(2kb + (j>>2)) + 4g + (j&3)] = W1[in][out]
    unsigned* __restrict__ smu = reinterpret_cast<unsigned*>(sm);
    constexpr int KB = I::KB;
    for (int i = tid; i < UT * KB * 256; i += THREADS) {
        const int dw = i & 3, l = (i >> 2) & 63, e = i >> 8, kb = e % KB, tp = e / KB, g = l >> 4, c = l & 15;
        v2f w;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = 2 * dw + h, t = 2 * kb + (j >> 2), in = 16 * t + 4 * g + (j & 3), out = 16 * tp + c;
            w[h] = (t < UT && in < U && out < U) ? a.W1[in * U + out] : 0.f;
        }
        unsigned t3[3];
        split_bf16<3>(w, t3);
#pragma unroll
        for (int term = 0; term < 3; ++term) smu[I::F1 + ((e * 3 + term) * 64 + l) * 4 + dw] = t3[term];
    }
    // F2: A[i = slot c][k = in 16t+4g+v] = W2[in][ty*Dy + d]
    for (int i = tid; i < UT * 256; i += THREADS) {
        const int v = i & 3, l = (i >> 2) & 63, t = i >> 8, g = l >> 4, c = l & 15;
        const int in = 16 * t + 4 * g + v, d = slot_d(c), ty = slot_ty(c);
        sm[I::F2 + i] = (in < U && d < Dy) ? a.W2[in * 2 * Dy + ty * Dy + d] : 0.f;
    }
    // F2S: A[i = slot c][k = dim 4kk+g] = Ws[dim][d] for the mean slots
    for (int i = tid; i < 256; i += THREADS) {
        const int v = i & 3, l = i >> 2, g = l >> 4, c = l & 15;
        const int dim = 4 * v + g, d = slot_d(c), ty = slot_ty(c);
        sm[I::F2S + i] = (v < 2 && dim < L && d < Dy && ty == 0) ? a.Ws[dim * Dy + d] : 0.f;
    }
    for (int i = tid; i < 16 * UT; i += THREADS) {
        sm[I::BIAS0 + i] = i < U ? a.b0[i] : 0.f;
        sm[I::BIAS1 + i] = i < U ? a.b1[i] : 0.f;
    }
    if (tid < 16) {
        const int d = slot_d(tid), ty = slot_ty(tid);
        sm[I::BIASO + tid] = d < Dy ? (ty == 0 ? a.b2[d] + a.bs1[d] : a.b2[Dy + d]) : 0.f;
    }
    if (tid < 8) {
        const float b = tid < Dy ? a.bs2[tid] : 0.f;
        sm[I::SP2 + tid] = log1pf(expf(b));            // the reference's naive form (vae.py:116)
        sm[I::SG2 + tid] = 1.0f / (1.0f + expf(-b));
    }
    if (BWD) {
        // B1: A[i = unit 16t'+c][k = slot 4g+v] = W2[unit][o(slot)]
        for (int i = tid; i < UT * 256; i += THREADS) {
            const int v = i & 3, l = (i >> 2) & 63, tp = i >> 8, g = l >> 4, c = l & 15;
            const int unit = 16 * tp + c, m = 4 * g + v, d = slot_d(m), ty = slot_ty(m);
            sm[I::B1 + i] = (unit < U && d < Dy) ? a.W2[unit * 2 * Dy + ty * Dy + d] : 0.f;
        }
        // B2 (bf16 x 3): A[i = in 16t'+c][k-slot 8g+j of block kb = out 16(2kb + (j>>2)) + 4g + (j&3)] = W1[in][out]
        for (int i = tid; i < UT * KB * 256; i += THREADS) {
            const int dw = i & 3, l = (i >> 2) & 63, e = i >> 8, kb = e % KB, tp = e / KB, g = l >> 4, c = l & 15;
            v2f w;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = 2 * dw + h, t = 2 * kb + (j >> 2), out = 16 * t + 4 * g + (j & 3), in = 16 * tp + c;
                w[h] = (t < UT && in < U && out < U) ? a.W1[in * U + out] : 0.f;
            }
            unsigned t3[3];
            split_bf16<3>(w, t3);
#pragma unroll
            for (int term = 0; term < 3; ++term) smu[I::B2 + ((e * 3 + term) * 64 + l) * 4 + dw] = t3[term];
        }
        // B3: A[i = dim c][k = unit 16t+4g+v] = W0[dim][unit]
        for (int i = tid; i < UT * 256; i += THREADS) {
            const int v = i & 3, l = (i >> 2) & 63, t = i >> 8, g = l >> 4, c = l & 15;
            const int unit = 16 * t + 4 * g + v;
            sm[I::B3 + i] = (c < L && unit < U) ? a.W0[c * U + unit] : 0.f;
        }
        // B3S: A[i = dim c][k = slot 4g+v] = Ws[dim][d] for the mean slots
        for (int i = tid; i < 256; i += THREADS) {
            const int v = i & 3, l = i >> 2, g = l >> 4, c = l & 15;
            const int m = 4 * g + v, d = slot_d(m), ty = slot_ty(m);
            sm[I::B3S + i] = (c < L && d < Dy && ty == 0) ? a.Ws[c * Dy + d] : 0.f;
        }
    }
}

__device__ __forceinline__ f32x4 mfma4(float av, float bv, f32x4 cv) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, cv, 0, 0, 0);
}
__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ u32x4 ldsu4(const float* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ f32x4 mfma_bf(u32x4 av, u32x4 bv, f32x4 cv) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), cv, 0, 0, 0);
}

// k-steps over hidden units are enumerated (t, v) <-> units 16t + 4g + v; in the last unit tile only v < VL reach a
// unit below U (VL = min(4, U - 16 (UT-1))), the others multiply zero padding and are skipped at compile time.
template <int UT, int VL>
__device__ __forceinline__ constexpr bool kstep_on(int t, int v) { return t < UT - 1 || v < VL; }

// row -> (cell, n) for the 16 rows of a tile: the tile's first row is divided on the scalar unit, the lane offset
// (< 16 + S) by an exact small-integer float division.
struct RowMap {
    unsigned cell, n;
};
__device__ __forceinline__ RowMap row_map(unsigned tile, int c, unsigned S, unsigned K, float invS, float invK,
                                          unsigned ncells) {
    const unsigned row0 = __builtin_amdgcn_readfirstlane(tile) * 16u;
    const unsigned cell0 = row0 / S, rem0 = row0 - cell0 * S;
    const unsigned n0 = cell0 / K, crem0 = cell0 - n0 * K;
    const unsigned q1 = (unsigned)(((float)(rem0 + (unsigned)c) + 0.5f) * invS);
    const unsigned q2 = (unsigned)(((float)(crem0 + q1) + 0.5f) * invK);
    RowMap m;
    m.cell = min(cell0 + q1, ncells - 1u);
    m.n = min(n0 + q2, (ncells - 1u) / K);
    return m;
}

// (unit 16t+4g+v, row c) activations of one layer -> the three bf16 terms of every value, packed per (v, v+1) pair:
// ts[term][2t + p] = values v = 2p, 2p+1 of tile t.  Registers 4kb .. 4kb+3 of a term ARE the B operand of k-block kb.
template <int UT>
__device__ __forceinline__ void split_tiles(const f32x4 (&h)[UT], unsigned (&ts)[3][4 * Img<UT>::KB]) {
#pragma unroll
    for (int i = 0; i < 4 * Img<UT>::KB; ++i) {
        if (i < 2 * UT) {
            unsigned t3[3];
            split_bf16<3>(v2f{h[i >> 1][2 * (i & 1)], h[i >> 1][2 * (i & 1) + 1]}, t3);
            ts[0][i] = t3[0]; ts[1][i] = t3[1]; ts[2][i] = t3[2];
        } else {
            ts[0][i] = 0u; ts[1][i] = 0u; ts[2][i] = 0u;
        }
    }
}

// out[t'] += W . act for a U x U weight image `img` (F1 or B2) on the XDL pipe: the six products of order <= 2 of the
// 3-term splits (hh, hm, hl, mh, mm, lh) - what is dropped is below 2^-24 of the product, as in the fp32 MFMA chain.
template <int UT>
__device__ __forceinline__ void gemm_uu_bf16(const float* __restrict__ img, int lane, const unsigned (&ts)[3][4 * Img<UT>::KB],
                                             f32x4 (&out)[UT]) {
    constexpr int KB = Img<UT>::KB;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int ta = 0; ta < 3; ++ta) {
            u32x4 w[UT];
#pragma unroll
            for (int tp = 0; tp < UT; ++tp) w[tp] = ldsu4(img + (((tp * KB + kb) * 3 + ta) * 64 + lane) * 4);
#pragma unroll
            for (int tb = 0; tb + ta < 3; ++tb) {
                const u32x4 bv = {ts[tb][4 * kb], ts[tb][4 * kb + 1], ts[tb][4 * kb + 2], ts[tb][4 * kb + 3]};
#pragma unroll
                for (int tp = 0; tp < UT; ++tp) out[tp] = mfma_bf(w[tp], bv, out[tp]);
            }
        }
    }
}

// forward of one 16-row tile; xb0/xb1 = x[row c][g], x[row c][4+g] (B operand of layer 0).  onev >= 0 on the lanes that
// own the free padding unit of the last tile: that unit's activation is forced to 1 (its weights are zero everywhere),
// see the bias-gradient note at dec_bwd_kernel.  h0s = bf16 terms of h0 (the backward pass transposes them for dW1).
template <int UT, int VL, bool ONES>
__device__ __forceinline__ void dec_forward_tile(const float* __restrict__ sm, int lane, float xb0, float xb1, int onev,
                                                 f32x4 (&h0)[UT], unsigned (&h0s)[3][4 * Img<UT>::KB], f32x4 (&h1)[UT], f32x4& O) {
    using I = Img<UT>;
    const int g = lane >> 4;
#pragma unroll
    for (int tp = 0; tp < UT; ++tp) {
        f32x4 acc = lds4(sm + I::BIAS0 + 16 * tp + 4 * g);
        const f32x4 w = lds4(sm + I::F0 + (tp * 64 + lane) * 4);
        acc = mfma4(w[0], xb0, acc);
        acc = mfma4(w[1], xb1, acc);
        h0[tp] = acc;
    }
#pragma unroll
    for (int tp = 0; tp < UT; ++tp) h0[tp] = tanh4(h0[tp]);
    if (ONES) {
#pragma unroll
        for (int v = 0; v < 4; ++v) h0[UT - 1][v] = onev == v ? 1.0f : h0[UT - 1][v];
    }
    split_tiles<UT>(h0, h0s);
#pragma unroll
    for (int tp = 0; tp < UT; ++tp) h1[tp] = lds4(sm + I::BIAS1 + 16 * tp + 4 * g);
    gemm_uu_bf16<UT>(sm + I::F1, lane, h0s, h1);
#pragma unroll
    for (int tp = 0; tp < UT; ++tp) h1[tp] = tanh4(h1[tp]);
    f32x4 o0 = lds4(sm + I::BIASO + 4 * g), o1 = {0.f, 0.f, 0.f, 0.f};
    {
        const f32x4 w = lds4(sm + I::F2S + lane * 4);
        o0 = mfma4(w[0], xb0, o0);
        o1 = mfma4(w[1], xb1, o1);
    }
#pragma unroll
    for (int t = 0; t < UT; ++t) {
        const f32x4 w = lds4(sm + I::F2 + (t * 64 + lane) * 4);
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (kstep_on<UT, VL>(t, v)) {
                if (v & 1) o1 = mfma4(w[v], h1[t][v], o1);
                else o0 = mfma4(w[v], h1[t][v], o0);
            }
    }
    O = o0 + o1;
}

template <int UT, int VL>
__global__ __launch_bounds__(FWD_THREADS, 2) void dec_fwd_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    using I = Img<UT>;
    fill_images<UT, false, FWD_THREADS>(sm, a);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const unsigned ntiles = (a.R + 15u) / 16u;
    const unsigned nwaves = gridDim.x * (FWD_THREADS / WAVE);
    const int L = a.L, Dy = a.Dy;
    const float invS = 1.0f / (float)a.S, invK = 1.0f / (float)a.K;
    const unsigned ncells = a.R / a.S;
    for (unsigned tile = blockIdx.x * (FWD_THREADS / WAVE) + wave; tile < ntiles; tile += nwaves) {
        const unsigned row = tile * 16u + c;
        const bool ok = row < a.R;
        const unsigned rr = ok ? row : a.R - 1u;
        const float* __restrict__ xr = a.x + (size_t)rr * L;
        const float xb0 = g < L ? xr[g] : 0.f;
        const float xb1 = 4 + g < L ? xr[4 + g] : 0.f;
        const RowMap rm = row_map(tile, c, a.S, a.K, invS, invK, ncells);
        float yv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) yv[j] = (2 * g + j < Dy && a.ll) ? a.y[(size_t)rm.n * Dy + 2 * g + j] : 0.f;
        f32x4 h0[UT], h1[UT], O;
        unsigned h0s[3][4 * I::KB];
        dec_forward_tile<UT, VL, false>(sm, lane, xb0, xb1, -1, h0, h0s, h1, O);
        float acc = 0.f;
        float mu[2], vr[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int d = 2 * g + j;
            mu[j] = O[j];
            vr[j] = softplus_f(O[2 + j]) + sm[I::SP2 + (d & 7)];
            const float df = yv[j] - mu[j];
            const float term = df * df * rcp_f(vr[j]) + __logf(vr[j] + 1e-8f);
            acc += d < Dy ? term : 0.f;
        }
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
        if (ok && g == 0 && a.ll) a.ll[row] = acc;
        if (ok && a.mean) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int d = 2 * g + j;
                if (d < Dy) {
                    a.mean[(size_t)row * Dy + d] = mu[j];
                    a.var[(size_t)row * Dy + d] = vr[j];
                }
            }
        }
    }
}

// Lanes of ONE wave exchange data through LDS: the LDS unit executes a wave's instructions in order, so no wait is
// needed - only the compiler must keep the program order of the accesses.
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (unit 4g+v, row c) accumulator block  ->  operand form: lane (g,c) gets block[unit c][rows 4g..4g+3]
__device__ __forceinline__ void tr_write(float* __restrict__ T, int g, int c, f32x4 z) {
#pragma unroll
    for (int v = 0; v < 4; ++v) T[(4 * g + v) * TS + c] = z[v];
}
__device__ __forceinline__ f32x4 tr_read(const float* __restrict__ T, int g, int c) { return lds4(T + c * TS + 4 * g); }

// The same transposition for the (h, m) bf16 terms of a layer's activations, for the weight-gradient product on the XDL
// pipe (data row = contraction index, 16 rows per tile).  Layout [term][tile][half = row>>3][unit][8 rows] of bf16:
// an operand read is one conflict-free ds_read_b128 per lane.  The 32 k-slots of the MFMA carry two terms of the 16
// rows: lanes g < 2 read rows 8g.. of the first term, lanes g >= 2 rows 8(g-2).. of the second, so
//   (h|m) x (h|h) = hh + mh   and   (h|m) x (m|m) = hm + mm
// - four products in two instructions, relative error <= 2^-17 per product (both factors carry 16+ bits).  That is
// below the fp32 rounding of a sum over >= 16 rows and far below that of the tens of thousands of rows a wave
// accumulates; the activations themselves (gemm_uu_bf16) keep all six products.
template <int UT>
__device__ __forceinline__ void trb_write(unsigned char* __restrict__ T, int g, int c, const unsigned (&ts)[3][4 * Img<UT>::KB]) {
    unsigned char* __restrict__ base = T + (c >> 3) * 256 + (4 * g) * 16 + 2 * (c & 7);
#pragma unroll
    for (int term = 0; term < 2; ++term)
#pragma unroll
        for (int t = 0; t < UT; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const unsigned pk = ts[term][2 * t + p];
                unsigned char* q = base + (term * UT + t) * 512 + (2 * p) * 16;
                *reinterpret_cast<unsigned short*>(q) = (unsigned short)pk;
                *reinterpret_cast<unsigned short*>(q + 16) = (unsigned short)(pk >> 16);
            }
}
__device__ __forceinline__ u32x4 trb_read(const unsigned char* __restrict__ p) { return *reinterpret_cast<const u32x4*>(p); }

// Bias gradients ride in the zero padding of the weight-gradient products: the x tile (A operand of dW0 and of the
// shortcut product) has rows L..15 free, so a row of ones at "dim 8" makes row 8 of those accumulators equal to
// sum_row dh0pre (= db0) and sum_row dO (= db2, dbs1); likewise a ones "unit U" in the transposed h0 block gives
// db1 as row U of dW1 when U is not a multiple of 16 (FS); otherwise db1 is summed on the VALU.
// GIN (gradient-input mode): the upstream gradients of (mean, var) are given per row instead of being derived from
// the log-likelihood - the same kernel then is the backward pass of a stand-alone Gaussian-head MLP (the encoder).
template <int UT, int VL, bool FS, bool GIN>
__global__ __launch_bounds__(BWD_THREADS, 2) void dec_bwd_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    using I = Img<UT>;
    fill_images<UT, true, BWD_THREADS>(sm, a);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    float* __restrict__ scrP = sm + I::SCR + wave * (I::PSZ + I::QSZ);
    float* __restrict__ scrQ = scrP + I::PSZ;
    unsigned char* __restrict__ scrPb = reinterpret_cast<unsigned char*>(scrP);
    unsigned char* __restrict__ scrQb = reinterpret_cast<unsigned char*>(scrQ);
    const int rd_hm = (((g >> 1) * UT) * 2 + (g & 1)) * 256 + c * 16;      // operand (h|m): + tile * 512
    const int rd_xx = (g & 1) * 256 + c * 16;                             // operand (h|h): + tile * 512; (m|m): + (UT + tile) * 512
    const unsigned ntiles = (a.R + 15u) / 16u;
    const unsigned nwaves = gridDim.x * BWD_WAVES;
    const int L = a.L, Dy = a.Dy, U = a.U;
    const float invS = 1.0f / (float)a.S, invK = 1.0f / (float)a.K;
    const unsigned ncells = a.R / a.S;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 aW1[UT][UT], aW2[UT], aWs, aW0[UT], ab1[FS ? 1 : UT];
    float abs2[2] = {0.f, 0.f};
    aWs = zero4;
    const int fsu = U & 15;                             // FS: the free unit slot of the last tile
    const int onev = (FS && g == (fsu >> 2)) ? (fsu & 3) : -1;
#pragma unroll
    for (int i = 0; i < (FS ? 1 : UT); ++i) ab1[i] = zero4;
#pragma unroll
    for (int i = 0; i < UT; ++i) {
        aW2[i] = zero4; aW0[i] = zero4;
#pragma unroll
        for (int j = 0; j < UT; ++j) aW1[i][j] = zero4;
    }

    for (unsigned tile = blockIdx.x * BWD_WAVES + wave; tile < ntiles; tile += nwaves) {
        const unsigned row = tile * 16u + c;
        const bool ok = row < a.R;
        const unsigned rr = ok ? row : a.R - 1u;
        const float* __restrict__ xr = a.x + (size_t)rr * L;
        const float xb0 = g < L ? xr[g] : 0.f;
        const float xb1 = 4 + g < L ? xr[4 + g] : 0.f;
        // x in operand form for the weight gradients: lane (g,c) <- x[row 4g+kk][dim c]
        f32x4 xT;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const unsigned r2 = tile * 16u + 4u * g + kk;
            xT[kk] = (r2 < a.R && c < L) ? a.x[(size_t)r2 * L + c] : (c == 8 ? 1.0f : 0.f);
        }
        float ga = 0.f, yv[2] = {0.f, 0.f}, gin_m[2] = {0.f, 0.f}, gin_v[2] = {0.f, 0.f};
        if (GIN) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool dv = ok && 2 * g + j < Dy;
                const float m_ = a.gmean[(size_t)rr * Dy + (dv ? 2 * g + j : 0)], v_ = a.gvar[(size_t)rr * Dy + (dv ? 2 * g + j : 0)];
                gin_m[j] = dv ? m_ : 0.f;
                gin_v[j] = dv ? v_ : 0.f;
            }
        } else {
            const RowMap rm = row_map(tile, c, a.S, a.K, invS, invK, ncells);
            ga = ok ? a.gA[rm.cell] : 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) yv[j] = 2 * g + j < Dy ? a.y[(size_t)rm.n * Dy + 2 * g + j] : 0.f;
        }

        f32x4 h0[UT], h1[UT], O;
        {
            unsigned h0s[3][4 * I::KB];
            dec_forward_tile<UT, VL, FS>(sm, lane, xb0, xb1, onev, h0, h0s, h1, O);
            trb_write<UT>(scrPb, g, c, h0s);
        }
#pragma unroll
        for (int t = 0; t < UT; ++t) tr_write(scrQ + t * TBLK, g, c, h1[t]);

        // ---- reconstruction term: gradients w.r.t. the output slots
        f32x4 dO;
        float llacc = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int d = 2 * g + j;
            const bool dv = d < Dy;
            const float raw2 = O[2 + j];
            const float vr = softplus_f(raw2) + sm[I::SP2 + (d & 7)];
            const float df = yv[j] - O[j], iv = rcp_f(vr);
            const float gm = GIN ? gin_m[j] : (dv ? ga * (-2.f * df * iv) : 0.f);
            const float gv = GIN ? gin_v[j] : (dv ? ga * (rcp_f(vr + 1e-8f) - df * df * iv * iv) : 0.f);
            const float gr = gv * sigmoid_f(raw2);
            dO[j] = gm;
            dO[2 + j] = gr;
            abs2[j] += gv;
            if (!GIN && a.ll) llacc += dv ? df * df * iv + __logf(vr + 1e-8f) : 0.f;
        }
        if (!GIN && a.ll) {                                      // value and gradient in one pass (wave-uniform branch)
            llacc += __shfl_xor(llacc, 16);
            llacc += __shfl_xor(llacc, 32);
            if (ok && g == 0) a.ll[row] = llacc;
        }
        tr_write(scrQ + UT * TBLK, g, c, dO);
        // ---- dh1 = W2 . dO
        f32x4 dh1[UT];
#pragma unroll
        for (int tp = 0; tp < UT; ++tp) {
            const f32x4 w = lds4(sm + I::B1 + (tp * 64 + lane) * 4);
            f32x4 acc = zero4;
#pragma unroll
            for (int v = 0; v < 4; ++v) acc = mfma4(w[v], dO[v], acc);
            dh1[tp] = acc;
        }
        // ---- dW2 (and shortcut W): [h1 ; x]^T . dO
        wave_lds_order();
        {
            f32x4 h1T[UT];
#pragma unroll
            for (int t = 0; t < UT; ++t) h1T[t] = tr_read(scrQ + t * TBLK, g, c);
            const f32x4 dOT = tr_read(scrQ + UT * TBLK, g, c);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int t = 0; t < UT; ++t) aW2[t] = mfma4(h1T[t][kk], dOT[kk], aW2[t]);
                aWs = mfma4(xT[kk], dOT[kk], aWs);
            }
        }
        // ---- through tanh of layer 1
#pragma unroll
        for (int tp = 0; tp < UT; ++tp) {
#pragma unroll
            for (int v = 0; v < 4; ++v) dh1[tp][v] *= 1.0f - h1[tp][v] * h1[tp][v];
            if (!FS) ab1[tp] += dh1[tp];
        }
        wave_lds_order();
        // ---- dh0 = W1 . dh1pre (XDL pipe), and the terms of dh1pre transposed for dW1
        f32x4 dh0[UT];
#pragma unroll
        for (int tp = 0; tp < UT; ++tp) dh0[tp] = zero4;
        {
            unsigned d1s[3][4 * I::KB];
            split_tiles<UT>(dh1, d1s);
            trb_write<UT>(scrQb, g, c, d1s);
            gemm_uu_bf16<UT>(sm + I::B2, lane, d1s, dh0);
        }
        // ---- dW1 = h0^T . dh1pre
        wave_lds_order();
        {
            u32x4 dTh[UT], dTm[UT];
#pragma unroll
            for (int t = 0; t < UT; ++t) {
                dTh[t] = trb_read(scrQb + rd_xx + t * 512);
                dTm[t] = trb_read(scrQb + rd_xx + (UT + t) * 512);
            }
#pragma unroll
            for (int ti = 0; ti < UT; ++ti) {
                const u32x4 h0T = trb_read(scrPb + rd_hm + ti * 512);
#pragma unroll
                for (int tj = 0; tj < UT; ++tj) aW1[ti][tj] = mfma_bf(h0T, dTh[tj], aW1[ti][tj]);
#pragma unroll
                for (int tj = 0; tj < UT; ++tj) aW1[ti][tj] = mfma_bf(h0T, dTm[tj], aW1[ti][tj]);
            }
        }
        // ---- through tanh of layer 0
#pragma unroll
        for (int tp = 0; tp < UT; ++tp) {
#pragma unroll
            for (int v = 0; v < 4; ++v) dh0[tp][v] *= 1.0f - h0[tp][v] * h0[tp][v];
        }
        wave_lds_order();
#pragma unroll
        for (int t = 0; t < UT; ++t) tr_write(scrP + t * TBLK, g, c, dh0[t]);
        // ---- dx = W0 . dh0pre + Ws . dO(mean)
        {
            f32x4 d0 = zero4, d1 = zero4;
            const f32x4 ws = lds4(sm + I::B3S + lane * 4);
            d0 = mfma4(ws[0], dO[0], d0);
            d1 = mfma4(ws[1], dO[1], d1);
            d0 = mfma4(ws[2], dO[2], d0);
            d1 = mfma4(ws[3], dO[3], d1);
#pragma unroll
            for (int t = 0; t < UT; ++t) {
                const f32x4 w = lds4(sm + I::B3 + (t * 64 + lane) * 4);
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (kstep_on<UT, VL>(t, v)) {
                        if (v & 1) d1 = mfma4(w[v], dh0[t][v], d1);
                        else d0 = mfma4(w[v], dh0[t][v], d0);
                    }
            }
            const f32x4 dxv = d0 + d1;                   // [dim 4g+v][row c]
            if (ok && a.dx) {
                if (L == 8) {
                    if (g < 2) *reinterpret_cast<f32x4*>(a.dx + (size_t)row * 8 + 4 * g) = dxv;
                } else {
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (4 * g + v < L) a.dx[(size_t)row * L + 4 * g + v] = dxv[v];
                }
            }
        }
        // ---- dW0 = x^T . dh0pre
        wave_lds_order();
        {
            f32x4 dT[UT];
#pragma unroll
            for (int t = 0; t < UT; ++t) dT[t] = tr_read(scrP + t * TBLK, g, c);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int tj = 0; tj < UT; ++tj) aW0[tj] = mfma4(xT[kk], dT[tj][kk], aW0[tj]);
        }
        wave_lds_order();
    }

    // ---- reduce the per-wave accumulators through LDS.  Every parameter index is owned by exactly one (lane,
    // register) of a wave, so each wave drops its accumulators into a slab of its own with plain stores (a turn-taking
    // read-modify-write over 8 waves cost ~50 us per launch - most of the kernel at minibatch sizes); then all threads
    // sum the slabs in wave order (deterministic).  Two rounds of BWD_WAVES/2 slabs stay inside the LDS allocation.
    const DecGeo q = dec_geo(L, U, Dy);
    __syncthreads();                                    // everybody is done with the operand images
    constexpr int HALF = BWD_WAVES / 2;
    float* __restrict__ accum = sm;
    for (int round = 0; round < 2; ++round) {
        if (wave / HALF == round) {
            float* __restrict__ slab = sm + (1 + wave % HALF) * q.PW;
#pragma unroll
            for (int ti = 0; ti < UT; ++ti)
#pragma unroll
                for (int tj = 0; tj < UT; ++tj)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int in = 16 * ti + 4 * g + v, out = 16 * tj + c;
                        if (in < U && out < U) slab[q.oW1 + in * U + out] = aW1[ti][tj][v];
                    }
#pragma unroll
            for (int t = 0; t < UT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int unit = 16 * t + 4 * g + v, d = slot_d(c), ty = slot_ty(c);
                    if (unit < U && d < Dy) slab[q.oW2 + unit * 2 * Dy + ty * Dy + d] = aW2[t][v];
                    const int dim = 4 * g + v, u2 = 16 * t + c;
                    if (dim < L && u2 < U) slab[q.oW0 + dim * U + u2] = aW0[t][v];
                }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int dim = 4 * g + v, d = slot_d(c), ty = slot_ty(c);
                if (dim < L && d < Dy && ty == 0) slab[q.oWs + dim * Dy + d] = aWs[v];
            }
            // biases out of the ones rows: dim 8 <-> lane group g == 2, register 0
            if (g == 2) {
#pragma unroll
                for (int t = 0; t < UT; ++t)
                    if (16 * t + c < U) slab[q.ob0 + 16 * t + c] = aW0[t][0];
                const int d = slot_d(c), ty = slot_ty(c);
                if (d < Dy) {
                    slab[q.ob2 + ty * Dy + d] = aWs[0];
                    if (ty == 0) slab[q.obs1 + d] = aWs[0];
                }
            }
            if (FS) {
                if (g == (fsu >> 2)) {
#pragma unroll
                    for (int tj = 0; tj < UT; ++tj) {
                        const f32x4 z = aW1[UT - 1][tj];
                        const int v = fsu & 3;
                        const float val = v == 0 ? z[0] : v == 1 ? z[1] : v == 2 ? z[2] : z[3];
                        if (16 * tj + c < U) slab[q.ob1 + 16 * tj + c] = val;
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < UT; ++t)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float s1 = row16_sum(ab1[t][v]);
                        const int unit = 16 * t + 4 * g + v;
                        if (c == 0 && unit < U) slab[q.ob1 + unit] = s1;
                    }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float s2_ = row16_sum(abs2[j]);
                const int d = 2 * g + j;
                if (c == 0 && d < Dy) slab[q.obs2 + d] = s2_;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < q.PW; i += BWD_THREADS) {
            float t = round ? accum[i] : 0.f;
#pragma unroll
            for (int w = 0; w < HALF; ++w) t += sm[(1 + w) * q.PW + i];
            accum[i] = t;
        }
        __syncthreads();
    }
    // sigmoid(bs2) factor of d/d bs2 log1p(exp(bs2)) is applied by the reduce kernel
    for (int i = threadIdx.x; i < q.PW; i += BWD_THREADS) a.part[(size_t)blockIdx.x * q.PW + i] = accum[i];
}

struct DecRedArgs {
    const float* part;
    const float* bs2;
    float* out;
    int blocks, PW, obs2, Dy;
};
__global__ __launch_bounds__(256) void dec_reduce_kernel(DecRedArgs r) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= r.PW) return;
    double s = 0.0;
    for (int b = 0; b < r.blocks; ++b) s += (double)r.part[(size_t)b * r.PW + i];
    if (i >= r.obs2) s *= 1.0 / (1.0 + exp(-(double)r.bs2[i - r.obs2]));
    r.out[i] = (float)s;
}

int dec_blocks(long long rows, int waves_per_block, int max_blocks) {
    const long long tiles = (rows + 15) / 16;
    long long b = (tiles + waves_per_block - 1) / waves_per_block;
    if (b > max_blocks) b = max_blocks;
    if (b < 1) b = 1;
    return (int)b;
}
int dec_fwd_blocks(long long rows) { return dec_blocks(rows, FWD_THREADS / WAVE, 512); }   // 2 blocks on each of 256 CUs
int dec_bwd_blocks(long long rows) { return dec_blocks(rows, BWD_WAVES, 256); }            // 1 block per CU

int dec_check(const char* what, long long N, int K, int S, int L, int Dy, int U) {
    if (N < 0 || K < 1 || S < 1 || L < 1 || L > 8 || Dy < 1 || Dy > 8 || U < 1 || U > 64) {
        set_error("%s: unsupported sizes N=%lld K=%d S=%d L=%d Dy=%d U=%d (L, Dy <= 8, U <= 64)", what, N, K, S, L, Dy, U);
        return VMP_E_DIM;
    }
    if ((double)N * K * S >= 2147483648.0) {
        set_error("%s: N*K*S = %.0f rows exceed 2^31 - split the batch", what, (double)N * K * S);
        return VMP_E_DIM;
    }
    return 0;
}

// dispatch on (UT, VL) = (unit tiles, live k-steps of the last tile)
#define DEC_DISPATCH(U, CALL)                                                                  \
    do {                                                                                       \
        const int ut_ = ((U) + 15) / 16, vl_ = ((U) - 16 * (ut_ - 1)) >= 4 ? 4 : ((U) - 16 * (ut_ - 1)); \
        switch (ut_ * 4 + vl_ - 1) {                                                           \
            case 4: CALL(1, 1); break;  case 5: CALL(1, 2); break;  case 6: CALL(1, 3); break;  case 7: CALL(1, 4); break;   \
            case 8: CALL(2, 1); break;  case 9: CALL(2, 2); break;  case 10: CALL(2, 3); break; case 11: CALL(2, 4); break;  \
            case 12: CALL(3, 1); break; case 13: CALL(3, 2); break; case 14: CALL(3, 3); break; case 15: CALL(3, 4); break;  \
            case 16: CALL(4, 1); break; case 17: CALL(4, 2); break; case 18: CALL(4, 3); break; default: CALL(4, 4); break;  \
        }                                                                                      \
    } while (0)

template <bool GIN>
int dec_bwd_launch(const DecArgs& a, int blocks, hipStream_t s) {
    const int U = a.U;
    const int red_floats = (1 + BWD_WAVES / 2) * dec_geo(a.L, a.U, a.Dy).PW;     // epilogue: accumulator + 4 slabs
#define DEC_BWD(UTV, VLV)                                                                                             \
    do {                                                                                                              \
        const int lds = (Img<UTV>::BWD_TOTAL > red_floats ? Img<UTV>::BWD_TOTAL : red_floats) * (int)sizeof(float);   \
        if (VLV == 4 && (U & 15) == 0) {                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_bwd_kernel<UTV, 4, false, GIN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            hipLaunchKernelGGL((dec_bwd_kernel<UTV, 4, false, GIN>), dim3(blocks), dim3(BWD_THREADS), lds, s, a);    \
        } else {                                                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_bwd_kernel<UTV, VLV, true, GIN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            hipLaunchKernelGGL((dec_bwd_kernel<UTV, VLV, true, GIN>), dim3(blocks), dim3(BWD_THREADS), lds, s, a);   \
        }                                                                                                             \
    } while (0)
    DEC_DISPATCH(U, DEC_BWD);
#undef DEC_BWD
    return check_launch(GIN ? "vmp_mlp_gauss_bwd" : "vmp_decoder_loglike_bwd");
}

}  // namespace

extern "C" {

int vmp_decoder_param_words(int L, int U, int Dy) { return dec_geo(L, U, Dy).PW; }

size_t vmp_decoder_workspace_bytes(int64_t N, int K, int S, int L, int U, int Dy) {
    return (size_t)dec_bwd_blocks((long long)N * K * S) * (size_t)dec_geo(L, U, Dy).PW * sizeof(float);
}

int vmp_decoder_loglike_fwd(const float* x, const float* y, const float* W0, const float* b0, const float* W1,
                            const float* b1, const float* W2, const float* b2, const float* Ws, const float* bs1,
                            const float* bs2, int64_t N, int K, int S, int L, int Dy, int U, float* ll, float* mean,
                            float* var, void* stream) {
    if (int e = dec_check("vmp_decoder_loglike_fwd", N, K, S, L, Dy, U)) return e;
    if (!x || (!y && ll) || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !Ws || !bs1 || !bs2 || (!ll && !mean) || (!mean != !var)) {
        set_error("vmp_decoder_loglike_fwd: NULL argument");
        return VMP_E_BADARG;
    }
    if (N == 0) return 0;
    DecArgs a{};
    a.x = x; a.y = y; a.W0 = W0; a.b0 = b0; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.Ws = Ws; a.bs1 = bs1; a.bs2 = bs2;
    a.ll = ll; a.mean = mean; a.var = var;
    a.R = (unsigned)(N * K * S); a.K = (unsigned)K; a.S = (unsigned)S; a.L = L; a.Dy = Dy; a.U = U;
    const int blocks = dec_fwd_blocks((long long)a.R);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define DEC_FWD(UTV, VLV)                                                                                             \
    do {                                                                                                              \
        const int lds = Img<UTV>::FWD_END * (int)sizeof(float);                                                       \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fwd_kernel<UTV, VLV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        hipLaunchKernelGGL((dec_fwd_kernel<UTV, VLV>), dim3(blocks), dim3(FWD_THREADS), lds, s, a);                   \
    } while (0)
    DEC_DISPATCH(U, DEC_FWD);
#undef DEC_FWD
    return check_launch("vmp_decoder_loglike_fwd");
}

int vmp_decoder_loglike_bwd(const float* x, const float* y, const float* gA, const float* W0, const float* b0,
                            const float* W1, const float* b1, const float* W2, const float* b2, const float* Ws,
                            const float* bs1, const float* bs2, int64_t N, int K, int S, int L, int Dy, int U, float* dx,
                            float* dparams, float* ll, void* ws, size_t ws_bytes, void* stream) {
    if (int e = dec_check("vmp_decoder_loglike_bwd", N, K, S, L, Dy, U)) return e;
    if (!x || !y || !gA || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !Ws || !bs1 || !bs2 || !dx || !dparams || !ws) {
        set_error("vmp_decoder_loglike_bwd: NULL argument");
        return VMP_E_BADARG;
    }
    const DecGeo q = dec_geo(L, U, Dy);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N == 0) {
        (void)hipMemsetAsync(dparams, 0, (size_t)q.PW * sizeof(float), s);
        return check_launch("vmp_decoder_loglike_bwd");
    }
    if (ws_bytes < vmp_decoder_workspace_bytes(N, K, S, L, U, Dy)) {
        set_error("vmp_decoder_loglike_bwd: workspace too small (%zu < %zu bytes)", ws_bytes,
                  vmp_decoder_workspace_bytes(N, K, S, L, U, Dy));
        return VMP_E_WS;
    }
    DecArgs a{};
    a.x = x; a.y = y; a.gA = gA; a.W0 = W0; a.b0 = b0; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.Ws = Ws; a.bs1 = bs1; a.bs2 = bs2;
    a.dx = dx; a.part = static_cast<float*>(ws); a.ll = ll;
    a.R = (unsigned)(N * K * S); a.K = (unsigned)K; a.S = (unsigned)S; a.L = L; a.Dy = Dy; a.U = U;
    const int blocks = dec_bwd_blocks((long long)a.R);
    if (int e = dec_bwd_launch<false>(a, blocks, s)) return e;
    DecRedArgs r{a.part, bs2, dparams, blocks, q.PW, q.obs2, Dy};
    hipLaunchKernelGGL(dec_reduce_kernel, dim3((q.PW + 255) / 256), dim3(256), 0, s, r);
    return check_launch("vmp_decoder_loglike_bwd(reduce)");
}

int vmp_mlp_gauss_bwd(const float* x, const float* gmean, const float* gvar, const float* W0, const float* b0,
                      const float* W1, const float* b1, const float* W2, const float* b2, const float* Ws, const float* bs1,
                      const float* bs2, int64_t R, int L, int Dy, int U, float* dx, float* dparams, void* ws,
                      size_t ws_bytes, void* stream) {
    if (int e = dec_check("vmp_mlp_gauss_bwd", R, 1, 1, L, Dy, U)) return e;
    if (!x || !gmean || !gvar || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !Ws || !bs1 || !bs2 || !dparams || !ws) {
        set_error("vmp_mlp_gauss_bwd: NULL argument");
        return VMP_E_BADARG;
    }
    const DecGeo q = dec_geo(L, U, Dy);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (R == 0) {
        (void)hipMemsetAsync(dparams, 0, (size_t)q.PW * sizeof(float), s);
        return check_launch("vmp_mlp_gauss_bwd");
    }
    if (ws_bytes < vmp_decoder_workspace_bytes(R, 1, 1, L, U, Dy)) {
        set_error("vmp_mlp_gauss_bwd: workspace too small");
        return VMP_E_WS;
    }
    DecArgs a{};
    a.x = x; a.gmean = gmean; a.gvar = gvar; a.W0 = W0; a.b0 = b0; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.Ws = Ws;
    a.bs1 = bs1; a.bs2 = bs2; a.dx = dx; a.part = static_cast<float*>(ws);
    a.R = (unsigned)R; a.K = 1; a.S = 1; a.L = L; a.Dy = Dy; a.U = U;
    const int blocks = dec_bwd_blocks((long long)a.R);
    if (int e = dec_bwd_launch<true>(a, blocks, s)) return e;
    DecRedArgs r{a.part, bs2, dparams, blocks, q.PW, q.obs2, Dy};
    hipLaunchKernelGGL(dec_reduce_kernel, dim3((q.PW + 255) / 256), dim3(256), 0, s, r);
    return check_launch("vmp_mlp_gauss_bwd(reduce)");
}

}  // extern "C"
