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
//   * every product runs on the XDL matrix pipe as v_mfma_f32_16x16x32_bf16 over 3-term bf16 splits of BOTH operands
//     (v = h + m + l, vmp_common.h): the six products of order <= 2 reproduce the fp32 product to 2^-24 - the accuracy of
//     the fp32 chain they replace (round 1 used v_mfma_f32_16x16x4_f32, which on gfx950 runs at the fp32 VECTOR rate in
//     the VALU issue slots: 32 cycles per K = 4 against 16 cycles per K = 32 here).  M = output unit, N = data row,
//     K = input unit.  The accumulator layout of layer l (lane (g,c), register v  <->  unit 16t+4g+v, row c) IS the
//     B-operand layout of layer l+1 when its 32 k-slots are enumerated as (tile pair, v) - the contraction order is
//     free - so the layers chain without any transpose or LDS round trip; the weights are split and pre-permuted into
//     that order once per block ("operand images" in LDS, conflict-free ds_read_b128).
//   * contractions shorter than 32 (K = L <= 8 input dims, K = 16 output slots) put SEVERAL TERMS of the same values
//     into the 32 k-slots, so one instruction yields two to four of the six products:
//         K = 16:  (h|m) x (h|h),  (h|m) x (m|m),  (l|h) x (h|l)                           3 instructions
//         K =  8:  (h|m|l|h) x (h|h|h|m),  (m|h|0|0) x (m|l|0|0)                            2 instructions
//   * the backward pass recomputes the forward (cheaper than storing 2 x rows x U floats), back-propagates with the
//     transposed operand images, and accumulates ALL weight gradients in MFMA accumulators with the data row as the
//     contraction index (16 rows: the (h|m) x (h|h), (h|m) x (m|m) pair gives hh + mh + hm + mm, relative error
//     <= 2^-17 per product - below the fp32 rounding of the row sums they feed).  The operands of those products need
//     (unit,row) -> (row,unit) transposes of the bf16 terms, done through a per-wave LDS scratch.  Per-wave
//     accumulators are reduced in a fixed order: waves of a block through LDS, blocks through a workspace and a second
//     tiny kernel in fp64 (deterministic, no atomics).
// MFMA work per 16 rows at U=50: 70 (fwd) / 195 (bwd incl. recompute) instructions of 16 cycles.  Measured on MI355X
// (tools/ubench/xdl_overlap.hip): beside enough VALU work a bf16 MFMA costs ~10-12 issue cycles of its SIMD, i.e. the
// XDL pipe overlaps the VALU only partially; the kernels are bound by VALU + MFMA issue.
#include "vmp_common.h"
#include "vmp_tail.h"
#include "vmp_step_parts.h"
#include "vmp_prep_parts.h"

using namespace vmp;

namespace {

#ifndef VMP_DEC_FWD_TERMS_FOLLOW_BT
#define VMP_DEC_FWD_TERMS_FOLLOW_BT 1      // the backward kernel's forward recompute uses its BT-term operands (0: always 3 terms, A/B builds)
#endif
constexpr int FWD_THREADS = 256;       // forward: 2+ blocks per CU
constexpr int BWD_THREADS = 512;       // backward: 1 block per CU (8 waves share one set of operand images)
constexpr int BWD_WAVES = BWD_THREADS / WAVE;

struct DecArgs {
    const float* x;        // (R, L)  rows = N*K*S
    const float* y;        // (N, Dy)
    const float* gA;       // (N, K)  backward: upstream gradient of A_nk = sum_s ll_row
    float logw;            // != 0: gA holds LOG weights and the upstream gradient is logw * exp(gA) (vmp_decoder_loglike_bwd_logw)
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
    int split;             // backward: % of a SIMD pair's tiles that go to the older wave (see dec_bwd_kernel)
    int red_one;           // backward epilogue: all 8 waves' slabs fit the LDS at once (one round instead of two)
    float vscale;          // the second head output is vscale * var (forward) and its upstream gradient is scaled alike
                           // (backward, gradient-input mode): 1 = 'standard' head, -1/2 = the encoder's 'natparam' head
#ifdef VMP_DEBUG_TS
    long long* dbg_t;      // exploration builds only (tools/build_variant.sh ts -DVMP_DEBUG_TS): stage stamps of block 0, thread 0
#endif
};
#ifdef VMP_DEBUG_TS
#define DEC_TS(i) do { if (a.dbg_t && blockIdx.x == 0 && threadIdx.x == 0) { a.dbg_t[i] = clock64(); a.dbg_t[32 + (i)] = wall_clock64(); } } while (0)
// stage stamps of ONE streaming tile (the 9th of block 0, wave 0): tools/dec_tile_ts.py
#define DEC_TT(i) do { if (a.dbg_t && blockIdx.x == 0 && threadIdx.x == 0 && tile == t0 + 8) a.dbg_t[64 + (i)] = clock64(); } while (0)
#else
#define DEC_TS(i) do { } while (0)
#define DEC_TT(i) do { } while (0)
#endif

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

// LDS layout (dwords).  A bf16 operand image entry = 4 dwords (8 bf16 k-slots) per lane: (e*64 + lane)*4.
//   K = 32 hidden units per k-block kb: k-slot j of lane group g  <->  unit 16 (2 kb + (j >> 2)) + 4 g + (j & 3), i.e. the
//     accumulator tiles 2kb, 2kb+1 of the producing layer; one image per (output tile, kb, term).
//   K = 16 output slots: k-slots 0-3 of lane group g <-> slots 4g..4g+3 of one term, k-slots 4-7 the same slots of another
//     term; two images per output tile: form 0 = (h|m), form 1 = (l|h).
//   K = 8 input dims (lane group g holds dims g and 4+g): k-slot 2q+d <-> dim g+4d in term combination q; form a =
//     (h,m,l,h) 4 dwords per lane, form b = (m,h) 2 dwords per lane.
template <int UT>
struct Img {
    static constexpr int KB = (UT + 1) / 2;             // k-blocks of 32 hidden units
    static constexpr int F0A = 0;                       // [t'][lane][4]                  layer 0, form a
    static constexpr int F0B = F0A + UT * 256;          // [t'][lane][2]                  layer 0, form b
    static constexpr int F1 = F0B + UT * 128;           // [t'][kb][term][lane][4]        layer 1
    static constexpr int F2 = F1 + UT * KB * 3 * 256;   // [kb][term][lane][4]            output layer
    static constexpr int F2SA = F2 + KB * 3 * 256;      // [lane][4]                      shortcut into the output layer, form a
    static constexpr int F2SB = F2SA + 256;             // [lane][2]                      form b
    static constexpr int BIAS0 = F2SB + 128;            // 16*UT   b0 zero-padded (fp32)
    static constexpr int BIAS1 = BIAS0 + 16 * UT;       // 16*UT   b1
    static constexpr int BIASO = BIAS1 + 16 * UT;       // 16      output bias in slot order
    static constexpr int SP2 = BIASO + 16;              // 8       log1p(exp(bs2_d))
    static constexpr int SG2 = SP2 + 8;                 // 8       sigmoid(bs2_d)
    static constexpr int FWD_END = SG2 + 8;
    static constexpr int B1 = FWD_END;                  // [t'][form][lane][4]            dh1 = W2 . dO          (K = 16 slots)
    static constexpr int B2 = B1 + UT * 2 * 256;        // [t'][kb][term][lane][4]        dh0 = W1 . dh1pre
    static constexpr int B3 = B2 + UT * KB * 3 * 256;   // [kb][term][lane][4]            dx  = W0 . dh0pre
    static constexpr int B3S = B3 + KB * 3 * 256;       // [form][lane][4]                dx += Ws . dO(mean slots) (K = 16 slots)
    static constexpr int BWD_END = B3S + 2 * 256;
    // per-wave transpose scratch for the weight-gradient products, bf16 (h, m) terms, [term][tile][half][unit][8 rows]:
    //   X: the x tile (dims 0..7; row "dim 8" = ones, see dec_bwd_kernel)                       256 dwords
    //   P: h0, later dh0pre                                                                      UT * 256
    //   Q: h1 and dO, later dh1pre                                                               (UT + 1) * 256
    static constexpr int XSZ = 256, PSZ = UT * 256, QSZ = (UT + 1) * 256;
    static constexpr int SCR = BWD_END;
    static constexpr int BWD_TOTAL = SCR + BWD_WAVES * (XSZ + PSZ + QSZ);
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

// tanh(x) = 1 - 2 / (1 + 2^(2 log2(e) x)): v_exp_f32 and v_rcp_f32 (1 ulp each) plus three plain operations per value -
// about half the issue time of the clamped rational P(x^2)/Q(x^2) of Eigen's generic_fast_tanh_float behind tf.tanh
// (13 packed operations + 2 v_med3 + 2 v_rcp per PAIR; measured cost table in DESIGN.md section 6).  Absolute error
// <= 1.5e-7 over the whole range (the rational: 3.5e-7 relative), saturates to +-1 exactly through exp overflow /
// underflow, |tanh| <= 1.  VMP_TANH_RATIONAL=1 builds the rational form.
#ifndef VMP_TANH_RATIONAL
#define VMP_TANH_RATIONAL 0
#endif
#if VMP_TANH_RATIONAL
#define TANH_PRESCALE 1.0f
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
#else
// tanh of a pre-activation that arrives ALREADY scaled by TANH_PRESCALE = 2 log2(e): fill_images folds the factor into the two
// hidden layers' forward weight images and biases (one multiply per weight per block instead of one per unit per row - 32 of a
// tile's 605 VALU instructions), so e^(2z) is a bare v_exp_f32 of the accumulator.
#ifndef VMP_DEC_TANH_FOLD
#define VMP_DEC_TANH_FOLD 1
#endif
#if VMP_DEC_TANH_FOLD
#define TANH_PRESCALE 2.8853900817779268f
#define TANH_INSCALE 1.0f
#else
#define TANH_PRESCALE 1.0f
#define TANH_INSCALE 2.8853900817779268f
#endif
__device__ __forceinline__ float tanh1(float x) {
    const float e = __builtin_amdgcn_exp2f(x * TANH_INSCALE);              // e^(2z), x = z * 2 log2(e) when folded
    return fmaf(-2.0f, rcp_f(e + 1.0f), 1.0f);
}
__device__ __forceinline__ f32x4 tanh4(f32x4 z) { return f32x4{tanh1(z[0]), tanh1(z[1]), tanh1(z[2]), tanh1(z[3])}; }
#endif

// hidden unit behind k-slot j of lane group g in k-block kb
__device__ __forceinline__ int kslot_unit(int g, int j, int kb) { return 16 * (2 * kb + (j >> 2)) + 4 * g + (j & 3); }

// Two phases: every global load of the thread is ISSUED first (fully unrolled, values held in registers), then the values
// are split and stored.  Interleaved (load - split - store per element, as this function first was) a thread paid one
// exposed L2 / HBM round trip per loop iteration and phase - about 10 of them in the backward kernel - which is most of
// the run time of a one-tile-per-wave launch (the reference's minibatches of 64-100 rows); at C3 sizes the fill is
// amortised over thousands of tiles either way.
template <int UT, bool BWD, int THREADS>
__device__ void fill_images(float* __restrict__ sm, const DecArgs& a) {
    using I = Img<UT>;
    constexpr int KB = I::KB;
    unsigned* __restrict__ smu = reinterpret_cast<unsigned*>(sm);
    const int L = a.L, U = a.U, Dy = a.Dy;
    const int tid = threadIdx.x;
    constexpr int N1 = (BWD ? 2 : 1) * UT * KB * 256, IT1 = N1 / THREADS;      // F1 (+ B2)
    constexpr int N2 = (BWD ? 2 : 1) * KB * 256, IT2 = N2 / THREADS;           // F2 (+ B3)
    constexpr int N0 = (UT + 1) * 64, IT0 = (N0 + THREADS - 1) / THREADS;      // F0 + F2S, B1 + B3S
    static_assert(N1 % THREADS == 0 && N2 % THREADS == 0, "whole iterations");
    // ---- phase 1: loads
    // F0: A[i = unit 16t'+c][dims g, 4+g] = W0[dim][unit]; F2S: A[i = slot c][dims g, 4+g] = Ws[dim][d] for the mean slots
    v2f w0[IT0];
#pragma unroll
    for (int it = 0; it < IT0; ++it) {
        const int i = tid + it * THREADS;
        const int l = i & 63, tp = i >> 6, g = l >> 4, c = l & 15;
        v2f w{0.f, 0.f};
        if (i < N0) {
            if (tp < UT) {
                const int unit = 16 * tp + c;
                w[0] = (g < L && unit < U) ? a.W0[g * U + unit] : 0.f;
                w[1] = (4 + g < L && unit < U) ? a.W0[(4 + g) * U + unit] : 0.f;
            } else {
                const int d = slot_d(c), ty = slot_ty(c);
                w[0] = (g < L && d < Dy && ty == 0) ? a.Ws[g * Dy + d] : 0.f;
                w[1] = (4 + g < L && d < Dy && ty == 0) ? a.Ws[(4 + g) * Dy + d] : 0.f;
            }
        }
        w0[it] = w;
    }
    // F1: A[i = out 16t'+c][k-slot -> in] = W1[in][out];  B2: A[i = in 16t'+c][k-slot -> out] = W1[in][out]
    v2f w1[IT1];
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int i = tid + it * THREADS;
        const int dw = i & 3, l = (i >> 2) & 63, e0 = i >> 8, bw = e0 >= UT * KB, e = bw ? e0 - UT * KB : e0;
        const int kb = e % KB, tp = e / KB, g = l >> 4, c = l & 15;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = 2 * dw + h, ku = kslot_unit(g, j, kb), mu = 16 * tp + c;
            const int in = bw ? mu : ku, out = bw ? ku : mu;
            w1[it][h] = (ku < 16 * UT && in < U && out < U) ? a.W1[in * U + out] : 0.f;
        }
    }
    // F2: A[i = slot c][k-slot -> unit] = W2[unit][ty*Dy + d];  B3: A[i = dim c][k-slot -> unit] = W0[dim][unit]
    v2f w2[IT2];
#pragma unroll
    for (int it = 0; it < IT2; ++it) {
        const int i = tid + it * THREADS;
        const int dw = i & 3, l = (i >> 2) & 63, e0 = i >> 8, bw = e0 >= KB, kb = bw ? e0 - KB : e0, g = l >> 4, c = l & 15;
        const int d = slot_d(c), ty = slot_ty(c);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int unit = kslot_unit(g, 2 * dw + h, kb);
            if (bw) w2[it][h] = (unit < U && c < L) ? a.W0[c * U + unit] : 0.f;
            else w2[it][h] = (unit < U && d < Dy) ? a.W2[unit * 2 * Dy + ty * Dy + d] : 0.f;
        }
    }
    // B1: A[i = unit 16t'+c][slots 4g..4g+3] = W2[unit][o(slot)];  B3S: A[i = dim c][slots 4g..4g+3] = Ws[dim][d], mean slots
    v2f w3[BWD ? IT0 : 1][2];
    if (BWD) {
#pragma unroll
        for (int it = 0; it < IT0; ++it) {
            const int i = tid + it * THREADS;
            const int l = i & 63, tp = i >> 6, g = l >> 4, c = l & 15;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                v2f w{0.f, 0.f};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int m = 4 * g + 2 * p + h, d = slot_d(m), ty = slot_ty(m);
                    if (i < N0) {
                        if (tp < UT) w[h] = (16 * tp + c < U && d < Dy) ? a.W2[(16 * tp + c) * 2 * Dy + ty * Dy + d] : 0.f;
                        else w[h] = (c < L && d < Dy && ty == 0) ? a.Ws[c * Dy + d] : 0.f;
                    }
                }
                w3[it][p] = w;
            }
        }
    }
    float bias0[(16 * UT + THREADS - 1) / THREADS], bias1[(16 * UT + THREADS - 1) / THREADS];
#pragma unroll
    for (int it = 0; it < (16 * UT + THREADS - 1) / THREADS; ++it) {
        const int i = tid + it * THREADS;
        bias0[it] = i < U ? a.b0[i] : 0.f;
        bias1[it] = i < U ? a.b1[i] : 0.f;
    }
    float bo = 0.f, bsp = 0.f;
    if (tid < 16) {
        const int d = slot_d(tid), ty = slot_ty(tid);
        bo = d < Dy ? (ty == 0 ? a.b2[d] + a.bs1[d] : a.b2[Dy + d]) : 0.f;
    }
    if (tid < 8) bsp = tid < Dy ? a.bs2[tid] : 0.f;
    // ---- phase 2: split and store (the hidden layers' FORWARD images and biases carry the tanh prescale - applied here, not at the
    //      loads: a multiply behind each predicated load would bring back one exposed round trip per load)
#pragma unroll
    for (int it = 0; it < IT0; ++it) {
        const int i = tid + it * THREADS;
        if (i >= N0) break;
        const int l = i & 63, tp = i >> 6;
        unsigned t3[3];
        split_bf16<3>(w0[it] * (tp < UT ? TANH_PRESCALE : 1.0f), t3);
        unsigned* __restrict__ pa = smu + (tp < UT ? I::F0A + (tp * 64 + l) * 4 : I::F2SA + l * 4);
        unsigned* __restrict__ pb = smu + (tp < UT ? I::F0B + (tp * 64 + l) * 2 : I::F2SB + l * 2);
        pa[0] = t3[0]; pa[1] = t3[1]; pa[2] = t3[2]; pa[3] = t3[0];
        pb[0] = t3[1]; pb[1] = t3[0];
    }
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
        const int i = tid + it * THREADS;
        const int dw = i & 3, l = (i >> 2) & 63, e0 = i >> 8, bw = e0 >= UT * KB, e = bw ? e0 - UT * KB : e0;
        unsigned t3[3];
        split_bf16<3>(w1[it] * (bw ? 1.0f : TANH_PRESCALE), t3);
#pragma unroll
        for (int term = 0; term < 3; ++term) smu[(bw ? I::B2 : I::F1) + ((e * 3 + term) * 64 + l) * 4 + dw] = t3[term];
    }
#pragma unroll
    for (int it = 0; it < IT2; ++it) {
        const int i = tid + it * THREADS;
        const int dw = i & 3, l = (i >> 2) & 63, e0 = i >> 8, bw = e0 >= KB, kb = bw ? e0 - KB : e0;
        unsigned t3[3];
        split_bf16<3>(w2[it], t3);
#pragma unroll
        for (int term = 0; term < 3; ++term) smu[(bw ? I::B3 : I::F2) + ((kb * 3 + term) * 64 + l) * 4 + dw] = t3[term];
    }
#pragma unroll
    for (int it = 0; it < (16 * UT + THREADS - 1) / THREADS; ++it) {
        const int i = tid + it * THREADS;
        if (i < 16 * UT) {
            sm[I::BIAS0 + i] = TANH_PRESCALE * bias0[it];
            sm[I::BIAS1 + i] = TANH_PRESCALE * bias1[it];
        }
    }
    if (tid < 16) sm[I::BIASO + tid] = bo;
    if (tid < 8) {
        sm[I::SP2 + tid] = log1p_f(expf(bsp));          // the reference's naive form (vae.py:116)
        sm[I::SG2 + tid] = 1.0f / (1.0f + expf(-bsp));
    }
    if (BWD) {
#pragma unroll
        for (int it = 0; it < IT0; ++it) {
            const int i = tid + it * THREADS;
            if (i >= N0) break;
            const int l = i & 63, tp = i >> 6;
            unsigned th[2], tm[2], tl[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                unsigned t3[3];
                split_bf16<3>(w3[it][p], t3);
                th[p] = t3[0]; tm[p] = t3[1]; tl[p] = t3[2];
            }
            unsigned* __restrict__ q = smu + (tp < UT ? I::B1 + (tp * 2 * 64 + l) * 4 : I::B3S + l * 4);
            q[0] = th[0]; q[1] = th[1]; q[2] = tm[0]; q[3] = tm[1];                 // form 0 = (h|m)
            q[256] = tl[0]; q[257] = tl[1]; q[258] = th[0]; q[259] = th[1];         // form 1 = (l|h)
        }
    }
}

__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ u32x4 ldsu4(const float* p) { return *reinterpret_cast<const u32x4*>(p); }
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 ldsu2(const float* p) { return *reinterpret_cast<const u32x2*>(p); }
__device__ __forceinline__ f32x4 mfma_bf(u32x4 av, u32x4 bv, f32x4 cv) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), cv, 0, 0, 0);
}

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
template <int UT, int TERMS = 3>
__device__ __forceinline__ void split_tiles(const f32x4 (&h)[UT], unsigned (&ts)[3][4 * Img<UT>::KB]) {
#pragma unroll
    for (int i = 0; i < 4 * Img<UT>::KB; ++i) {
        if (i < 2 * UT) {
            unsigned t3[3];
            split_bf16<TERMS>(v2f{h[i >> 1][2 * (i & 1)], h[i >> 1][2 * (i & 1) + 1]}, t3);
            ts[0][i] = t3[0]; ts[1][i] = t3[1]; ts[2][i] = t3[2];
        } else {
            ts[0][i] = 0u; ts[1][i] = 0u; ts[2][i] = 0u;
        }
    }
}

// out[t'] += W . act over the hidden units (K = 16 UT, k-blocks of 32), W = weight image `img` with OT output tiles:
// the six products of order <= 2 of the 3-term splits (hh, hm, hl, mh, mm, lh); TERMS = 2: the three products hh, hm, mh of
// 2-term splits (relative error <= 2^-16 per product - the backward data path of large batches, see dec_bwd_kernel).
template <int UT, int OT, int TERMS = 3>
__device__ __forceinline__ void gemm_units(const float* __restrict__ img, int lane, const unsigned (&ts)[3][4 * Img<UT>::KB],
                                           f32x4 (&out)[OT]) {
    constexpr int KB = Img<UT>::KB;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int ta = 0; ta < TERMS; ++ta) {
            u32x4 w[OT];
#pragma unroll
            for (int tp = 0; tp < OT; ++tp) w[tp] = ldsu4(img + (((tp * KB + kb) * 3 + ta) * 64 + lane) * 4);
#pragma unroll
            for (int tb = 0; tb + ta < TERMS; ++tb) {
                const u32x4 bv = {ts[tb][4 * kb], ts[tb][4 * kb + 1], ts[tb][4 * kb + 2], ts[tb][4 * kb + 3]};
#pragma unroll
                for (int tp = 0; tp < OT; ++tp) out[tp] = mfma_bf(w[tp], bv, out[tp]);
            }
        }
    }
}
// the same with ONE output tile: one accumulator chain per k-block (a single chain would be 6 KB dependent instructions)
template <int UT, int TERMS = 3>
__device__ __forceinline__ f32x4 gemm_units_1(const float* __restrict__ img, int lane, const unsigned (&ts)[3][4 * Img<UT>::KB],
                                              f32x4 init) {
    constexpr int KB = Img<UT>::KB;
    f32x4 acc[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) acc[kb] = kb == 0 ? init : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ta = 0; ta < TERMS; ++ta) {
        u32x4 w[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) w[kb] = ldsu4(img + ((kb * 3 + ta) * 64 + lane) * 4);
#pragma unroll
        for (int tb = 0; tb + ta < TERMS; ++tb)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const u32x4 bv = {ts[tb][4 * kb], ts[tb][4 * kb + 1], ts[tb][4 * kb + 2], ts[tb][4 * kb + 3]};
                acc[kb] = mfma_bf(w[kb], bv, acc[kb]);
            }
    }
    f32x4 r = acc[0];
#pragma unroll
    for (int kb = 1; kb < KB; ++kb) r += acc[kb];
    return r;
}

// K = 8 input dims: xs = the three terms of the lane's (x[row c][g], x[row c][4+g]) pair
struct XOps {
    u32x4 b1, b2;
};
__device__ __forceinline__ XOps x_operands(const unsigned (&xs)[3]) {
    return XOps{u32x4{xs[0], xs[0], xs[0], xs[1]}, u32x4{xs[1], xs[2], 0u, 0u}};
}
__device__ __forceinline__ f32x4 gemm_dims(const float* __restrict__ img_a, const float* __restrict__ img_b, int lane, const XOps& xo, f32x4 acc) {
    const u32x4 a1 = ldsu4(img_a + lane * 4);
    const u32x2 a2 = ldsu2(img_b + lane * 2);
    acc = mfma_bf(a1, xo.b1, acc);
    return mfma_bf(u32x4{a2[0], a2[1], 0u, 0u}, xo.b2, acc);
}

// K = 16 output slots: ds = the three terms of the lane's slot pairs (4g, 4g+1), (4g+2, 4g+3); img = [form][lane][4]
struct SOps {
    u32x4 hh, mm, hl;
};
__device__ __forceinline__ SOps slot_operands(const unsigned (&ds)[3][2]) {
    return SOps{u32x4{ds[0][0], ds[0][1], ds[0][0], ds[0][1]}, u32x4{ds[1][0], ds[1][1], ds[1][0], ds[1][1]},
                u32x4{ds[0][0], ds[0][1], ds[2][0], ds[2][1]}};
}
template <int TERMS = 3>
__device__ __forceinline__ f32x4 gemm_slots(const float* __restrict__ img, int lane, const SOps& so, f32x4 acc) {
    const u32x4 a0 = ldsu4(img + lane * 4);
    acc = mfma_bf(a0, so.hh, acc);
    acc = mfma_bf(a0, so.mm, acc);                           // hh + mh + hm + mm
    if constexpr (TERMS == 3) acc = mfma_bf(ldsu4(img + 256 + lane * 4), so.hl, acc);   // + lh + hl
    return acc;
}

// forward of one 16-row tile.  onev >= 0 on the lanes that own the free padding unit of the last tile: that unit's
// activation is forced to 1 (its weights are zero everywhere), see the bias-gradient note at dec_bwd_kernel.
// h0s / h1s = bf16 terms of the activations (the backward pass transposes them for the weight gradients).
// TERMS = 2 (round 5: the backward kernel's recompute at >= 2^19 rows, VMP_DEC_FWD_TERMS_FOLLOW_BT): 2-term operands for the two
// hidden layers and the output layer as well - 2^-17 per product on activations whose every consumer is a sum over >= 2^19 rows or
// the per-row dx.  profiles/r05_decoder_tile_breakdown.txt: the 3-term splits of h0 / h1 were 128 of the tile's 808 VALU
// instructions and 60 of its 160 MFMAs carried their third-term products; same box 7.14 -> 6.29 ms per 4.2e7 rows.  Against the
// fp64 oracle at N = 65 536 (1e7 rows): every gradient <= 3.5e-6 relative (3-term recompute: <= 6e-7; bar 1e-4; the reference's own
// fp32 arithmetic: 3e-7 .. 1e-5), ELBO 2.7e-8.
template <int UT, bool ONES, int TERMS = 3>
__device__ __forceinline__ void dec_forward_tile(const float* __restrict__ sm, int lane, const XOps& xo, int onev,
                                                 f32x4 (&h0)[UT], unsigned (&h0s)[3][4 * Img<UT>::KB], f32x4 (&h1)[UT],
                                                 unsigned (&h1s)[3][4 * Img<UT>::KB], f32x4& O) {
    using I = Img<UT>;
    const int g = lane >> 4;
    {
        u32x4 a1[UT];
        u32x2 a2[UT];
#pragma unroll
        for (int tp = 0; tp < UT; ++tp) {
            h0[tp] = lds4(sm + I::BIAS0 + 16 * tp + 4 * g);
            a1[tp] = ldsu4(sm + I::F0A + (tp * 64 + lane) * 4);
            a2[tp] = ldsu2(sm + I::F0B + (tp * 64 + lane) * 2);
        }
#pragma unroll
        for (int tp = 0; tp < UT; ++tp) h0[tp] = mfma_bf(a1[tp], xo.b1, h0[tp]);
#pragma unroll
        for (int tp = 0; tp < UT; ++tp) h0[tp] = mfma_bf(u32x4{a2[tp][0], a2[tp][1], 0u, 0u}, xo.b2, h0[tp]);
    }
#pragma unroll
    for (int tp = 0; tp < UT; ++tp) h0[tp] = tanh4(h0[tp]);
    if (ONES) {
#pragma unroll
        for (int v = 0; v < 4; ++v) h0[UT - 1][v] = onev == v ? 1.0f : h0[UT - 1][v];
    }
    split_tiles<UT, TERMS>(h0, h0s);
#pragma unroll
    for (int tp = 0; tp < UT; ++tp) h1[tp] = lds4(sm + I::BIAS1 + 16 * tp + 4 * g);
    gemm_units<UT, UT, TERMS>(sm + I::F1, lane, h0s, h1);
#pragma unroll
    for (int tp = 0; tp < UT; ++tp) h1[tp] = tanh4(h1[tp]);
    split_tiles<UT, TERMS>(h1, h1s);
    const f32x4 os = gemm_dims(sm + I::F2SA, sm + I::F2SB, lane, xo, f32x4{0.f, 0.f, 0.f, 0.f});
    O = gemm_units_1<UT, TERMS>(sm + I::F2, lane, h1s, lds4(sm + I::BIASO + 4 * g)) + os;
}

// block `bid` of `nb` blocks of FWD_THREADS threads
template <int UT>
__device__ __forceinline__ void dec_fwd_body(const DecArgs& a, float* __restrict__ sm, const unsigned bid, const unsigned nb) {
    using I = Img<UT>;
    fill_images<UT, false, FWD_THREADS>(sm, a);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const unsigned ntiles = (a.R + 15u) / 16u;
    const unsigned nwaves = nb * (FWD_THREADS / WAVE);
    const int L = a.L, Dy = a.Dy;
    const float invS = 1.0f / (float)a.S, invK = 1.0f / (float)a.K;
    const unsigned ncells = a.R / a.S;
    for (unsigned tile = bid * (FWD_THREADS / WAVE) + wave; tile < ntiles; tile += nwaves) {
        asm volatile("" ::: "memory");                  // keep the (loop-invariant) operand-image reads inside the loop: hoisted, they spill
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
        unsigned xs[3];
        split_bf16<3>(v2f{xb0, xb1}, xs);
        const XOps xo = x_operands(xs);
        f32x4 h0[UT], h1[UT], O;
        unsigned h0s[3][4 * I::KB], h1s[3][4 * I::KB];
        dec_forward_tile<UT, false>(sm, lane, xo, -1, h0, h0s, h1, h1s, O);
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
        acc = rows4_sum(acc);                                    // same order of additions as the backward kernel's value
        if (ok && g == 0 && a.ll) a.ll[row] = acc;
        if (ok && a.mean) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int d = 2 * g + j;
                if (d < Dy) {
                    a.mean[(size_t)row * Dy + d] = mu[j];
                    a.var[(size_t)row * Dy + d] = vr[j] * a.vscale;
                }
            }
        }
    }
}

template <int UT>
__global__ __launch_bounds__(FWD_THREADS, 2) void dec_fwd_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    dec_fwd_body<UT>(a, sm, blockIdx.x, gridDim.x);
}

// The first launch of the minibatch training step (round 6): everything that depends on the parameters and the minibatch only, as
// ONE grid - blocks [0, nb) the encoder's forward pass (dec_fwd_body), blocks [nb, nb + K) the recognition unpacking and
// [nb + K, nb + 2K) the theta packing (wave 0 of the block: phi_prep_body / theta_pack_body, vmp_prep_parts.h; prep_both_kernel was
// a launch of its own), and - for a step replayed from a HIP graph - block nb, thread 64, moves the step's three scalars
// [Philox key | CVI step size | Adam step size] from row `*counter` of a table the host filled in advance to the 16 bytes the
// later launches read, and advances the counter (it was an eager launch per replay: vmp_svae_step_scalars).
struct StepTable {
    const unsigned long long* table;    // (rows, 2) 64-bit words: [key | (rho, lr_t) as two floats]
    unsigned long long* counter;        // next row
    unsigned long long* dst16;
    unsigned rows;
};
template <int UT, int LP>
__global__ __launch_bounds__(FWD_THREADS, 2) void enc_prep_kernel(DecArgs a, PhiArgs p, ThetaArgs t, StepTable st, unsigned nb) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    if (blockIdx.x < nb) {
        dec_fwd_body<UT>(a, sm, blockIdx.x, nb);
        return;
    }
    const unsigned b = blockIdx.x - nb;
    if (b == 0 && threadIdx.x == WAVE && st.table) {
        unsigned long long i = *st.counter;
        *st.counter = i + 1;
        if (i >= st.rows) i = st.rows - 1;
        st.dst16[0] = st.table[2 * i];
        st.dst16[1] = st.table[2 * i + 1];
    }
    if (threadIdx.x >= PREP_THREADS) return;
    if (b < (unsigned)p.K) phi_prep_body<LP, false, false, true>(p, (int)b);
    else theta_pack_body<LP, true>(t, (int)(b - p.K));
}

// Lanes of ONE wave exchange data through LDS: the LDS unit executes a wave's instructions in order, so no wait is
// needed - only the compiler must keep the program order of the accesses.
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Transposition of the (h, m) bf16 terms of a (unit 4g+v, row c) accumulator block set for the weight-gradient products
// (data row = contraction index, 16 rows per tile).  Layout [term][tile][half = row>>3][unit][8 rows] of bf16: an operand
// read is one conflict-free ds_read_b128 per lane.  The 32 k-slots of the MFMA carry two terms of the 16 rows: lanes
// g < 2 read rows 8g.. of the first term, lanes g >= 2 rows 8(g-2).. of the second, so
//   (h|m) x (h|h) = hh + mh   and   (h|m) x (m|m) = hm + mm
// - four products in two instructions, relative error <= 2^-17 per product (both factors carry 16+ bits).
// NT tiles; ts[term][2t + p] = packed values (v = 2p, 2p+1) of tile t (split_tiles).
// Swizzle: in the second half (rows 8..15) a unit's 16-byte slot is rotated by two within its group of four, so that
// the two half-rows of a write instruction (lanes c < 8 and c >= 8, 256 bytes apart = the same banks) land on different
// banks; a read still covers each half's 256 bytes exactly once.
__device__ __forceinline__ int trb_slot(int unit, int half) { return (unit & ~3) | ((unit + 2 * half) & 3); }
// One ds_write_b32 per packed register instead of two ds_write_b16 (a 16-bit LDS write costs ~8 LDS cycles, a 32-bit one
// ~4; with the b16 form the LDS pipe was 83 % busy, profiles/r02_decoder_pmc_summary.txt): neighbouring lanes (rows c, c^1)
// first swap halves, so that the even lane holds (unit 2p: rows c, c+1) and the odd lane (unit 2p+1: rows c-1, c).
template <int NT, int NTS>
__device__ __forceinline__ void trb_write(unsigned char* __restrict__ T, int g, int c, const unsigned (&ts)[3][NTS]) {
    const int half = c >> 3, odd = c & 1;
    unsigned char* __restrict__ base = T + half * 256 + 2 * ((c & 7) - odd);
    const unsigned sel = odd ? 0x03020706u : 0x05040100u;
    int off[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) off[p] = trb_slot(4 * g + 2 * p + odd, half) * 16;
#pragma unroll
    for (int term = 0; term < 2; ++term)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const unsigned pk = ts[term][2 * t + p];
                const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk, 0xB1, 0xF, 0xF, true);   // quad_perm:[1,0,3,2]
                *reinterpret_cast<unsigned*>(base + (term * NT + t) * 512 + off[p]) = __builtin_amdgcn_perm(nb, pk, sel);
            }
}
__device__ __forceinline__ u32x4 trb_read(const unsigned char* __restrict__ p) { return *reinterpret_cast<const u32x4*>(p); }

// Round 5: the same transposition by gfx950's LDS TRANSPOSE READ (ds_read_b64_tr_b16) instead of VALU lane exchanges.
// Layout per (term, tile): the 16 rows x 16 units of bf16 ROW-major (512 bytes, as before): lane (g, c) writes its four units
// 4g..4g+3 of row c as ONE 8-byte chunk (no v_mov_dpp / v_perm: 136 VALU instructions per tile less); an operand read is two
// transpose reads - within a 16-lane group lane i passes the address of 8-byte chunk i of a 4-row x 16-unit block (chunk 4j + q =
// row j, units 4q..4q+3) and receives column i: unit i of the 4 rows (tools/ubench/ds_read_tr16.hip) - i.e. the 8 k-slots
// "rows 8 (g & 1) .. + 7 of unit c" that one ds_read_b128 of the old layout delivered.  Bank spread: the chunks of a row are
// XOR-swizzled by the row's bits 2..3 (lanes c, c + 4, .. of a write instruction hit different banks), and the block of rows
// 8..15 is rotated by 128 bytes (the two 16-lane groups of a read cycle, rows 0..3 and 8..11, use different bank halves).
#ifndef VMP_DEC_TR16
#define VMP_DEC_TR16 1            // 0: A/B builds with the VALU transposition (trb_write / trb_read)
#endif
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int tr_row_off(int r) { return r < 8 ? 32 * r : 256 + ((32 * (r - 8) + 128) & 255); }
// byte offset of lane (g, c)'s chunk inside a (term, tile) image
__device__ __forceinline__ int tr_write_off(int g, int c) { return tr_row_off(c) + 8 * (g ^ ((c >> 2) & 3)); }
// byte offsets of the two transpose reads of lane (g, c): rows r0 .. r0 + 3 and r0 + 4 .. r0 + 7, r0 = 8 (g & 1)
__device__ __forceinline__ void tr_read_offs(int g, int c, int& o1, int& o2) {
    const int r0 = 8 * (g & 1), j = c >> 2, q = c & 3;
    o1 = tr_row_off(r0 + j) + 8 * (q ^ ((r0 >> 2) & 3));
    o2 = tr_row_off(r0 + 4 + j) + 8 * (q ^ (((r0 + 4) >> 2) & 3));
}
template <int NT, int NTS>
__device__ __forceinline__ void trt_write(unsigned char* __restrict__ T, int woff, const unsigned (&ts)[3][NTS]) {
#pragma unroll
    for (int term = 0; term < 2; ++term)
#pragma unroll
        for (int t = 0; t < NT; ++t)
            *reinterpret_cast<u32x2*>(T + (term * NT + t) * 512 + woff) = u32x2{ts[term][2 * t], ts[term][2 * t + 1]};
}
__device__ __forceinline__ u32x4 trt_read(const unsigned char* __restrict__ img, int o1, int o2) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + o1));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + o2));
    const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
    return u32x4{ua[0], ua[1], ub[0], ub[1]};
}

// Bias gradients ride in the zero padding of the weight-gradient products: the x tile (A operand of dW0 and of the
// shortcut product) has rows L..15 free, so a row of ones at "dim 8" makes row 8 of those accumulators equal to
// sum_row dh0pre (= db0) and sum_row dO (= db2, dbs1); likewise a ones "unit U" in h0 gives db1 as row U of dW1 when U
// is not a multiple of 16 (FS); otherwise db1 is summed on the VALU.
// GIN (gradient-input mode): the upstream gradients of (mean, var) are given per row instead of being derived from
// the log-likelihood - the same kernel then is the backward pass of a stand-alone Gaussian-head MLP (the encoder).
// BT = bf16 terms per operand on the backward DATA path (dh1 = W2 dO, dh0 = W1 dh1pre, dx = W0 dh0pre + Ws dO): 3 = the six
// products of fp32 accuracy; 2 = three products, relative error <= 2^-16 per product.  The consumers of that path are sums
// over rows (the weight gradients, which take (h, m) terms of it anyway) and dL/dx, which the E-step backward and the
// encoder sum over samples, components and rows - for batches of >= 2^19 sample rows the 2-term form's rounding is below
// the fp32 rounding of those sums (gated by the 21-gradient bars of tests/test_fullsize_gpu.py at N = 65 536 and 1e5),
// and it saves 34 of 195 MFMAs and 64 of 840 VALU instructions per 16-row tile.  Small batches keep BT = 3.
template <int UT, bool FS, bool GIN, int BT>
__global__ __launch_bounds__(BWD_THREADS, 2) void dec_bwd_kernel(DecArgs a) {
    DEC_TS(0);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    using I = Img<UT>;
    constexpr int KB = I::KB;
    fill_images<UT, true, BWD_THREADS>(sm, a);
    DEC_TS(1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    float* __restrict__ scr = sm + I::SCR + wave * (I::XSZ + I::PSZ + I::QSZ);
    unsigned char* __restrict__ scrXb = reinterpret_cast<unsigned char*>(scr);
    unsigned char* __restrict__ scrPb = reinterpret_cast<unsigned char*>(scr + I::XSZ);
    unsigned char* __restrict__ scrQb = reinterpret_cast<unsigned char*>(scr + I::XSZ + I::PSZ);
    unsigned char* __restrict__ scrOb = scrQb + 2 * UT * 512;                  // dO terms behind the h1 terms
    {   // x tile: dims 8..15 are constants - "dim 8" = ones (h term 1.0, m term 0), the rest zero
        unsigned* __restrict__ xz = reinterpret_cast<unsigned*>(scr);
#pragma unroll
        for (int i = 0; i < 4; ++i) xz[lane + 64 * i] = 0u;
        wave_lds_order();
        if (lane < 8) xz[(lane >> 2) * 64 + trb_slot(8, lane >> 2) * 4 + (lane & 3)] = 0x3F803F80u;
    }
    __syncthreads();
    const int rd_c = trb_slot(c, g & 1) * 16;
    const int rd_hm = (((g >> 1) * UT) * 2 + (g & 1)) * 256 + rd_c;        // operand (h|m) of a UT-tile set: + tile * 512
    const int rd_hm1 = ((g >> 1) * 2 + (g & 1)) * 256 + rd_c;              // operand (h|m) of a 1-tile set
    const int rd_xx = (g & 1) * 256 + rd_c;                               // operand (h|h): + tile * 512; (m|m): + (NT + tile) * 512
    // transpose-read form (P / Q scratch; the x tile keeps the VALU form): chunk offset of this lane's writes, offsets of its two
    // reads, and the image of its lane group's term for (h|m) operands
    const int tr_w = tr_write_off(g, c);
    int tr_o1, tr_o2;
    tr_read_offs(g, c, tr_o1, tr_o2);
    const int tr_hm = (g >> 1) * UT * 512;                                // (h|m) of a UT-tile set: + tile * 512
    const int tr_hm1 = (g >> 1) * 512;                                    // (h|m) of a 1-tile set
    const unsigned ntiles = (a.R + 15u) / 16u;
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

    // Tile ranges.  A block owns a contiguous run of tiles, each SIMD pair of waves (w, w + 4) a quarter of it.  The sequencer
    // serves the OLDER wave of a pair first, so with equal shares the younger one would run the last third of the kernel
    // alone, with nothing to hide its latencies behind (measured on the T1 pass kernel, tools/pass_ts.py): the older wave
    // gets a.split % of the pair's tiles.  Static ranges: the summation order, hence every bit of the result, is fixed.
    const unsigned tpb = (ntiles + gridDim.x - 1) / gridDim.x;
    const unsigned b0 = min(blockIdx.x * tpb, ntiles), b1 = min(b0 + tpb, ntiles);
    const unsigned tpp = (tpb + 3) / 4;
    const unsigned p0 = min(b0 + (wave & 3) * tpp, b1), p1 = min(p0 + tpp, b1);
    const unsigned pm = min(p0 + (tpp >= 8u ? (tpp * (unsigned)a.split + 50u) / 100u : (tpp + 1u) / 2u), p1);   // few tiles: even shares
    const unsigned t0 = wave < 4 ? p0 : pm, t1 = wave < 4 ? pm : p1;
    DEC_TS(2);
    // per-tile inputs of a lane: x[row c][g], x[row c][4+g] and either (gA, y) or the two upstream gradient pairs; the
    // NEXT tile's are fetched while the current tile is processed
    struct TileIn {
        float xb0, xb1, ga, p0, p1, q0, q1;
    };
    auto fetch = [&](unsigned tile) -> TileIn {
        TileIn t{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (tile >= t1) return t;
        const unsigned row = tile * 16u + c;
        const bool ok = row < a.R;
        const unsigned rr = ok ? row : a.R - 1u;
        const float* __restrict__ xr = a.x + (size_t)rr * L;
        t.xb0 = (ok && g < L) ? xr[g] : 0.f;
        t.xb1 = (ok && 4 + g < L) ? xr[4 + g] : 0.f;
        if (GIN) {
            const bool d0 = ok && 2 * g < Dy, d1 = ok && 2 * g + 1 < Dy;
            const float m0 = a.gmean[(size_t)rr * Dy + (d0 ? 2 * g : 0)], v0 = a.gvar[(size_t)rr * Dy + (d0 ? 2 * g : 0)];
            const float m1 = a.gmean[(size_t)rr * Dy + (d1 ? 2 * g + 1 : 0)], v1 = a.gvar[(size_t)rr * Dy + (d1 ? 2 * g + 1 : 0)];
            t.p0 = d0 ? m0 : 0.f; t.p1 = d1 ? m1 : 0.f; t.q0 = d0 ? v0 * a.vscale : 0.f; t.q1 = d1 ? v1 * a.vscale : 0.f;
        } else {
            const RowMap rm = row_map(tile, c, a.S, a.K, invS, invK, ncells);
            t.ga = ok ? a.gA[rm.cell] : 0.f;
            t.p0 = 2 * g < Dy ? a.y[(size_t)rm.n * Dy + 2 * g] : 0.f;
            t.p1 = 2 * g + 1 < Dy ? a.y[(size_t)rm.n * Dy + 2 * g + 1] : 0.f;
        }
        return t;
    };
    TileIn nxt = fetch(t0);
    for (unsigned tile = t0; tile < t1; ++tile) {
        const unsigned row = tile * 16u + c;
        const bool ok = row < a.R;
        DEC_TT(0);
        const TileIn cur = nxt;
        nxt = fetch(tile + 1);
        const float xb0 = cur.xb0, xb1 = cur.xb1;
        const float ga = (!GIN && a.logw != 0.f) ? (ok ? a.logw * expf(cur.ga) : 0.f) : cur.ga;
        const float yv[2] = {cur.p0, cur.p1}, gin_m[2] = {cur.p0, cur.p1}, gin_v[2] = {cur.q0, cur.q1};

        unsigned xs[3];
        split_bf16<3>(v2f{xb0, xb1}, xs);
        const XOps xo = x_operands(xs);
        {   // x (h, m) terms transposed: lo half = dim g, hi half = dim 4+g of row c
            const int odd = c & 1;
            unsigned char* __restrict__ q = scrXb + (c >> 3) * 256 + 2 * ((c & 7) - odd) + trb_slot(odd ? 4 + g : g, c >> 3) * 16;
            const unsigned sel = odd ? 0x03020706u : 0x05040100u;
#pragma unroll
            for (int term = 0; term < 2; ++term) {
                const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)xs[term], 0xB1, 0xF, 0xF, true);
                *reinterpret_cast<unsigned*>(q + term * 512) = __builtin_amdgcn_perm(nb, xs[term], sel);
            }
        }
        DEC_TT(1);
        f32x4 h0[UT], h1[UT], O;
        {
            unsigned h0s[3][4 * KB], h1s[3][4 * KB];
            dec_forward_tile<UT, FS, (VMP_DEC_FWD_TERMS_FOLLOW_BT ? BT : 3)>(sm, lane, xo, onev, h0, h0s, h1, h1s, O);
            if constexpr (VMP_DEC_TR16) { trt_write<UT>(scrPb, tr_w, h0s); trt_write<UT>(scrQb, tr_w, h1s); }
            else { trb_write<UT>(scrPb, g, c, h0s); trb_write<UT>(scrQb, g, c, h1s); }
        }

        DEC_TT(2);
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
            llacc = rows4_sum(llacc);                            // over the 4 lane groups: two VALU lane swaps (was two ds_bpermute round trips)
            if (ok && g == 0) a.ll[row] = llacc;
        }
        unsigned dOs[3][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            unsigned t3[3];
            split_bf16<3>(v2f{dO[2 * p], dO[2 * p + 1]}, t3);
            dOs[0][p] = t3[0]; dOs[1][p] = t3[1]; dOs[2][p] = t3[2];
        }
        if constexpr (VMP_DEC_TR16) trt_write<1>(scrOb, tr_w, dOs); else trb_write<1>(scrOb, g, c, dOs);
        const SOps so = slot_operands(dOs);
        DEC_TT(3);
        // ---- dh1 = W2 . dO
        f32x4 dh1[UT];
        {
            u32x4 a0[UT], a1[UT];
#pragma unroll
            for (int tp = 0; tp < UT; ++tp) {
                a0[tp] = ldsu4(sm + I::B1 + tp * 512 + lane * 4);
                a1[tp] = ldsu4(sm + I::B1 + tp * 512 + 256 + lane * 4);
            }
#pragma unroll
            for (int tp = 0; tp < UT; ++tp) dh1[tp] = mfma_bf(a0[tp], so.hh, zero4);
#pragma unroll
            for (int tp = 0; tp < UT; ++tp) dh1[tp] = mfma_bf(a0[tp], so.mm, dh1[tp]);
            if constexpr (BT == 3) {
#pragma unroll
                for (int tp = 0; tp < UT; ++tp) dh1[tp] = mfma_bf(a1[tp], so.hl, dh1[tp]);
            }
        }
        DEC_TT(4);
        // ---- dW2 (and shortcut W): [h1 ; x]^T . dO
        wave_lds_order();
        u32x4 xT;
        {
            const u32x4 dTh = VMP_DEC_TR16 ? trt_read(scrOb, tr_o1, tr_o2) : trb_read(scrOb + rd_xx);
            const u32x4 dTm = VMP_DEC_TR16 ? trt_read(scrOb + 512, tr_o1, tr_o2) : trb_read(scrOb + rd_xx + 512);
            xT = trb_read(scrXb + rd_hm1);
#pragma unroll
            for (int t = 0; t < UT; ++t) {
                const u32x4 h1T = VMP_DEC_TR16 ? trt_read(scrQb + tr_hm + t * 512, tr_o1, tr_o2) : trb_read(scrQb + rd_hm + t * 512);
                aW2[t] = mfma_bf(h1T, dTh, aW2[t]);
                aW2[t] = mfma_bf(h1T, dTm, aW2[t]);
            }
            aWs = mfma_bf(xT, dTh, aWs);
            aWs = mfma_bf(xT, dTm, aWs);
        }
        DEC_TT(5);
        // ---- through tanh of layer 1
#pragma unroll
        for (int tp = 0; tp < UT; ++tp) {
#pragma unroll
            for (int v = 0; v < 4; ++v) dh1[tp][v] *= 1.0f - h1[tp][v] * h1[tp][v];
            if (!FS) ab1[tp] += dh1[tp];
        }
        wave_lds_order();
        DEC_TT(6);
        // ---- dh0 = W1 . dh1pre, and the terms of dh1pre transposed for dW1
        f32x4 dh0[UT];
#pragma unroll
        for (int tp = 0; tp < UT; ++tp) dh0[tp] = zero4;
        {
            unsigned d1s[3][4 * KB];
            split_tiles<UT, BT>(dh1, d1s);
            if constexpr (VMP_DEC_TR16) trt_write<UT>(scrQb, tr_w, d1s); else trb_write<UT>(scrQb, g, c, d1s);
            gemm_units<UT, UT, BT>(sm + I::B2, lane, d1s, dh0);
        }
        DEC_TT(7);
        // ---- dW1 = h0^T . dh1pre
        wave_lds_order();
        {
            u32x4 dTh[UT], dTm[UT];
#pragma unroll
            for (int t = 0; t < UT; ++t) {
                dTh[t] = VMP_DEC_TR16 ? trt_read(scrQb + t * 512, tr_o1, tr_o2) : trb_read(scrQb + rd_xx + t * 512);
                dTm[t] = VMP_DEC_TR16 ? trt_read(scrQb + (UT + t) * 512, tr_o1, tr_o2) : trb_read(scrQb + rd_xx + (UT + t) * 512);
            }
#pragma unroll
            for (int ti = 0; ti < UT; ++ti) {
                const u32x4 h0T = VMP_DEC_TR16 ? trt_read(scrPb + tr_hm + ti * 512, tr_o1, tr_o2) : trb_read(scrPb + rd_hm + ti * 512);
#pragma unroll
                for (int tj = 0; tj < UT; ++tj) aW1[ti][tj] = mfma_bf(h0T, dTh[tj], aW1[ti][tj]);
#pragma unroll
                for (int tj = 0; tj < UT; ++tj) aW1[ti][tj] = mfma_bf(h0T, dTm[tj], aW1[ti][tj]);
            }
        }
        DEC_TT(8);
        // ---- through tanh of layer 0
#pragma unroll
        for (int tp = 0; tp < UT; ++tp) {
#pragma unroll
            for (int v = 0; v < 4; ++v) dh0[tp][v] *= 1.0f - h0[tp][v] * h0[tp][v];
        }
        wave_lds_order();
        DEC_TT(9);
        // ---- dx = W0 . dh0pre + Ws . dO(mean)
        {
            unsigned d0s[3][4 * KB];
            split_tiles<UT, BT>(dh0, d0s);
            if constexpr (VMP_DEC_TR16) trt_write<UT>(scrPb, tr_w, d0s); else trb_write<UT>(scrPb, g, c, d0s);
            const f32x4 ds_ = gemm_slots<BT>(sm + I::B3S, lane, so, zero4);
            const f32x4 dxv = gemm_units_1<UT, BT>(sm + I::B3, lane, d0s, zero4) + ds_;        // [dim 4g+v][row c]
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
        DEC_TT(10);
        // ---- dW0 = x^T . dh0pre
        wave_lds_order();
#pragma unroll
        for (int tj = 0; tj < UT; ++tj) {
            const u32x4 dTh = VMP_DEC_TR16 ? trt_read(scrPb + tj * 512, tr_o1, tr_o2) : trb_read(scrPb + rd_xx + tj * 512);
            const u32x4 dTm = VMP_DEC_TR16 ? trt_read(scrPb + (UT + tj) * 512, tr_o1, tr_o2) : trb_read(scrPb + rd_xx + (UT + tj) * 512);
            aW0[tj] = mfma_bf(xT, dTh, aW0[tj]);
            aW0[tj] = mfma_bf(xT, dTm, aW0[tj]);
        }
        wave_lds_order();
        DEC_TT(11);
    }

    // ---- reduce the per-wave accumulators through LDS.  Every parameter index is owned by exactly one (lane,
    // register) of a wave, so each wave drops its accumulators into a slab of its own with plain stores (a turn-taking
    // read-modify-write over 8 waves cost ~50 us per launch - most of the kernel at minibatch sizes); then all threads
    // sum the slabs in wave order (deterministic).  One round when the 8 slabs fit the LDS (a.red_one: U <= 50 or so),
    // else two rounds of 4 - the same sequence of additions either way.  (At minibatch sizes this epilogue is a third of
    // the launch, and the two rounds ran one after the other: 8.0 of 16.7 us for the encoder's backward at 64 rows.)
    const DecGeo q = dec_geo(L, U, Dy);
    __syncthreads();                                    // everybody is done with the operand images
    DEC_TS(3);
    const int HALF = a.red_one ? BWD_WAVES : BWD_WAVES / 2;
    float* __restrict__ accum = sm;
    // waves without a tile (few-tile launches) hold all-zero accumulators: they neither store a slab nor are they read
    unsigned has = 0;
#pragma unroll
    for (int w = 0; w < BWD_WAVES; ++w) {
        const unsigned wp0 = min(b0 + (w & 3) * tpp, b1), wp1 = min(wp0 + tpp, b1);
        const unsigned wpm = min(wp0 + (tpp >= 8u ? (tpp * (unsigned)a.split + 50u) / 100u : (tpp + 1u) / 2u), wp1);
        if ((w < 4 ? wpm - wp0 : wp1 - wpm) > 0) has |= 1u << w;
    }
    for (int round = 0; round < BWD_WAVES / HALF; ++round) {
        if (wave / HALF == round && ((has >> wave) & 1u)) {
            float* __restrict__ slab = sm + (1 + wave % HALF) * q.PW;
            // Padding elements (units >= U, dims >= L ...) go to a dump word behind the last slab through an index
            // select, not around a branch: with ~100 conditional stores the exec-mask bookkeeping was 15 instructions
            // per store, SGPR spills included (1800 instructions: 3.5 us per launch at minibatch sizes).
            const int dump = (1 + HALF) * q.PW - (1 + wave % HALF) * q.PW;
            const int base1 = q.oW1 + 4 * g * U + c;
#pragma unroll
            for (int ti = 0; ti < UT; ++ti)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const bool okin = 16 * ti + 4 * g + v < U;
#pragma unroll
                    for (int tj = 0; tj < UT; ++tj) {
                        const bool ok = okin && 16 * tj + c < U;
                        slab[ok ? base1 + (16 * ti + v) * U + 16 * tj : dump] = aW1[ti][tj][v];
                    }
                }
            {
                const int d = slot_d(c), ty = slot_ty(c);
                const int base2 = q.oW2 + 4 * g * 2 * Dy + ty * Dy + d, base0 = q.oW0 + 4 * g * U + c;
#pragma unroll
                for (int t = 0; t < UT; ++t)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const bool ok2 = 16 * t + 4 * g + v < U && d < Dy;
                        slab[ok2 ? base2 + (16 * t + v) * 2 * Dy : dump] = aW2[t][v];
                        const bool ok0 = 4 * g + v < L && 16 * t + c < U;
                        slab[ok0 ? base0 + v * U + 16 * t : dump] = aW0[t][v];
                    }
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
        if (round == 0) DEC_TS(6);
        __syncthreads();
        if (round == 0) DEC_TS(7);
        // four parameters per thread and pass: their slab reads are in flight together
        for (int i0 = threadIdx.x; i0 < q.PW; i0 += 4 * BWD_THREADS) {
            float t[4];
            int ix[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ix[u] = i0 + u * BWD_THREADS < q.PW ? i0 + u * BWD_THREADS : i0;
                t[u] = round ? accum[ix[u]] : 0.f;
            }
            for (int w = 0; w < HALF; ++w) {
                if (!((has >> (round * HALF + w)) & 1u)) continue;            // wave-uniform
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] += sm[(1 + w) * q.PW + ix[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * BWD_THREADS < q.PW) accum[i0 + u * BWD_THREADS] = t[u];
        }
        __syncthreads();
    }
    // sigmoid(bs2) factor of d/d bs2 log1p(exp(bs2)) is applied by the reduce kernel
    DEC_TS(4);
    for (int i = threadIdx.x; i < q.PW; i += BWD_THREADS) a.part[(size_t)blockIdx.x * q.PW + i] = accum[i];
    DEC_TS(5);
}

// (partials -> parameter gradients: dec_reduce_sum, vmp_step_parts.h)
__device__ __forceinline__ void dec_reduce_body(const DecRedArgs& r, const int blk) {
    __shared__ double part[DEC_RED_GROUPS][64];
    const double s = dec_reduce_sum(r, blk, part);
    const int i = blk * 64 + (threadIdx.x & 63);
    if ((threadIdx.x >> 6) == 0 && i < r.PW) r.out[i] = (float)s;
}
__global__ __launch_bounds__(64 * DEC_RED_GROUPS) void dec_reduce_kernel(DecRedArgs r) { dec_reduce_body(r, blockIdx.x); }

// The reduction of the decoder's parameter partials and the scalar tail of the ELBO (vmp_tail.h) both wait for the decoder
// kernel and for nothing else: one launch, blocks [0, red_blocks) reduce, the others run the tail.
__global__ __launch_bounds__(WAVE) void dec_elbo_final_kernel(TailArgs t, unsigned ntb) { elbo_final_body(t, ntb); }
__global__ __launch_bounds__(64 * DEC_RED_GROUPS) void dec_reduce_tail_kernel(DecRedArgs r, TailArgs t, int red_blocks) {
    if ((int)blockIdx.x < red_blocks) dec_reduce_body(r, blockIdx.x);
    else elbo_tail_body(t, blockIdx.x - red_blocks, gridDim.x - red_blocks);
}

int dec_blocks(long long rows, int waves_per_block, int max_blocks) {
    const long long tiles = (rows + 15) / 16;
    long long b = (tiles + waves_per_block - 1) / waves_per_block;
    if (b > max_blocks) b = max_blocks;
    if (b < 1) b = 1;
    return (int)b;
}
#ifdef VMP_DEBUG_TS
static long long* g_dbg_dec = nullptr;
#endif
int dec_fwd_blocks(long long rows) {
    constexpr int bpc = 16;   // blocks per CU: 4 are resident (82 VGPRs, 39 KB LDS), the rest back-fill as the older blocks - which the sequencer favours - finish (2 -> 4 -> 16: 4.5 -> 3.9 -> 3.5 ms per 4.2e7 rows)
    return dec_blocks(rows, FWD_THREADS / WAVE, 256 * bpc);
}
// 1 block per CU.  Few tiles (<= 4 per CU): one tile per SIMD - waves 0-3 of a block take one each and waves 4-7 none, so
// that a block's epilogue (every wave with work stores a PW-word slab, then the slabs are summed) runs with one wave per
// SIMD and four slabs; with 8 one-tile waves per block it was 40 % of the launch at the reference's minibatch sizes.
int dec_bwd_blocks(long long rows) {
    const long long tiles = (rows + 15) / 16;
    return tiles <= 4 * 256 ? dec_blocks(rows, BWD_WAVES / 2, 256) : dec_blocks(rows, BWD_WAVES, 256);
}

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

// dispatch on UT = unit tiles of 16
#define DEC_DISPATCH(U, CALL)                          \
    do {                                               \
        switch (((U) + 15) / 16) {                     \
            case 1: CALL(1); break;                    \
            case 2: CALL(2); break;                    \
            case 3: CALL(3); break;                    \
            default: CALL(4); break;                   \
        }                                              \
    } while (0)

template <bool GIN>
int dec_bwd_launch(const DecArgs& a0, int blocks, hipStream_t s) {
    // measured optimum at 4.2e7 rows: 58 % for U = 50 (10.0 -> 9.3 ms), 54 % for U = 64 (11.8 -> 11.3 ms)
    DecArgs a = a0;
#ifdef VMP_DEBUG_TS
    a.dbg_t = g_dbg_dec;
#endif
    a.split = (a0.U & 15) ? 58 : 54;
    const int U = a.U;
    const int PWl = dec_geo(a.L, a.U, a.Dy).PW;
    a.red_one = ((size_t)(1 + BWD_WAVES) * PWl + 64) * sizeof(float) <= 160u * 1024u;     // accumulator + 8 slabs inside the CU's LDS
    const int red_floats = (1 + (a.red_one ? BWD_WAVES : BWD_WAVES / 2)) * PWl + 64;   // epilogue: accumulator + slabs + dump word
#ifndef VMP_DEC_BT2_ROWS
#define VMP_DEC_BT2_ROWS (1u << 19)   // sample rows from which the backward data path uses 2-term operands (0xffffffff: never)
#endif
    const bool bt2 = a.R >= (unsigned)VMP_DEC_BT2_ROWS;
#define DEC_BWD_L(UTV, FSV, BTV)                                                                                      \
    do {                                                                                                              \
        if (const int rc_ = set_dyn_lds(reinterpret_cast<const void*>(dec_bwd_kernel<UTV, FSV, GIN, BTV>), (size_t)lds, "dec_bwd_kernel")) return rc_; \
        hipLaunchKernelGGL((dec_bwd_kernel<UTV, FSV, GIN, BTV>), dim3(blocks), dim3(BWD_THREADS), lds, s, a);         \
    } while (0)
#define DEC_BWD(UTV)                                                                                                  \
    do {                                                                                                              \
        const int lds = (Img<UTV>::BWD_TOTAL > red_floats ? Img<UTV>::BWD_TOTAL : red_floats) * (int)sizeof(float);   \
        if ((U & 15) == 0) { if (bt2) DEC_BWD_L(UTV, false, 2); else DEC_BWD_L(UTV, false, 3); }                      \
        else { if (bt2) DEC_BWD_L(UTV, true, 2); else DEC_BWD_L(UTV, true, 3); }                                      \
    } while (0)
    DEC_DISPATCH(U, DEC_BWD);
#undef DEC_BWD
#undef DEC_BWD_L
    return check_launch(GIN ? "vmp_mlp_gauss_bwd" : "vmp_decoder_loglike_bwd");
}

// lazy: the per-block parameter partials stay in `ws` ((vmp_decoder_bwd_blocks, PW) fp32) for a later launch to reduce - no reduce launch here
int decoder_loglike_bwd_impl(const char* what, float logw, const TailArgs* tail, unsigned tail_blocks, bool lazy, const float* x, const float* y, const float* gA,
                             const float* W0, const float* b0, const float* W1, const float* b1, const float* W2, const float* b2, const float* Ws,
                            const float* bs1, const float* bs2, int64_t N, int K, int S, int L, int Dy, int U, float* dx,
                            float* dparams, float* ll, void* ws, size_t ws_bytes, void* stream) {
    if (int e = dec_check(what, N, K, S, L, Dy, U)) return e;
    if (!x || !y || !gA || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !Ws || !bs1 || !bs2 || !dx || (!dparams && !lazy) || !ws) {
        set_error("%s: NULL argument", what);
        return VMP_E_BADARG;
    }
    const DecGeo q = dec_geo(L, U, Dy);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N == 0) {
        if (lazy) { set_error("%s: N = 0", what); return VMP_E_DIM; }
        (void)hipMemsetAsync(dparams, 0, (size_t)q.PW * sizeof(float), s);
        return check_launch(what);
    }
    const size_t need = (size_t)dec_bwd_blocks((long long)N * K * S) * (size_t)q.PW * sizeof(float);
    if (ws_bytes < need) {
        set_error("%s: workspace too small (%zu < %zu bytes)", what, ws_bytes, need);
        return VMP_E_WS;
    }
    DecArgs a{};
    a.x = x; a.y = y; a.gA = gA; a.W0 = W0; a.b0 = b0; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.Ws = Ws; a.bs1 = bs1; a.bs2 = bs2;
    a.dx = dx; a.part = static_cast<float*>(ws); a.ll = ll; a.logw = logw; a.vscale = 1.0f;
    a.R = (unsigned)(N * K * S); a.K = (unsigned)K; a.S = (unsigned)S; a.L = L; a.Dy = Dy; a.U = U;
    const int blocks = dec_bwd_blocks((long long)a.R);
    if (int e = dec_bwd_launch<false>(a, blocks, s)) return e;
    if (lazy) return 0;
    DecRedArgs r{a.part, bs2, dparams, blocks, q.PW, q.obs2, Dy};
    const int red_blocks = (q.PW + 63) / 64;
    if (tail) {
        hipLaunchKernelGGL(dec_reduce_tail_kernel, dim3(red_blocks + tail_blocks), dim3(64 * DEC_RED_GROUPS), 0, s, r, *tail, red_blocks);
        if (tail_blocks > 1) hipLaunchKernelGGL(dec_elbo_final_kernel, dim3(1), dim3(WAVE), 0, s, *tail, tail_blocks);      // the three scalars, behind a kernel boundary (one tail block: written by that block)
    }
    else hipLaunchKernelGGL(dec_reduce_kernel, dim3(red_blocks), dim3(64 * DEC_RED_GROUPS), 0, s, r);
    return check_launch(what);
}

int decoder_fwd_impl(const char* what, float vscale, const float* x, const float* y, const float* W0, const float* b0, const float* W1,
                            const float* b1, const float* W2, const float* b2, const float* Ws, const float* bs1,
                            const float* bs2, int64_t N, int K, int S, int L, int Dy, int U, float* ll, float* mean,
                            float* var, void* stream) {
    if (int e = dec_check(what, N, K, S, L, Dy, U)) return e;
    if (!x || (!y && ll) || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !Ws || !bs1 || !bs2 || (!ll && !mean) || (!mean != !var)) {
        set_error("%s: NULL argument", what);
        return VMP_E_BADARG;
    }
    if (N == 0) return 0;
    DecArgs a{};
    a.x = x; a.y = y; a.W0 = W0; a.b0 = b0; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.Ws = Ws; a.bs1 = bs1; a.bs2 = bs2;
    a.ll = ll; a.mean = mean; a.var = var; a.vscale = vscale;
    a.R = (unsigned)(N * K * S); a.K = (unsigned)K; a.S = (unsigned)S; a.L = L; a.Dy = Dy; a.U = U;
    const int blocks = dec_fwd_blocks((long long)a.R);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define DEC_FWD(UTV)                                                                                                  \
    do {                                                                                                              \
        const int lds = Img<UTV>::FWD_END * (int)sizeof(float);                                                       \
        if (const int rc_ = set_dyn_lds(reinterpret_cast<const void*>(dec_fwd_kernel<UTV>), (size_t)lds, "dec_fwd_kernel")) return rc_; \
        hipLaunchKernelGGL((dec_fwd_kernel<UTV>), dim3(blocks), dim3(FWD_THREADS), lds, s, a);                        \
    } while (0)
    DEC_DISPATCH(U, DEC_FWD);
#undef DEC_FWD
    return check_launch(what);
}

int mlp_gauss_bwd_impl(const char* what, float vscale, bool lazy, const float* x, const float* gmean, const float* gvar, const float* W0, const float* b0,
                      const float* W1, const float* b1, const float* W2, const float* b2, const float* Ws, const float* bs1,
                      const float* bs2, int64_t R, int L, int Dy, int U, float* dx, float* dparams, void* ws,
                      size_t ws_bytes, void* stream) {
    if (int e = dec_check(what, R, 1, 1, L, Dy, U)) return e;
    if (!x || !gmean || !gvar || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !Ws || !bs1 || !bs2 || (!dparams && !lazy) || !ws) {
        set_error("%s: NULL argument", what);
        return VMP_E_BADARG;
    }
    const DecGeo q = dec_geo(L, U, Dy);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (R == 0) {
        if (lazy) { set_error("%s: R = 0", what); return VMP_E_DIM; }
        (void)hipMemsetAsync(dparams, 0, (size_t)q.PW * sizeof(float), s);
        return check_launch(what);
    }
    if (ws_bytes < (size_t)dec_bwd_blocks((long long)R) * (size_t)q.PW * sizeof(float)) {
        set_error("%s: workspace too small", what);
        return VMP_E_WS;
    }
    DecArgs a{};
    a.x = x; a.gmean = gmean; a.gvar = gvar; a.W0 = W0; a.b0 = b0; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.Ws = Ws;
    a.bs1 = bs1; a.bs2 = bs2; a.dx = dx; a.part = static_cast<float*>(ws); a.vscale = vscale;
    a.R = (unsigned)R; a.K = 1; a.S = 1; a.L = L; a.Dy = Dy; a.U = U;
    const int blocks = dec_bwd_blocks((long long)a.R);
    if (int e = dec_bwd_launch<true>(a, blocks, s)) return e;
    if (lazy) return 0;
    DecRedArgs r{a.part, bs2, dparams, blocks, q.PW, q.obs2, Dy};
    hipLaunchKernelGGL(dec_reduce_kernel, dim3((q.PW + 63) / 64), dim3(64 * DEC_RED_GROUPS), 0, s, r);
    return check_launch(what);
}

}  // namespace

extern "C" {

#ifdef VMP_DEBUG_TS
void vmp_debug_set_decoder_timestamps(long long* p) { g_dbg_dec = p; }    // exploration builds only (tools/dec_ts.py)
#endif
int vmp_decoder_param_words(int L, int U, int Dy) { return dec_geo(L, U, Dy).PW; }

size_t vmp_decoder_workspace_bytes(int64_t N, int K, int S, int L, int U, int Dy) {
    return (size_t)dec_bwd_blocks((long long)N * K * S) * (size_t)dec_geo(L, U, Dy).PW * sizeof(float);
}

int vmp_decoder_loglike_fwd(const float* x, const float* y, const float* W0, const float* b0, const float* W1,
                            const float* b1, const float* W2, const float* b2, const float* Ws, const float* bs1,
                            const float* bs2, int64_t N, int K, int S, int L, int Dy, int U, float* ll, float* mean,
                            float* var, void* stream) {
    return decoder_fwd_impl("vmp_decoder_loglike_fwd", 1.0f, x, y, W0, b0, W1, b1, W2, b2, Ws, bs1, bs2, N, K, S, L, Dy, U, ll, mean, var,
                            stream);
}

int vmp_mlp_gauss_head_fwd(const float* x, const float* W0, const float* b0, const float* W1, const float* b1, const float* W2,
                           const float* b2, const float* Ws, const float* bs1, const float* bs2, int64_t R, int L, int Dy, int U,
                           float var_scale, float* out1, float* out2, void* stream) {
    return decoder_fwd_impl("vmp_mlp_gauss_head_fwd", var_scale, x, nullptr, W0, b0, W1, b1, W2, b2, Ws, bs1, bs2, R, 1, 1, L, Dy, U,
                            nullptr, out1, out2, stream);
}

// ---- round 6: encoder forward + recognition unpacking + theta packing (+ the replayed step's scalars) in ONE launch
int vmp_mlp_gauss_head_fwd_prep(const float* x, const float* W0, const float* b0, const float* W1, const float* b1, const float* W2,
                                const float* b2, const float* Ws, const float* bs1, const float* bs2, int64_t R, int L, int Dy, int U,
                                float var_scale, float* out1, float* out2, const float* mu_k, const float* L_raw, const float* pi_raw,
                                const float* alpha, const float* A, const float* b, const float* beta, const float* v_hat, int K,
                                float* Lk, float* P, float* bias, float* m, float* W, float* kappa, double* logpi,
                                const void* scalar_table, int table_rows, void* counter, void* dst16, void* stream) {
    const char* what = "vmp_mlp_gauss_head_fwd_prep";
    if (int e = dec_check(what, R, 1, 1, L, Dy, U)) return e;
    if (R < 1 || K < 1 || K > VMP_MAX_K) { set_error("%s: R = %lld, K = %d", what, (long long)R, K); return VMP_E_DIM; }
    if (!x || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !Ws || !bs1 || !bs2 || !out1 || !out2 || !mu_k || !L_raw || !pi_raw || !alpha || !A ||
        !b || !beta || !v_hat || !Lk || !P || !bias || !m || !W || !kappa) {
        set_error("%s: NULL argument", what);
        return VMP_E_BADARG;
    }
    if (scalar_table && (table_rows < 1 || !counter || !dst16 || (reinterpret_cast<uintptr_t>(dst16) & 7) || (reinterpret_cast<uintptr_t>(scalar_table) & 7))) {
        set_error("%s: scalar table without rows / counter / 8-byte aligned destination", what);
        return VMP_E_BADARG;
    }
    DecArgs a{};
    a.x = x; a.W0 = W0; a.b0 = b0; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.Ws = Ws; a.bs1 = bs1; a.bs2 = bs2;
    a.mean = out1; a.var = out2; a.vscale = var_scale;
    a.R = (unsigned)R; a.K = 1; a.S = 1; a.L = L; a.Dy = Dy; a.U = U;
    PhiArgs p{};
    p.mu = mu_k; p.Lraw = L_raw; p.piraw = pi_raw; p.Lk = Lk; p.P = P; p.bias = bias; p.K = K; p.L = Dy; p.logpi_out = logpi;
    ThetaArgs t{alpha, A, b, beta, v_hat, m, W, kappa, K, Dy};
    StepTable st{static_cast<const unsigned long long*>(scalar_table), static_cast<unsigned long long*>(counter),
                 static_cast<unsigned long long*>(dst16), (unsigned)(table_rows > 0 ? table_rows : 0)};
    const unsigned nb = (unsigned)dec_fwd_blocks((long long)a.R);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define ENC_PREP_L(UTV, LP)                                                                                           \
    do {                                                                                                              \
        const int lds = Img<UTV>::FWD_END * (int)sizeof(float);                                                       \
        if (const int rc_ = set_dyn_lds(reinterpret_cast<const void*>(enc_prep_kernel<UTV, LP>), (size_t)lds, "enc_prep_kernel")) return rc_; \
        hipLaunchKernelGGL((enc_prep_kernel<UTV, LP>), dim3(nb + 2 * K), dim3(FWD_THREADS), lds, s, a, p, t, st, nb);  \
    } while (0)
#define ENC_PREP(UTV)                                                                                                 \
    do {                                                                                                              \
        switch (Dy) {                                                                                                 \
            case 1: ENC_PREP_L(UTV, 1); break; case 2: ENC_PREP_L(UTV, 2); break; case 3: ENC_PREP_L(UTV, 3); break;  \
            case 4: ENC_PREP_L(UTV, 4); break; case 5: ENC_PREP_L(UTV, 5); break; case 6: ENC_PREP_L(UTV, 6); break;  \
            case 7: ENC_PREP_L(UTV, 7); break; default: ENC_PREP_L(UTV, 8); break;                                    \
        }                                                                                                             \
    } while (0)
    DEC_DISPATCH(U, ENC_PREP);
#undef ENC_PREP
#undef ENC_PREP_L
    return check_launch(what);
}

int vmp_decoder_loglike_bwd(const float* x, const float* y, const float* gA, const float* W0, const float* b0,
                            const float* W1, const float* b1, const float* W2, const float* b2, const float* Ws,
                            const float* bs1, const float* bs2, int64_t N, int K, int S, int L, int Dy, int U, float* dx,
                            float* dparams, float* ll, void* ws, size_t ws_bytes, void* stream) {
    return decoder_loglike_bwd_impl("vmp_decoder_loglike_bwd", 0.f, nullptr, 0, false, x, y, gA, W0, b0, W1, b1, W2, b2, Ws, bs1, bs2, N, K, S, L, Dy, U,
                                    dx, dparams, ll, ws, ws_bytes, stream);
}

int vmp_decoder_loglike_bwd_logw(const float* x, const float* y, const float* log_w, float w_scale, const float* W0,
                                 const float* b0, const float* W1, const float* b1, const float* W2, const float* b2,
                                 const float* Ws, const float* bs1, const float* bs2, int64_t N, int K, int S, int L, int Dy,
                                 int U, float* dx, float* dparams, float* ll, void* ws, size_t ws_bytes, void* stream) {
    if (w_scale == 0.f) {
        set_error("vmp_decoder_loglike_bwd_logw: w_scale must not be 0");
        return VMP_E_BADARG;
    }
    return decoder_loglike_bwd_impl("vmp_decoder_loglike_bwd_logw", w_scale, nullptr, 0, false, x, y, log_w, W0, b0, W1, b1, W2, b2, Ws, bs1, bs2, N, K,
                                    S, L, Dy, U, dx, dparams, ll, ws, ws_bytes, stream);
}

int vmp_decoder_elbo(const float* x, const float* y, const float* log_z, const float* T_prime, float sigma, const float* W0,
                     const float* b0, const float* W1, const float* b1, const float* W2, const float* b2, const float* Ws,
                     const float* bs1, const float* bs2, int64_t N, int K, int S, int L, int Dy, int U, float* dx,
                     float* dparams, float* ll, float* scalars, float* g_log_z, float* g_T_prime, float* r, void* ws,
                     size_t ws_bytes, void* tail_ws, size_t tail_ws_bytes, void* stream) {
    if (sigma == 0.f || !ll || !log_z || !T_prime || !scalars || !g_log_z || !g_T_prime || !r || !tail_ws) {
        set_error("vmp_decoder_elbo: NULL argument or sigma == 0");
        return VMP_E_BADARG;
    }
    if (tail_ws_bytes < tail_workspace_bytes()) {
        set_error("vmp_decoder_elbo: tail workspace too small (%zu < %zu bytes)", tail_ws_bytes, tail_workspace_bytes());
        return VMP_E_WS;
    }
    if (N <= 0 || K < 1 || S < 1) {
        set_error("vmp_decoder_elbo: N = %lld, K = %d, S = %d", (long long)N, K, S);
        return VMP_E_DIM;
    }
    TailArgs t{};
    const unsigned tb = tail_setup(t, log_z, T_prime, ll, N, K, S, Dy, sigma, scalars, g_log_z, g_T_prime, r, tail_ws, 64 * DEC_RED_GROUPS);
    return decoder_loglike_bwd_impl("vmp_decoder_elbo", -sigma * 0.5f / (float)S, &t, tb, false, x, y, log_z, W0, b0, W1, b1, W2, b2, Ws, bs1,
                                    bs2, N, K, S, L, Dy, U, dx, dparams, ll, ws, ws_bytes, stream);
}

int vmp_mlp_gauss_bwd(const float* x, const float* gmean, const float* gvar, const float* W0, const float* b0,
                      const float* W1, const float* b1, const float* W2, const float* b2, const float* Ws, const float* bs1,
                      const float* bs2, int64_t R, int L, int Dy, int U, float* dx, float* dparams, void* ws,
                      size_t ws_bytes, void* stream) {
    return mlp_gauss_bwd_impl("vmp_mlp_gauss_bwd", 1.0f, false, x, gmean, gvar, W0, b0, W1, b1, W2, b2, Ws, bs1, bs2, R, L, Dy, U, dx, dparams,
                              ws, ws_bytes, stream);
}

int vmp_mlp_gauss_head_bwd(const float* x, const float* g_out1, const float* g_out2, float var_scale, const float* W0,
                           const float* b0, const float* W1, const float* b1, const float* W2, const float* b2, const float* Ws,
                           const float* bs1, const float* bs2, int64_t R, int L, int Dy, int U, float* dx, float* dparams,
                           void* ws, size_t ws_bytes, void* stream) {
    return mlp_gauss_bwd_impl("vmp_mlp_gauss_head_bwd", var_scale, false, x, g_out1, g_out2, W0, b0, W1, b1, W2, b2, Ws, bs1, bs2, R, L, Dy, U,
                              dx, dparams, ws, ws_bytes, stream);
}

// ---- round 6: the minibatch training step reduces the parameter partials inside its ONE closing launch (vmp_svae_step_final,
// vmp_step.hip); these entry points run the fused MLP backward kernel alone and leave (vmp_decoder_bwd_blocks(rows), PW) fp32
// partials in `ws`.
int vmp_decoder_bwd_blocks(int64_t rows) { return rows > 0 ? dec_bwd_blocks((long long)rows) : 0; }

int vmp_decoder_elbo_lazy(const float* x, const float* y, const float* log_z, float sigma, const float* W0, const float* b0,
                          const float* W1, const float* b1, const float* W2, const float* b2, const float* Ws, const float* bs1,
                          const float* bs2, int64_t N, int K, int S, int L, int Dy, int U, float* dx, float* ll, void* ws,
                          size_t ws_bytes, void* stream) {
    if (sigma == 0.f || !ll || !log_z) {
        set_error("vmp_decoder_elbo_lazy: NULL argument or sigma == 0");
        return VMP_E_BADARG;
    }
    if (N <= 0 || K < 1 || S < 1) {
        set_error("vmp_decoder_elbo_lazy: N = %lld, K = %d, S = %d", (long long)N, K, S);
        return VMP_E_DIM;
    }
    return decoder_loglike_bwd_impl("vmp_decoder_elbo_lazy", -sigma * 0.5f / (float)S, nullptr, 0, true, x, y, log_z, W0, b0, W1, b1, W2, b2,
                                    Ws, bs1, bs2, N, K, S, L, Dy, U, dx, nullptr, ll, ws, ws_bytes, stream);
}

int vmp_mlp_gauss_head_bwd_lazy(const float* x, const float* g_out1, const float* g_out2, float var_scale, const float* W0,
                                const float* b0, const float* W1, const float* b1, const float* W2, const float* b2,
                                const float* Ws, const float* bs1, const float* bs2, int64_t R, int L, int Dy, int U, float* dx,
                                void* ws, size_t ws_bytes, void* stream) {
    return mlp_gauss_bwd_impl("vmp_mlp_gauss_head_bwd_lazy", var_scale, true, x, g_out1, g_out2, W0, b0, W1, b1, W2, b2, Ws, bs1, bs2, R, L,
                              Dy, U, dx, nullptr, ws, ws_bytes, stream);
}

}  // extern "C"
