// Shared host/device helpers for libvmp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/vmp_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace vmp {

constexpr int WAVE = 64;

typedef float v2f __attribute__((ext_vector_type(2)));

// ---- bf16 operand splitting for the XDL matrix pipe (v_mfma_f32_16x16x32_bf16) ------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// bf16(lo) | bf16(hi) << 16, round-to-nearest-even: one v_cvt_pk_bf16_f32.  Deliberately NOT inline assembly: the
// compiler's hazard recogniser does not see register writes inside an asm block, and a cvt that overwrites a register an
// in-flight MFMA still reads as its C operand (write-after-read, needs wait states) corrupted one instantiation of the
// decoder kernel.
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v2f{lo, hi}, bf16x2));
}
// v = h + m + l + O(2^-26 |v|): three bf16 terms (8 significant bits each) per value, packed per pair of values.  The
// moment GEMM keeps the six products of order <= 2 (hh, hm, mh, hl, lh, mm); what it drops (ml, lm, ll and the split
// residuals) is below 2^-24 |w phi| - fp32 accuracy, as the fp32 MFMA it replaces (a two-term split, three products,
// is ~2^-17 per product: fine on average at N = 1e6 but visible - 1.4e-5 on r - on a 60-row problem).
template <int TERMS>
__device__ __forceinline__ void split_bf16(v2f v, unsigned (&t)[3]) {
    t[0] = cvt_pk_bf16(v.x, v.y);
    v2f rem = v - v2f{__uint_as_float(t[0] << 16), __uint_as_float(t[0] & 0xffff0000u)};
    t[1] = cvt_pk_bf16(rem.x, rem.y);
    if constexpr (TERMS == 3) {
        rem = rem - v2f{__uint_as_float(t[1] << 16), __uint_as_float(t[1] & 0xffff0000u)};
        t[2] = cvt_pk_bf16(rem.x, rem.y);
    } else {
        t[2] = 0u;
    }
}

// ---- packed fp32 (v_pk_*_f32) against a broadcast parameter -------------------------------------------
// x holds two independent values (two data rows, or two samples of one cell); the parameter is ONE float,
// stored two to a 64-bit register pair and broadcast to both halves with op_sel.  A {p, p} splat written in
// C++ is hoisted out of the loop by the compiler and doubles the resident parameter registers.  h selects
// the half of `p` that holds the parameter (a compile-time constant after unrolling).
// HARDWARE NOTE (measured on MI355X, tools/ubench/pk_beside_mfma.hip): a VOP3P packed-fp32 instruction whose LOW result
// reads the HIGH half of src1 (op_sel[1] = 1) returns a wrong low result in lanes 48-63, about once per 1e6
// executions, while ANOTHER wave of the same SIMD is executing bf16 (XDL) MFMAs.  op_sel on src0 or src2, op_sel_hi on
// any source, the un-modified forms and fp32-MFMA partners are all clean.  The h = 1 forms below therefore pass the
// parameter as src0 (the products commute; the subtraction negates src0 instead of src1).
__device__ __forceinline__ v2f pk_fma_b(v2f x, v2f p, v2f acc, int h) {            // acc + x * p.h
    v2f d;
    if (h) asm("v_pk_fma_f32 %0, %2, %1, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(x), "v"(p), "v"(acc));
    else   asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(x), "v"(p), "v"(acc));
    return d;
}
__device__ __forceinline__ v2f pk_fnma_b(v2f x, v2f p, v2f acc, int h) {           // acc - x * p.h
    v2f d;
    if (h) asm("v_pk_fma_f32 %0, %2, %1, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(x), "v"(p), "v"(acc));
    else   asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(d) : "v"(x), "v"(p), "v"(acc));
    return d;
}
__device__ __forceinline__ v2f pk_mul_b(v2f x, v2f p, int h) {                     // x * p.h
    v2f d;
    if (h) asm("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "v"(p));
    else   asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(p));
    return d;
}
__device__ __forceinline__ v2f pk_add_b(v2f x, v2f p, int h) {                     // x + p.h
    v2f d;
    if (h) asm("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "v"(p));
    else   asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(p));
    return d;
}
__device__ __forceinline__ v2f pk_sub_b(v2f x, v2f p, int h) {                     // x - p.h
    v2f d;
    if (h) asm("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[1,0]" : "=v"(d) : "v"(x), "v"(p));
    else   asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(p));
    return d;
}
// c.lo - q * c.hi on both halves of q  (c = {constant, scale}); c.hi enters as src0 (see the note above)
__device__ __forceinline__ v2f pk_const_minus_scaled(v2f q, v2f c) {
    v2f d;
    asm("v_pk_fma_f32 %0, %2, %1, %2 op_sel:[1,0,0] op_sel_hi:[1,1,0] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(d) : "v"(q), "v"(c));
    return d;
}

// log(1 + x) without the device library's log1pf: that routine's double-float arithmetic is SLP-packed by the compiler
// into v_pk_add_f32 ... op_sel:[0,1], the very operand form of the hardware note above (tools/erratum_scan.py found it
// in every kernel that called log1pf).  log(u) x / (u - 1) with u = fl(1 + x) cancels the rounding of the addition
// (the HP-15C identity): <= 2 ulp with the library's logf, and no packed instruction.
__device__ __forceinline__ float log1p_f(float x) {
    const float u = 1.0f + x;
    const float d = u - 1.0f;
    if (d == 0.f) return x;                  // |x| < 2^-24: log1p(x) = x to working precision
    if (u == INFINITY) return u;
    return logf(u) * (x / d);
}

// all-reduce over the 16 lanes of a DPP row by row rotations, one value per lane.  The two wait states a DPP read
// needs after a VALU write of the same VGPR are explicit (hipcc's hazard recogniser does not look inside asm).
#define VMP_DPP1(OP, CTRL) "v_" OP "_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
__device__ __forceinline__ float row16_max(float v) {
    asm("s_nop 1\n\t" VMP_DPP1("max", "row_ror:8") VMP_DPP1("max", "row_ror:4") VMP_DPP1("max", "row_ror:2") VMP_DPP1("max", "row_ror:1") : "+v"(v));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    asm("s_nop 1\n\t" VMP_DPP1("add", "row_ror:8") VMP_DPP1("add", "row_ror:4") VMP_DPP1("add", "row_ror:2") VMP_DPP1("add", "row_ror:1") : "+v"(v));
    return v;
}

// sum over lanes l, l^16, l^32, l^48 (same component, 4 rows) on the VALU: gfx950's v_permlane32_swap / v_permlane16_swap
// exchange the 32-lane halves / the odd and even 16-lane rows of two registers, so (lower + upper) and then (even + odd) are
// two swaps and two adds - the same sums, bit for bit, as v + shfl_xor(v, 32) and t + shfl_xor(t, 16), which compile to
// ds_bpermute_b32: an address computation, an LDS-crossbar round trip and a wait per exchange, 90 of them per tile.
#ifndef VMP_ROWS4_SHFL
#define VMP_ROWS4_SHFL 0      // 1: the ds_bpermute form (A/B measurements: tools/build_variant.sh)
#endif
__device__ __forceinline__ float rows4_sum(float v) {
#if VMP_ROWS4_SHFL
    const float t = v + __shfl_xor(v, 32);
    return t + __shfl_xor(t, 16);
#else
    const unsigned x = __float_as_uint(v);
    const auto h = __builtin_amdgcn_permlane32_swap(x, x, false, false);      // h[0] = lower-half values, h[1] = upper-half values
    const float t = __uint_as_float(h[0]) + __uint_as_float(h[1]);
    const unsigned y = __float_as_uint(t);
    const auto q = __builtin_amdgcn_permlane16_swap(y, y, false, false);      // q[0] = even-row values, q[1] = odd-row values
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
#endif
}

// ---- error plumbing (thread-local message) ------------------------------------------------
void set_error(const char* fmt, ...);
int  check_launch(const char* what);
// LDS a workgroup may ask for on the current device (hipDeviceAttributeMaxSharedMemoryPerBlock, at most the 160 KB of a gfx950
// CU); the launch plans size their per-wave buffers against it instead of a constant.
size_t lds_budget();
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) with its status checked: 0, or the HIP error code with the message set
int  set_dyn_lds(const void* kernel, size_t bytes, const char* what);

// ---- geometry of the mixture kernels -----------------------------------------------------------------
template <int D>
struct Geo {
    static constexpr int TRI   = D * (D + 1) / 2;
    static constexpr int F     = 1 + D + TRI;        // MFMA feature columns  [1 | x_d | x_d x_e (d<=e)]
    static constexpr int FT    = (F + 15) / 16;      // 16-wide feature tiles
    static constexpr int PACK  = D + TRI + 4;        // E-step parameter words per component
    static constexpr int SW    = 2 + D + D * D;      // public stats words per component
    static constexpr int PF    = F + 1;              // partial words per component: features + Nk
    static constexpr int XROWS = D + 2;              // LDS rows: x_0..x_{D-1}, ones, zeros
};

inline int pack_words(int D)  { return D + D * (D + 1) / 2 + 4; }
inline int stats_words(int D) { return 2 + D + D * D; }
inline int partial_words(int D) { return 1 + D + D * (D + 1) / 2 + 1; }

// ---- raw moments of a SMALL batch ------------------------------------------------------------------------
// (the training step at the reference's minibatch sizes, experiments.py:26: 64-100 rows): one block of 12 x 80 threads per
// component, thread (row group g, statistic i) sums the rows n = g (mod 12) in fp64 on the un-shifted data, the group sums
// are added in a fixed order.  One launch of a few microseconds instead of pivot + streaming pass + reduction (three
// launches, ~22 us at N = 64, most of it fixed cost).  Shared by vmp_mix_stats (vmp_mix.hip) and the fused moments + CVI
// update of the SVAE training step (vmp_prep.hip).
struct SmallStatsArgs { const float* x; const float* r; const float* u; double* stats; int N, D, K; };
constexpr int SMALL_STATS_MAX_N = 512;
constexpr int SMALL_STATS_GROUPS = 12;          // row groups per component: 12 x 80 threads, <= 6 rows each at N = 64
// statistic i (< 2 + D + D*D) of component k in the threads of row group 0 (threadIdx.x < 80); other threads: garbage
__device__ __forceinline__ double small_stats_component(const SmallStatsArgs& a, int k, double (*part)[80]) {
    const int g = threadIdx.x / 80, i = threadIdx.x % 80;
    const int D = a.D, SW = 2 + D + D * D;
    double s = 0.0;
    if (i < SW && g < SMALL_STATS_GROUPS) {                 // (blocks of more than 12 x 80 threads: the surplus threads idle)
        const int d = i < 2 + D ? (i < 2 ? 0 : i - 2) : (i - 2 - D) / D, e = i < 2 + D ? 0 : (i - 2 - D) % D;
#pragma unroll 2
        for (int n = g; n < a.N; n += SMALL_STATS_GROUPS) {
            const float rf = a.r[(long long)n * a.K + k];
            const float wf = a.u ? rf * a.u[(long long)n * a.K + k] : rf;          // w = r u in fp32, as the pass kernel forms it
            const double xd = (double)a.x[(long long)n * D + d], xe = (double)a.x[(long long)n * D + e];
            double t;
            if (i == 0) t = rf;
            else if (i == 1) t = wf;
            else if (i < 2 + D) t = (double)wf * xd;
            else t = (double)wf * xd * xe;
            s += t;
        }
    }
    if (g < SMALL_STATS_GROUPS) part[g][i] = s;
    __syncthreads();
    double t = part[0][i];
#pragma unroll
    for (int j = 1; j < SMALL_STATS_GROUPS; ++j) t += part[j][i];
    return t;
}

}  // namespace vmp
