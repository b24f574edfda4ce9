// Does a packed-fp32 VALU instruction return wrong results when ANOTHER wave on the same SIMD runs bf16 MFMAs?
// Block = 8 waves: waves 0-3 (one per SIMD) run VARIANT of v_pk_fma_f32 in a loop and compare every result with the
// scalar v_fma_f32 reference; waves 4-7 (their SIMD partners) spin on v_mfma_f32_16x16x32_bf16 (or idle / fp32 MFMA).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int V> __device__ __forceinline__ v2f op(v2f x, v2f p, v2f c) {
    v2f d;
    if constexpr (V == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(p), "v"(c));
    if constexpr (V == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(x), "v"(p), "v"(c));
    if constexpr (V == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(x), "v"(p), "v"(c));
    if constexpr (V == 3) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "v"(p));
    if constexpr (V == 4) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(p));
    if constexpr (V == 5) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(p));
    if constexpr (V == 6) asm volatile("v_pk_fma_f32 %0, %2, %1, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(x), "v"(p), "v"(c));   // param as src0
    if constexpr (V == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(x), "v"(p), "v"(c));   // src2.hi -> both
    if constexpr (V == 8) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "v"(p));
    if constexpr (V == 9) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(x), "v"(p));   // d.lo = x.hi, d.hi = p.hi... (swap test)
    if constexpr (V == 10) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(x), "v"(p), "v"(c));  // lo<-p.hi, hi<-p.lo (swapped)
    return d;
}
template <int V> __device__ __forceinline__ v2f ref(v2f x, v2f p, v2f c) {
    if constexpr (V == 0) return v2f{fmaf(x.x, p.x, c.x), fmaf(x.y, p.y, c.y)};
    if constexpr (V == 1) return v2f{fmaf(x.x, p.x, c.x), fmaf(x.y, p.x, c.y)};
    if constexpr (V == 2) return v2f{fmaf(x.x, p.y, c.x), fmaf(x.y, p.y, c.y)};
    if constexpr (V == 3) return v2f{x.x * p.y, x.y * p.y};
    if constexpr (V == 4) return v2f{x.x - p.x, x.y - p.x};
    if constexpr (V == 6) return v2f{fmaf(p.y, x.x, c.x), fmaf(p.y, x.y, c.y)};
    if constexpr (V == 7) return v2f{fmaf(x.x, p.x, c.y), fmaf(x.y, p.y, c.y)};
    if constexpr (V == 8) return v2f{x.x + p.y, x.y + p.y};
    if constexpr (V == 9) return v2f{x.y, p.y};
    if constexpr (V == 10) return v2f{fmaf(x.x, p.y, c.x), fmaf(x.y, p.x, c.y)};
    return v2f{x.x * p.x, x.y * p.y};
}

template <int V, int PARTNER>
__global__ __launch_bounds__(512) void k(const float* in, unsigned long long* bad, unsigned* badlanes, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < 4) {
        unsigned long long nbad = 0;
        for (int it = 0; it < iters; ++it) {
            const float* q = in + ((it * 7 + blockIdx.x * 13 + wave) & 1023) * 64 * 6 + lane * 6;
            v2f x{q[0], q[1]}, p{q[2], q[3]}, c{q[4], q[5]};
            const v2f want = ref<V>(x, p, c);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const v2f got = op<V>(x, p, c);
                if (__float_as_uint(got.x) != __float_as_uint(want.x)) { ++nbad; atomicOr(&badlanes[lane], 1u); }
                if (__float_as_uint(got.y) != __float_as_uint(want.y)) { ++nbad; atomicOr(&badlanes[lane], 2u); }
            }
        }
        if (nbad) atomicAdd(bad, nbad);
    } else if (PARTNER) {
        f32x4 acc = {0, 0, 0, 0};
        u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        float fa = 1.0f + lane, fb = 0.5f;
        for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (PARTNER == 1) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc, 0, 0, 0);
            }
        }
        if (acc[0] == 12345.f) bad[1] = 1;
    }
}

template <int V, int PARTNER> void run(const float* din, unsigned long long* dbad, unsigned* dl, const char* name) {
    (void)hipMemset(dbad, 0, 16); (void)hipMemset(dl, 0, 256);
    hipLaunchKernelGGL((k<V, PARTNER>), dim3(512), dim3(512), 0, 0, din, dbad, dl, 2000);
    (void)hipDeviceSynchronize();
    unsigned long long b[2]; unsigned l[64];
    (void)hipMemcpy(b, dbad, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(l, dl, 256, hipMemcpyDeviceToHost);
    unsigned lo = 0, hi = 0; int first = -1, last = -1;
    for (int i = 0; i < 64; ++i) if (l[i]) { if (first < 0) first = i; last = i; lo |= l[i] & 1; hi |= l[i] & 2; }
    printf("%-52s partner %-9s mismatches %10llu of %.2e   lanes %d..%d  halves %s%s\n", name, PARTNER == 1 ? "bf16 MFMA" : PARTNER == 2 ? "fp32 MFMA" : "idle", b[0],
           512.0 * 4 * 64 * 2000 * 16 * 2, first, last, lo ? "lo " : "", hi ? "hi" : "");
}
int main() {
    const size_t n = 1024 * 64 * 6;
    float* h = (float*)malloc(n * 4);
    srand(5);
    for (size_t i = 0; i < n; ++i) h[i] = (float)rand() / 2147483648.f * 4.f - 2.f;
    float* din; unsigned long long* dbad; unsigned* dl;
    (void)hipMalloc(&din, n * 4); (void)hipMalloc(&dbad, 16); (void)hipMalloc(&dl, 256);
    (void)hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice);
#define ALL(V, NAME) run<V, 0>(din, dbad, dl, NAME); run<V, 2>(din, dbad, dl, NAME); run<V, 1>(din, dbad, dl, NAME);
    ALL(6, "v_pk_fma_f32 op_sel:[1,0,0] (src0.hi -> both)")
    ALL(7, "v_pk_fma_f32 op_sel:[0,0,1] (src2.hi -> both)")
    ALL(8, "v_pk_add_f32 op_sel:[0,1] (src1.hi -> both)")
    ALL(9, "v_pk_mov_b32 op_sel:[1,0] (lo <- src0.hi)")
    ALL(10, "v_pk_fma_f32 lo<-src1.hi, hi<-src1.lo")
    ALL(0, "v_pk_fma_f32 (no op_sel)")
    ALL(5, "v_pk_mul_f32 (no op_sel)")
    ALL(1, "v_pk_fma_f32 op_sel_hi:[1,0,1] (src1.lo -> both)")
    ALL(2, "v_pk_fma_f32 op_sel:[0,1,0] (src1.hi -> both)")
    ALL(3, "v_pk_mul_f32 op_sel:[0,1] (src1.hi -> both)")
    ALL(4, "v_pk_add_f32 op_sel_hi:[1,0] neg (src1.lo -> both)")
    return 0;
}
