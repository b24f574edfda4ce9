// Cost model for mixing v_mfma_f32_16x16x32_bf16 (XDL pipe) with VALU work on gfx950.
// Loop body = 4 x [ MFMA (4 independent accumulators) ; KV independent VALU instructions ], run with 1 and with 2 waves
// per SIMD (256- / 512-thread blocks, one block per CU).  Prints SIMD cycles per MFMA slot, i.e. per [MFMA + KV VALU]
// of ONE wave divided by the waves per SIMD... reported as cycles per slot per SIMD (time * clock / slots issued on the SIMD).
// MFMA kind: 0 = bf16 16x16x32 (XDL), 1 = fp32 16x16x4.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/xdl_overlap.hip -o tools/ubench/xdl_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KV, int KIND, int THREADS, int NM, int VM = 0>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters) {
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = (float)(threadIdx.x & 7) + 1.f, b = a * 0.5f;
    bf16x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (__bf16)(a + j); bv[j] = (__bf16)(b - j); }
    float v[8];
    v2f pv[8];
    for (int j = 0; j < 8; ++j) { v[j] = a + j; pv[j] = v2f{a + j, b + j}; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m < NM) {
                // asm with VGPR accumulators: the builtin lets the compiler park them in AGPRs and shuffle them around the loop
                if (KIND == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[m]) : "v"(av), "v"(bv));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[m]) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int j = 0; j < KV; ++j) {
                if (VM == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j & 7]) : "v"(b));
                else if (VM == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pv[j & 7]) : "v"(pv[(j + 1) & 7]));
                else if (VM == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[j & 7]));
                else if (VM == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
                else if (VM == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(b));
                else if (VM == 5) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[j & 7]));
                else if (VM == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pv[j & 7]) : "v"(pv[(j + 1) & 7]));
                else if (VM == 7) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(v[j & 7]) : "v"(b));
            }
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float s = 0;
    for (int j = 0; j < 8; ++j) s += v[j] + pv[j][0] + pv[j][1];
    out[blockIdx.x * THREADS + threadIdx.x] = s + c[0][0] + c[1][1] + c[2][2] + c[3][3];
}

template <int KV, int KIND, int THREADS, int NM, int VM = 0>
void run(float* d) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 20000;
    hipLaunchKernelGGL((k<KV, KIND, THREADS, NM, VM>), dim3(256), dim3(THREADS), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<KV, KIND, THREADS, NM, VM>), dim3(256), dim3(THREADS), 0, 0, d, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const int wps = THREADS / 256;
    static const char* vm[] = {"v_fma_f32", "v_pk_fma_f32", "v_rcp_f32", "v_exp_f32", "v_cvt_pk_bf16", "v_and_b32", "v_pk_mul_f32", "v_med3_f32"};
    printf("%-14s ", vm[VM]);
    printf("%s x%d/slot  KV=%2d  waves/SIMD %d : %7.3f ms = %6.1f cycles per wave-slot, %6.1f SIMD cycles per slot @2.4GHz\n",
           KIND ? "fp32 16x16x4 " : "bf16 16x16x32", NM ? 1 : 0, KV, wps, ms, ms * 1e-3 * 2.4e9 / (iters * 4.0), ms * 1e-3 * 2.4e9 / (iters * 4.0 * wps));
}
template <int KIND, int THREADS>
void sweep(float* d) {
    run<0, KIND, THREADS, 4>(d); run<1, KIND, THREADS, 4>(d); run<2, KIND, THREADS, 4>(d); run<3, KIND, THREADS, 4>(d);
    run<4, KIND, THREADS, 4>(d); run<6, KIND, THREADS, 4>(d); run<8, KIND, THREADS, 4>(d); run<12, KIND, THREADS, 4>(d); run<16, KIND, THREADS, 4>(d);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
    printf("# VALU only (no MFMA)\n");
    run<4, 0, 256, 0>(d); run<8, 0, 256, 0>(d); run<4, 0, 512, 0>(d); run<8, 0, 512, 0>(d); run<8, 0, 1024, 0>(d);
    printf("# bf16 XDL, 1 wave/SIMD\n"); sweep<0, 256>(d);
    printf("# bf16 XDL, 2 waves/SIMD\n"); sweep<0, 512>(d);
    printf("# bf16 XDL, 4 waves/SIMD\n"); run<0, 0, 1024, 4>(d); run<4, 0, 1024, 4>(d); run<8, 0, 1024, 4>(d);
    printf("# VALU kinds, no MFMA, 8 per slot: 2 and 4 waves/SIMD\n");
    run<8, 0, 512, 0, 0>(d); run<8, 0, 512, 0, 1>(d); run<8, 0, 512, 0, 2>(d); run<8, 0, 512, 0, 3>(d); run<8, 0, 512, 0, 4>(d); run<8, 0, 512, 0, 5>(d); run<8, 0, 512, 0, 6>(d); run<8, 0, 512, 0, 7>(d);
    run<8, 0, 1024, 0, 0>(d); run<8, 0, 1024, 0, 1>(d); run<8, 0, 1024, 0, 2>(d); run<8, 0, 1024, 0, 4>(d); run<8, 0, 1024, 0, 5>(d);
    printf("# VALU kinds beside 1 bf16 MFMA per 8, 2 waves/SIMD\n");
    run<8, 0, 512, 4, 1>(d); run<8, 0, 512, 4, 2>(d); run<8, 0, 512, 4, 4>(d); run<8, 0, 512, 4, 5>(d);
    printf("# fp32 MFMA, 2 waves/SIMD\n"); run<0, 1, 512, 4>(d); run<4, 1, 512, 4>(d); run<8, 1, 512, 4>(d);
    return 0;
}
