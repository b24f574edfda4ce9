// How many independent VALU instructions of the SAME wave issue for free in the shadow of one v_mfma_f32_16x16x4_f32
// (32-cycle issue)?  One wave per SIMD; loop body = 4 x [ MFMA ; KV x v_fma_f32 ] on 4 accumulators.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_cover.hip -o tools/ubench/mfma_cover.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KV, int MODE>   // MODE 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_exp_f32 (transcendental), 3: v_rcp
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = (float)(threadIdx.x & 7) + 1.f, b = a * 0.5f;
    float v[8]; v2f p[8];
    for (int j = 0; j < 8; ++j) { v[j] = a + j; p[j] = v2f{a + j, b + j}; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            c[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[m], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < KV; ++j) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j & 7]) : "v"(b));
                else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[j & 7]) : "v"(p[(j + 1) & 7]));
                else if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
                else asm volatile("v_rcp_f32 %0, %0" : "+v"(v[j & 7]));
            }
        }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += v[j] + p[j][0] + p[j][1];
    out[blockIdx.x * 256 + threadIdx.x] = s + c[0][0] + c[1][1] + c[2][2] + c[3][3];
}

template <int KV, int MODE>
void run(float* d, const char* nm) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 20000;
    hipLaunchKernelGGL((k<KV, MODE>), dim3(256), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<KV, MODE>), dim3(256), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-12s x%2d per MFMA: %7.3f ms  = %6.1f cycles per MFMA slot @2.2GHz\n", nm, KV, ms, ms * 1e-3 * 2.2e9 / (iters * 4.0));
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 256 * 4);
    run<0, 0>(d, "none");
    run<2, 0>(d, "v_fma_f32"); run<4, 0>(d, "v_fma_f32"); run<6, 0>(d, "v_fma_f32"); run<7, 0>(d, "v_fma_f32"); run<8, 0>(d, "v_fma_f32"); run<12, 0>(d, "v_fma_f32");
    run<2, 1>(d, "v_pk_fma_f32"); run<4, 1>(d, "v_pk_fma_f32"); run<6, 1>(d, "v_pk_fma_f32"); run<8, 1>(d, "v_pk_fma_f32");
    run<1, 2>(d, "v_exp_f32"); run<2, 2>(d, "v_exp_f32"); run<4, 2>(d, "v_exp_f32");
    run<1, 3>(d, "v_rcp_f32"); run<2, 3>(d, "v_rcp_f32");
    return 0;
}
