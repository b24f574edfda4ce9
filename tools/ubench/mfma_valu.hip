// Does the fp32 MFMA (v_mfma_f32_16x16x4_f32) overlap with VALU work of a co-resident wave on the same SIMD?
// 512-thread blocks, 1 per CU: waves 0-3 (one per SIMD) run `role_a`, waves 4-7 `role_b`.
// roles: 0 idle, 1 fp32 MFMA chain (4 accumulators), 2 v_fma_f32 stream, 3 bf16 MFMA chain, 4 v_exp_f32 stream.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o tools/ubench/mfma_valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float run_role(int role, int iters, float seed) {
    if (role == 1) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        float a = seed, b = seed * 0.5f;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
        return c0[0] + c1[1] + c2[2] + c3[3];
    } else if (role == 2) {
        float a = seed, b = 1.0001f, c = 0.5f, d = 0.25f, e = 2.f, f = 3.f, g = 4.f, h = 5.f;
        for (int i = 0; i < iters * 8; ++i) {       // 8 v_fma per inner step; iters*8 steps -> 64*iters fma ~ same cycles as 4 MFMA x 32 cyc ... scaled below
            a = fmaf(a, b, c); d = fmaf(d, b, c); e = fmaf(e, b, c); f = fmaf(f, b, c);
            g = fmaf(g, b, c); h = fmaf(h, b, c); c = fmaf(c, b, a); b = fmaf(b, 0.9999f, 1e-9f);
        }
        return a + d + e + f + g + h + c + b;
    } else if (role == 3) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (short)(seed + j); b[j] = (short)(j * 3); }
        for (int i = 0; i < iters * 4; ++i) {       // 16x16x32 bf16: 8 passes... 4x as many to match duration roughly
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
        return c0[0] + c1[1] + c2[2] + c3[3];
    } else if (role == 4) {
        float a = seed * 1e-3f, d = a + 0.1f, e = a + 0.2f, f = a + 0.3f;
        for (int i = 0; i < iters * 4; ++i) {
            a = __builtin_amdgcn_exp2f(a) - 1.f; d = __builtin_amdgcn_exp2f(d) - 1.f;
            e = __builtin_amdgcn_exp2f(e) - 1.f; f = __builtin_amdgcn_exp2f(f) - 1.f;
        }
        return a + d + e + f;
    }
    return 0.f;
}

__global__ __launch_bounds__(512) void k(float* out, int role_a, int role_b, int iters) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? role_a : role_b;
    const float r = run_role(role, iters, (float)(threadIdx.x & 7) + 1.f);
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    const char* names[] = {"idle", "mfma_f32", "v_fma_f32", "mfma_bf16", "v_exp_f32"};
    int pairs[][2] = {{1, 0}, {2, 0}, {3, 0}, {4, 0}, {1, 1}, {2, 2}, {1, 2}, {2, 1}, {3, 2}, {1, 4}, {3, 4}, {3, 3}};
    for (auto& p : pairs) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, p[0], p[1], iters);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, p[0], p[1], iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("waves0-3 %-10s waves4-7 %-10s : %8.3f ms\n", names[p[0]], names[p[1]], ms);
    }
    return 0;
}
