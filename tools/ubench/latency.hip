// micro-benchmarks: empty kernel, dependent global-load chain, launch gaps.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void empty_k(int* p) { if (p && threadIdx.x == 12345) p[0] = 1; }
__global__ void chase_k(const int* __restrict__ next, int* out, int hops) {
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    for (int h = 0; h < hops; ++h) i = next[i];
    out[threadIdx.x + blockIdx.x * blockDim.x] = i;
}
__global__ void fma_k(float* out, int iters) {
    float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f, e = 2.f, f = 3.f, g = 4.f, h = 5.f;
    for (int i = 0; i < iters; ++i) { a = fmaf(a, b, c); d = fmaf(d, b, c); e = fmaf(e, b, c); f = fmaf(f, b, c);
                                      g = fmaf(g, b, c); h = fmaf(h, b, c); c = fmaf(c, b, a); b = fmaf(b, 0.9999f, 1e-9f); }
    out[threadIdx.x + blockIdx.x * blockDim.x] = a + d + e + f + g + h + c + b;
}
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void pkfma_k(float* out, int iters) {
    v2f a = {threadIdx.x * 1.f, 1.f}, b = {1.0001f, 1.0002f}, c = {0.5f, 0.25f}, d = a, e = b, f = c, g = a + b, h = b + c;
    for (int i = 0; i < iters; ++i) { a = __builtin_elementwise_fma(a, b, c); d = __builtin_elementwise_fma(d, b, c);
        e = __builtin_elementwise_fma(e, b, c); f = __builtin_elementwise_fma(f, b, c); g = __builtin_elementwise_fma(g, b, c);
        h = __builtin_elementwise_fma(h, b, c); c = __builtin_elementwise_fma(c, b, a); b = __builtin_elementwise_fma(b, b, a); }
    v2f s = a + d + e + f + g + h + c + b;
    out[threadIdx.x + blockIdx.x * blockDim.x] = s.x + s.y;
}
static float timeit(void (*launch)(), int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3f / reps;
}
int *d_next, *d_out; float* d_f; int g_hops, g_blocks, g_threads, g_iters;
int main() {
    const int n = 1 << 24;
    std::vector<int> nx(n);
    for (int i = 0; i < n; ++i) nx[i] = (int)(((long long)i * 1103515245LL + 12345) & (n - 1));
    hipMalloc(&d_next, n * 4); hipMalloc(&d_out, 1 << 22); hipMalloc(&d_f, 1 << 24);
    hipMemcpy(d_next, nx.data(), n * 4, hipMemcpyHostToDevice);
    printf("empty kernel back-to-back (1 block x 64):      %.2f us/launch\n", timeit([] { hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, 0, nullptr); }, 200));
    printf("empty kernel back-to-back (256 blocks x 1024): %.2f us/launch\n", timeit([] { hipLaunchKernelGGL(empty_k, dim3(256), dim3(1024), 0, 0, nullptr); }, 200));
    for (int hops : {0, 1, 2, 4, 8, 16, 32}) {
        g_hops = hops;
        printf("chase  1 block x 64, %2d dependent hops:  %.2f us/launch\n", hops, timeit([] { hipLaunchKernelGGL(chase_k, dim3(1), dim3(64), 0, 0, d_next, d_out, g_hops); }, 100));
    }
    for (int hops : {0, 8, 32}) {
        g_hops = hops;
        printf("chase 256 blocks x 256, %2d dependent hops: %.2f us/launch\n", hops, timeit([] { hipLaunchKernelGGL(chase_k, dim3(256), dim3(256), 0, 0, d_next, d_out, g_hops); }, 100));
    }
    for (int wpb : {4, 8, 12, 16}) {
        g_threads = wpb * 64; g_iters = 4096;
        float t1 = timeit([] { hipLaunchKernelGGL(fma_k, dim3(256), dim3(g_threads), 0, 0, d_f, g_iters); }, 20);
        float t2 = timeit([] { hipLaunchKernelGGL(pkfma_k, dim3(256), dim3(g_threads), 0, 0, d_f, g_iters); }, 20);
        // per SIMD: wpb/4 waves, each iters*8 instr
        double instr = (double)g_iters * 8 * (wpb / 4.0);
        printf("waves/SIMD %d: v_fma_f32 %.2f cycles/instr/SIMD (%.1f us)   v_pk_fma_f32 %.2f cycles/instr/SIMD (%.1f us)  [@2.4GHz]\n",
               wpb / 4, t1 * 1e-6 * 2.4e9 / instr, t1, t2 * 1e-6 * 2.4e9 / instr, t2);
    }
    return 0;
}
