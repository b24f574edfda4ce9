// Does v_mfma_f32_16x16x32_bf16 keep the low bits of its 32 products when they CANCEL?  Terms of magnitude ~2^3 whose sum is ~2^-3:
// error of the result against the exact sum, in units of the result's ulp and of the largest term's ulp.  (Decides whether the bias of
// y = W x' + b belongs into the SAME MFMA as the leading products: csrc/vmp_mix.hip, pass_xdl_kernel.)
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_cancel.exe mfma_cancel_numerics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
__global__ void k(const u16* A, const u16* B, const float* C, float* D) {
    const int l = threadIdx.x, i16 = l & 15, g = l >> 4;
    union { bf16x8 v; u16 s[8]; } a, b;
    for (int t = 0; t < 8; ++t) { a.s[t] = A[i16 * 32 + 8 * g + t]; b.s[t] = B[(8 * g + t) * 16 + i16]; }
    f32x4 c;
    for (int v = 0; v < 4; ++v) c[v] = C[(4 * g + v) * 16 + i16];
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[(4 * g + v) * 16 + i16] = c[v];
}
static u16 f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (u16)(u >> 16); }
static float bf2f(u16 h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    srand(3);
    std::vector<u16> A(512), B(512); std::vector<float> C(256, 0.f), D(256);
    u16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: 16 big cancelling pairs (+t, -t') + small terms; mode 1: same with the result carried in C; mode 2: products spanning 2^-12 .. 2^3
        double e_res = 0, e_big = 0; int n = 0;
        for (int rep = 0; rep < 300; ++rep) {
            for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 32; ++kk) {
                float v = (float)rand() / RAND_MAX * 2.f - 1.f;
                float scale = mode == 2 ? ldexpf(1.f, -(kk % 8) * 2) : (kk < 16 ? 3.f : 0.01f);
                A[i * 32 + kk] = f2bf(v * scale);
            }
            for (int kk = 0; kk < 32; ++kk) for (int j = 0; j < 16; ++j) {
                float v = (float)rand() / RAND_MAX * 2.f - 1.f;
                B[kk * 16 + j] = f2bf(v * 3.f);
            }
            // make the first 16 terms cancel pairwise up to ~1e-2: term 2q+1 = -(term 2q) (1 + small)
            if (mode < 2) for (int i = 0; i < 16; ++i) for (int q = 0; q < 8; ++q) A[i * 32 + 2 * q + 1] = f2bf(-bf2f(A[i * 32 + 2 * q]) * 1.004f);
            if (mode < 2) for (int q = 0; q < 8; ++q) for (int j = 0; j < 16; ++j) B[(2 * q + 1) * 16 + j] = B[(2 * q) * 16 + j];
            for (auto& c : C) c = mode == 1 ? ((float)rand() / RAND_MAX - 0.5f) : 0.f;
            hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
            hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
            hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                double ex = C[i * 16 + j], big = fabs(ex);
                for (int kk = 0; kk < 32; ++kk) { double t = (double)bf2f(A[i * 32 + kk]) * bf2f(B[kk * 16 + j]); ex += t; if (fabs(t) > big) big = fabs(t); }
                double err = fabs((double)D[i * 16 + j] - ex);
                int er, eb; frexp(ex, &er); frexp(big, &eb);
                double ulp_r = ldexp(1.0, er - 24), ulp_b = ldexp(1.0, eb - 24);
                if (fabs(ex) > 1e-3) { if (err / ulp_r > e_res) e_res = err / ulp_r; if (err / ulp_b > e_big) e_big = err / ulp_b; ++n; }
            }
        }
        printf("mode %d: max error = %.2f ulp(result) = %.3f ulp(largest term)   (%d sums)\n", mode, e_res, e_big, n);
    }
    return 0;
}
