// How exactly does v_mfma_f32_16x16x32_bf16 sum?  (a) 32 exact bf16 x bf16 products into C = 0; (b) the same into a
// large C; compared with the fp64 sum and with an fp32 RNE fma chain.  Build: hipcc --offload-arch=gfx950 -O2 -o x.bin x.hip
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
    // A: [16][32] row-major (i, k), B: [32][16] (k, j), C/D: [16][16]
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
    srand(1);
    std::vector<u16> A(512), B(512); std::vector<float> C(256), D(256);
    u16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    for (int mode = 0; mode < 4; ++mode) {
        double emax = 0, esum = 0, bias = 0, emax_f32 = 0; int n = 0;
        for (int rep = 0; rep < 200; ++rep) {
            for (auto& a : A) a = f2bf((float)rand() / RAND_MAX * (mode & 1 ? 1.f : 2.f) - (mode & 1 ? 0.f : 1.f));
            for (auto& b : B) b = f2bf((float)rand() / RAND_MAX * (mode & 1 ? 1.f : 2.f) - (mode & 1 ? 0.f : 1.f));
            for (auto& c : C) c = (mode & 2) ? 1000.f * ((float)rand() / RAND_MAX + 0.5f) : 0.f;
            hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
            hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
            hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                double ex = C[i * 16 + j]; float ch = C[i * 16 + j];
                for (int kk = 0; kk < 32; ++kk) { double p = (double)bf2f(A[i * 32 + kk]) * bf2f(B[kk * 16 + j]); ex += p; ch = fmaf(bf2f(A[i * 32 + kk]), bf2f(B[kk * 16 + j]), ch); }
                const double ulp = ldexp(1.0, ilogb(fabs(ex) > 1e-30 ? fabs(ex) : 1e-30) - 23);
                const double e = (D[i * 16 + j] - ex) / ulp;
                emax = fmax(emax, fabs(e)); esum += fabs(e); bias += e; ++n;
                emax_f32 = fmax(emax_f32, fabs((ch - ex) / ulp));
            }
        }
        printf("mode %d (%s operands, C %s): error in ulps of the result: max %.2f mean|e| %.3f bias %.3f | fp32 fma chain max %.2f\n", mode,
               mode & 1 ? "positive" : "signed", mode & 2 ? "~1000" : "0", emax, esum / n, bias / n, emax_f32);
    }
    return 0;
}
