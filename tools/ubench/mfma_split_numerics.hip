// Accuracy of sum_n w_n * phi_n over a 64-row tile computed the way the T1 moment GEMM does it: 3-term bf16 split of both
// operands, six products on v_mfma_f32_16x16x32_bf16 - (A) all six into ONE fp32 accumulator, (B) hh into one, the
// five small products into another, added in fp64 - against fp64 and against the fp32 MFMA (16x16x4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned cvt(float a, float b) { unsigned p; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(a), "v"(b)); return p; }
__device__ void split3(float a, float b, unsigned (&t)[3]) {
    t[0] = cvt(a, b);
    float ra = a - __uint_as_float(t[0] << 16), rb = b - __uint_as_float(t[0] & 0xffff0000u);
    t[1] = cvt(ra, rb);
    ra -= __uint_as_float(t[1] << 16); rb -= __uint_as_float(t[1] & 0xffff0000u);
    t[2] = cvt(ra, rb);
}
// W: [64 rows][16 comps], P: [64 rows][16 feats]; out: [3 variants][16][16]
__global__ void k(const float* W, const float* P, double* out) {
    const int l = threadIdx.x, i16 = l & 15, g = l >> 4;
    f32x4 a1 = {0, 0, 0, 0}, big = {0, 0, 0, 0}, small = {0, 0, 0, 0}, f32acc = {0, 0, 0, 0};
    for (int body = 0; body < 2; ++body) {
        unsigned As[3][4], Bs[3][4];
        for (int u = 0; u < 4; ++u) {
            const int r0 = body * 32 + 8 * u + g, r1 = r0 + 4;
            unsigned t[3];
            split3(W[r0 * 16 + i16], W[r1 * 16 + i16], t);
            for (int q = 0; q < 3; ++q) As[q][u] = t[q];
            split3(P[r0 * 16 + i16], P[r1 * 16 + i16], t);
            for (int q = 0; q < 3; ++q) Bs[q][u] = t[q];
            f32acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[r0 * 16 + i16], P[r0 * 16 + i16], f32acc, 0, 0, 0);
            f32acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[r1 * 16 + i16], P[r1 * 16 + i16], f32acc, 0, 0, 0);
        }
        bf16x8 a[3], b[3];
        for (int q = 0; q < 3; ++q) {
            a[q] = __builtin_bit_cast(bf16x8, u32x4{As[q][0], As[q][1], As[q][2], As[q][3]});
            b[q] = __builtin_bit_cast(bf16x8, u32x4{Bs[q][0], Bs[q][1], Bs[q][2], Bs[q][3]});
        }
        for (int ta = 0; ta < 3; ++ta)
            for (int tb = 0; tb + ta < 3; ++tb) {
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ta], b[tb], a1, 0, 0, 0);
                if (ta + tb == 0) big = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ta], b[tb], big, 0, 0, 0);
            }
        for (int s = 2; s >= 1; --s)               // smallest products first
            for (int ta = 0; ta <= s; ++ta)
                small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ta], b[s - ta], small, 0, 0, 0);
    }
    for (int v = 0; v < 4; ++v) {
        const int o = (4 * g + v) * 16 + i16;
        out[o] = a1[v]; out[256 + o] = (double)big[v] + (double)small[v]; out[512 + o] = f32acc[v];
    }
}
int main() {
    srand(3);
    std::vector<float> W(1024), P(1024); std::vector<double> O(768);
    float *dW, *dP; double* dO;
    (void)hipMalloc(&dW, 4096); (void)hipMalloc(&dP, 4096); (void)hipMalloc(&dO, 768 * 8);
    const char* names[3] = {"bf16x3, one accumulator      ", "bf16x3, big + small (fp64 add)", "fp32 MFMA 16x16x4             "};
    for (int mode = 0; mode < 2; ++mode) {
        double emax[3] = {0, 0, 0}, bias[3] = {0, 0, 0}, eabs[3] = {0, 0, 0}; int n = 0;
        for (int rep = 0; rep < 300; ++rep) {
            for (auto& w : W) { float u = (float)rand() / 2147483648.f; w = mode ? expf(-12.f * u) : u; }
            for (auto& p : P) { float u = (float)rand() / 2147483648.f; p = mode ? (u * 40.f - 20.f) * (u * 40.f - 20.f) : 1.f + 0.f * u; }
            (void)hipMemcpy(dW, W.data(), 4096, hipMemcpyHostToDevice); (void)hipMemcpy(dP, P.data(), 4096, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dW, dP, dO);
            (void)hipMemcpy(O.data(), dO, 768 * 8, hipMemcpyDeviceToHost);
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                double ex = 0;
                for (int r = 0; r < 64; ++r) ex += (double)W[r * 16 + i] * (double)P[r * 16 + j];
                for (int v = 0; v < 3; ++v) {
                    const double e = (O[v * 256 + i * 16 + j] - ex) / fabs(ex);
                    emax[v] = fmax(emax[v], fabs(e)); bias[v] += e; eabs[v] += fabs(e);
                }
                ++n;
            }
        }
        printf("== %s\n", mode ? "w = exp(-12 u) (wide dynamic range), phi = (40u - 20)^2" : "w uniform [0,1), phi = 1 (the N_k column)");
        for (int v = 0; v < 3; ++v) printf("   %s relative error: max %.2e  mean|e| %.2e  bias %+.2e\n", names[v], emax[v], eabs[v] / n, bias[v] / n);
    }
    return 0;
}
