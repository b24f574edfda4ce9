// Prototype: T1 E-step with one lane per data-row PAIR (softmax over the 16 components in registers, parameters
// broadcast from LDS) instead of lane = (component, row) with DPP softmax.  D = 8, K = 16.  Prints time per launch at N = 1e6
// and the max deviation from a direct fp64 evaluation on a few rows.
// hipcc --offload-arch=gfx950 -O3 -I vmp-for-svae_amd/csrc tools/ubench/estep_rowlane.hip -o tools/ubench/estep_rowlane.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "vmp_common.h"
using namespace vmp;
constexpr int D = 8, K = 16, TRI = 36, PACK = 48;

__global__ __launch_bounds__(256) void estep_rl(const float* __restrict__ x, const float* __restrict__ pack, float* __restrict__ r,
                                                long long N, long long rpw) {
    __shared__ __attribute__((aligned(16))) float pk[K * PACK];
    for (int i = threadIdx.x; i < K * PACK; i += blockDim.x) pk[i] = pack[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long lo = ((long long)blockIdx.x * nw + wave) * rpw;
    const long long hi = lo + rpw < N ? lo + rpw : N;
    for (long long base = lo; base < hi; base += 128) {
        const long long n0 = base + lane, n1 = base + 64 + lane;
        const bool v0 = n0 < hi, v1 = n1 < hi;
        v2f xv[D];
        {
            const float4* p0 = reinterpret_cast<const float4*>(x + (v0 ? n0 : lo) * D);
            const float4* p1 = reinterpret_cast<const float4*>(x + (v1 ? n1 : lo) * D);
            const float4 a0 = p0[0], a1 = p0[1], b0 = p1[0], b1 = p1[1];
            xv[0] = v2f{a0.x, b0.x}; xv[1] = v2f{a0.y, b0.y}; xv[2] = v2f{a0.z, b0.z}; xv[3] = v2f{a0.w, b0.w};
            xv[4] = v2f{a1.x, b1.x}; xv[5] = v2f{a1.y, b1.y}; xv[6] = v2f{a1.z, b1.z}; xv[7] = v2f{a1.w, b1.w};
        }
        v2f lg[K];
        v2f nxt[PACK / 2];
        auto load_prm = [&](int k, v2f (&o)[PACK / 2]) {
#pragma unroll
            for (int q = 0; q < PACK / 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(pk + k * PACK + 4 * q);
                o[2 * q] = v2f{t.x, t.y}; o[2 * q + 1] = v2f{t.z, t.w};
            }
        };
        load_prm(0, nxt);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            v2f prm[PACK / 2];
#pragma unroll
            for (int q = 0; q < PACK / 2; ++q) prm[q] = nxt[q];
            asm volatile("" ::: "memory");               // keeps the parameter reads of later components below this point
            if (k + 1 < K) load_prm(k + 1, nxt);
            v2f dv[D], y[D];
#pragma unroll
            for (int j = 0; j < D; ++j) dv[j] = pk_sub_b(xv[j], prm[j >> 1], j & 1);
#pragma unroll
            for (int i = 0; i < D; ++i) { const int e = D + i * (i + 1) / 2; y[i] = pk_mul_b(dv[0], prm[e >> 1], e & 1); }
#pragma unroll
            for (int j = 1; j < D; ++j)
#pragma unroll
                for (int i = j; i < D; ++i) { const int e = D + i * (i + 1) / 2 + j; y[i] = pk_fma_b(dv[j], prm[e >> 1], y[i], e & 1); }
            v2f q = y[0] * y[0], q1 = v2f{0.f, 0.f};
#pragma unroll
            for (int i = 1; i < D; ++i) { if (i & 1) q1 = __builtin_elementwise_fma(y[i], y[i], q1); else q = __builtin_elementwise_fma(y[i], y[i], q); }
            q += q1;
            lg[k] = pk_const_minus_scaled(q, prm[(D + TRI) >> 1]);
        }
        v2f mx = lg[0];
#pragma unroll
        for (int k = 1; k < K; ++k) mx = __builtin_elementwise_max(mx, lg[k]);
        v2f ss = v2f{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K; ++k) { lg[k] = v2f{__builtin_amdgcn_exp2f(lg[k].x - mx.x), __builtin_amdgcn_exp2f(lg[k].y - mx.y)}; ss += lg[k]; }
        const v2f inv = v2f{__builtin_amdgcn_rcpf(ss.x), __builtin_amdgcn_rcpf(ss.y)};
#pragma unroll
        for (int k = 0; k < K; ++k) lg[k] = lg[k] * inv;
        if (v0) {
            float4* o = reinterpret_cast<float4*>(r + n0 * K);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = make_float4(lg[4 * q].x, lg[4 * q + 1].x, lg[4 * q + 2].x, lg[4 * q + 3].x);
        }
        if (v1) {
            float4* o = reinterpret_cast<float4*>(r + n1 * K);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = make_float4(lg[4 * q].y, lg[4 * q + 1].y, lg[4 * q + 2].y, lg[4 * q + 3].y);
        }
    }
}

// ---- packed ops against a parameter held in an SGPR pair (scalar-loaded, wave-uniform) ----------------------
__device__ __forceinline__ v2f pk_fma_s(v2f x, v2f p, v2f acc, int h) {
    v2f d;
    if (h) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(x), "s"(p), "v"(acc));
    else   asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(x), "s"(p), "v"(acc));
    return d;
}
__device__ __forceinline__ v2f pk_mul_s(v2f x, v2f p, int h) {
    v2f d;
    if (h) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "s"(p));
    else   asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "s"(p));
    return d;
}
__device__ __forceinline__ v2f pk_sub_s(v2f x, v2f p, int h) {
    v2f d;
    if (h) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "s"(p));
    else   asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "s"(p));
    return d;
}
__device__ __forceinline__ v2f pk_cms_s(v2f q, v2f c) {       // c.lo - q * c.hi
    v2f d;
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(q), "s"(c));
    return d;
}

__global__ __launch_bounds__(256) void estep_rs(const float* __restrict__ x, const float* __restrict__ pack, float* __restrict__ r,
                                                long long N, long long rpw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long lo = ((long long)blockIdx.x * nw + wave) * rpw;
    const long long hi = lo + rpw < N ? lo + rpw : N;
    const v2f* __restrict__ pk2 = reinterpret_cast<const v2f*>(pack);
    for (long long base = lo; base < hi; base += 128) {
        const long long n0 = base + lane, n1 = base + 64 + lane;
        const bool v0 = n0 < hi, v1 = n1 < hi;
        v2f xv[D];
        {
            const float4* p0 = reinterpret_cast<const float4*>(x + (v0 ? n0 : lo) * D);
            const float4* p1 = reinterpret_cast<const float4*>(x + (v1 ? n1 : lo) * D);
            const float4 a0 = p0[0], a1 = p0[1], b0 = p1[0], b1 = p1[1];
            xv[0] = v2f{a0.x, b0.x}; xv[1] = v2f{a0.y, b0.y}; xv[2] = v2f{a0.z, b0.z}; xv[3] = v2f{a0.w, b0.w};
            xv[4] = v2f{a1.x, b1.x}; xv[5] = v2f{a1.y, b1.y}; xv[6] = v2f{a1.z, b1.z}; xv[7] = v2f{a1.w, b1.w};
        }
        v2f lg[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            v2f prm[PACK / 2];
#pragma unroll
            for (int q = 0; q < PACK / 2; ++q) prm[q] = pk2[k * (PACK / 2) + q];     // wave-uniform: scalar loads
            v2f dv[D], y[D];
#pragma unroll
            for (int j = 0; j < D; ++j) dv[j] = pk_sub_s(xv[j], prm[j >> 1], j & 1);
#pragma unroll
            for (int i = 0; i < D; ++i) { const int e = D + i * (i + 1) / 2; y[i] = pk_mul_s(dv[0], prm[e >> 1], e & 1); }
#pragma unroll
            for (int j = 1; j < D; ++j)
#pragma unroll
                for (int i = j; i < D; ++i) { const int e = D + i * (i + 1) / 2 + j; y[i] = pk_fma_s(dv[j], prm[e >> 1], y[i], e & 1); }
            v2f q = y[0] * y[0], q1 = v2f{0.f, 0.f};
#pragma unroll
            for (int i = 1; i < D; ++i) { if (i & 1) q1 = __builtin_elementwise_fma(y[i], y[i], q1); else q = __builtin_elementwise_fma(y[i], y[i], q); }
            q += q1;
            lg[k] = pk_cms_s(q, prm[(D + TRI) >> 1]);
            asm volatile("" ::: "memory");
        }
        v2f mx = lg[0];
#pragma unroll
        for (int k = 1; k < K; ++k) mx = __builtin_elementwise_max(mx, lg[k]);
        v2f ss = v2f{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K; ++k) { lg[k] = v2f{__builtin_amdgcn_exp2f(lg[k].x - mx.x), __builtin_amdgcn_exp2f(lg[k].y - mx.y)}; ss += lg[k]; }
        const v2f inv = v2f{__builtin_amdgcn_rcpf(ss.x), __builtin_amdgcn_rcpf(ss.y)};
#pragma unroll
        for (int k = 0; k < K; ++k) lg[k] = lg[k] * inv;
        if (v0) {
            float4* o = reinterpret_cast<float4*>(r + n0 * K);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = make_float4(lg[4 * q].x, lg[4 * q + 1].x, lg[4 * q + 2].x, lg[4 * q + 3].x);
        }
        if (v1) {
            float4* o = reinterpret_cast<float4*>(r + n1 * K);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = make_float4(lg[4 * q].y, lg[4 * q + 1].y, lg[4 * q + 2].y, lg[4 * q + 3].y);
        }
    }
}

int main() {
    const long long N = 1000000;
    std::vector<float> hx(N * D), hp(K * PACK, 0.f);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hx) v = rnd() * 10.f;
    for (int k = 0; k < K; ++k) {
        for (int j = 0; j < D; ++j) hp[k * PACK + j] = rnd() * 8.f;
        for (int i = 0; i < D; ++i) for (int j = 0; j <= i; ++j) hp[k * PACK + D + i * (i + 1) / 2 + j] = (i == j ? 0.6f : 0.1f * rnd());
        hp[k * PACK + D + TRI] = rnd();            // c (log2 domain)
        hp[k * PACK + D + TRI + 1] = 0.72f;        // h (log2 domain)
    }
    float *dx, *dp, *dr;
    (void)hipMalloc(&dx, N * D * 4); (void)hipMalloc(&dp, K * PACK * 4); (void)hipMalloc(&dr, N * K * 4);
    (void)hipMemcpy(dx, hx.data(), N * D * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dp, hp.data(), K * PACK * 4, hipMemcpyHostToDevice);
    for (int wpb : {4}) for (int blocks : {256, 512}) {
        const long long waves = (long long)blocks * wpb;
        long long rpw = (N + waves - 1) / waves; rpw = (rpw + 127) / 128 * 128;
        const int nb = (int)((N + rpw * wpb - 1) / (rpw * wpb));
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(estep_rl, dim3(nb), dim3(wpb * 64), 0, 0, dx, dp, dr, N, rpw);
        (void)hipEventRecord(a);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(estep_rl, dim3(nb), dim3(wpb * 64), 0, 0, dx, dp, dr, N, rpw);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("blocks %4d x %d waves (rpw %lld): %.1f us per launch\n", nb, wpb, rpw, ms * 1e3 / 20);
    }
    for (int blocks : {256, 512, 768}) {
        const int wpb = 4;
        const long long waves = (long long)blocks * wpb;
        long long rpw = (N + waves - 1) / waves; rpw = (rpw + 127) / 128 * 128;
        const int nb = (int)((N + rpw * wpb - 1) / (rpw * wpb));
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(estep_rs, dim3(nb), dim3(wpb * 64), 0, 0, dx, dp, dr, N, rpw);
        (void)hipEventRecord(a);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(estep_rs, dim3(nb), dim3(wpb * 64), 0, 0, dx, dp, dr, N, rpw);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("SGPR params: blocks %4d x %d waves (rpw %lld): %.1f us per launch\n", nb, wpb, rpw, ms * 1e3 / 20);
    }
    std::vector<float> hr(64 * K);
    (void)hipMemcpy(hr.data(), dr + (N - 64) * K, 64 * K * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int t = 0; t < 64; ++t) {
        const long long n = N - 64 + t;
        double l[K], mx = -1e300, sum = 0;
        for (int k = 0; k < K; ++k) {
            double q = 0;
            for (int i = 0; i < D; ++i) { double y = 0; for (int j = 0; j <= i; ++j) y += (double)hp[k * PACK + D + i * (i + 1) / 2 + j] * ((double)hx[n * D + j] - (double)hp[k * PACK + j]); q += y * y; }
            l[k] = (double)hp[k * PACK + D + TRI] - (double)hp[k * PACK + D + TRI + 1] * q; mx = fmax(mx, l[k]);
        }
        for (int k = 0; k < K; ++k) sum += exp2(l[k] - mx);
        for (int k = 0; k < K; ++k) worst = fmax(worst, fabs(exp2(l[k] - mx) / sum - (double)hr[t * K + k]));
    }
    printf("max |r - fp64| on the last 64 rows: %.2e\n", worst);
    return 0;
}
