// Does the ORDER in which a tile's 128-byte lines are written matter?  (round 6: the T2 forward kernel writes 5.1 GB of samples; with its
// stores it runs 1.40 ms, without them 1.05-1.10 ms, while a sequential fill of the same bytes takes 0.75 ms.)
// Layout as the forward kernel's: a tile = 64 cells of 320 B (20 KB, contiguous), a wave owns tiles t, t + stride, ..; 256 blocks x 8 waves.
//   seq   : the tile's 20 KB as 20 back-to-back 1 KB store instructions, in address order
//   pst2  : the two-pair staging form's order - per couple of cells (640 B = lines 0..4): flush 1 writes lines {0, 3}, flush 2 {1, 4},
//           flush 3 {2}; `gap` idle cycles between flushes (the sample arithmetic in the real kernel)
//   pst4  : flush A writes lines {0, 1, 3, 4} (256 contiguous bytes per cell), flush B {2}
// hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern.exe ; ./store_pattern.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int CELL = 320, TILE = 64 * CELL, LINES_PER_COUPLE = 5;

__device__ float g_sink;
// gap > 0: `cycles` / 4 dependent-free packed-FMA instructions (VALU-busy, as the real kernel's sample arithmetic); gap < 0: idle sleeps
__device__ __forceinline__ void idle(int cycles) {
    if (cycles < 0) { for (int c = 0; c < -cycles; c += 64 * 16) __builtin_amdgcn_s_sleep(16); return; }
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f a = {1.f, 2.f}, b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f}, d = a, e = b, f = c, g = a + b, h = b + c;
    for (int i = 0; i < cycles / 32; ++i) {
        a = __builtin_elementwise_fma(a, b, c); d = __builtin_elementwise_fma(d, b, c); e = __builtin_elementwise_fma(e, b, c); f = __builtin_elementwise_fma(f, b, c);
        g = __builtin_elementwise_fma(g, b, c); h = __builtin_elementwise_fma(h, b, c); c = __builtin_elementwise_fma(c, b, a); b = __builtin_elementwise_fma(b, b, a);
    }
    v2f s = a + d + e + f + g + h + c + b;
    if (s.x + s.y == 12345.678f) g_sink = s.x;
}
// one store instruction: lane l writes 16 B of line `line` (128 B = 8 lanes) of couple (8 * grp + (l >> 3)) ...  64 lanes = 8 lines
__device__ __forceinline__ void store_lines(char* tile, int lane, int first_couple, int line, f32x4 v) {
    const int couple = first_couple + (lane >> 3);
    *reinterpret_cast<f32x4*>(tile + (size_t)couple * (2 * CELL) + line * 128 + (lane & 7) * 16) = v;
}
template <int MODE>
__global__ __launch_bounds__(512) void k(char* buf, long long ntiles, int gap, unsigned long long* clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    for (long long t = (long long)blockIdx.x * nw + wave; t < ntiles; t += (long long)gridDim.x * nw) {
        char* tile = buf + t * TILE;
        if (MODE == 0) {
            idle(3 * gap);
#pragma unroll
            for (int it = 0; it < 20; ++it) *reinterpret_cast<f32x4*>(tile + it * 1024 + lane * 16) = v;
        } else if (MODE == 1) {
            const int fl[3][2] = {{0, 3}, {1, 4}, {2, -1}};
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                idle(gap);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (fl[f][j] >= 0)
#pragma unroll
                        for (int g = 0; g < 4; ++g) store_lines(tile, lane, 8 * g, fl[f][j], v);     // 32 couples = 4 instructions per line index
            }
        } else if (MODE == 3) {
            idle(3 * gap);                                     // arithmetic only: no store
        } else if (MODE == 4) {
            // the same 20 store instructions, one after each twentieth of the arithmetic (never more than one store queued per wave at a time)
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f a = {1.f, 2.f}, b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f}, d = a, e = b, f = c, g = a + b, h = b + c;
            const int inner = (3 * gap / 32) / 20;
            for (int it = 0; it < 20; ++it) {
                for (int i = 0; i < inner; ++i) {
                    a = __builtin_elementwise_fma(a, b, c); d = __builtin_elementwise_fma(d, b, c); e = __builtin_elementwise_fma(e, b, c); f = __builtin_elementwise_fma(f, b, c);
                    g = __builtin_elementwise_fma(g, b, c); h = __builtin_elementwise_fma(h, b, c); c = __builtin_elementwise_fma(c, b, a); b = __builtin_elementwise_fma(b, b, a);
                }
                *reinterpret_cast<f32x4*>(tile + it * 1024 + lane * 16) = v;
            }
            v2f s = a + d + e + f + g + h + c + b;
            if (s.x + s.y == 12345.678f) g_sink = s.x;
        } else if (MODE == 5) {
            // bursts of four stores after each fifth of the arithmetic
            for (int it = 0; it < 5; ++it) {
                idle(3 * gap / 5);
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(tile + (4 * it + j) * 1024 + lane * 16) = v;
            }
        } else {
            idle(2 * gap + gap / 2);
            const int la[4] = {0, 1, 3, 4};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) store_lines(tile, lane, 8 * g, la[j], v);
            idle(gap / 2);
#pragma unroll
            for (int g = 0; g < 4; ++g) store_lines(tile, lane, 8 * g, 2, v);
        }
    }
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = clock64() - c0; clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }   // shader cycles, 100 MHz ticks
}
int main() {
    const long long ntiles = 250000;                           // 1e6 rows x 16 cells / 64 = 5.12 GB
    char* buf; hipMalloc(&buf, ntiles * TILE);
    unsigned long long* clk; hipMallocManaged(&clk, 512 * sizeof(unsigned long long));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char* names[6] = {"seq ", "pst2", "pst4", "none", "1/20 ", "4/5  "};
    for (int gap : {0, -4000, 3000, 4000}) {
        for (int mode = 0; mode < 6; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, buf, ntiles, gap, clk);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, buf, ntiles, gap, clk);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, buf, ntiles, gap, clk);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, buf, ntiles, gap, clk);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, buf, ntiles, gap, clk);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, buf, ntiles, gap, clk);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (rep && ms < best) best = ms;
            }
            double cyc = 0, tick = 0; for (int bl = 0; bl < 256; ++bl) { cyc += clk[2 * bl]; tick += clk[2 * bl + 1]; }
            printf("gap %5d cycles/flush (%s)  %s : %.3f ms  %.0f GB/s   shader clock %.0f MHz\n", gap, gap < 0 ? "sleep" : "packed FMAs", names[mode], best, ntiles * (double)TILE / (best * 1e-3) / 1e9, cyc / tick * 100.0);
        }
    }
    return 0;
}
