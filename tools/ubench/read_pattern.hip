// What read bandwidth does this chip give a kernel that streams two 5.12 GB tensors (the T2 backward: x and dL/dx, 320-byte cells)?
//   seq   : every wave reads 1 KB contiguous per instruction, tiles of 20 KB back to back (the ceiling for a streaming kernel)
//   seg64 : the ring kernel's pattern - per instruction 16 cells x one 64-byte segment (4 lanes per cell), cells 320 B apart; the five
//           segments of a cell row in five consecutive instructions (sample pairs 0..4), both tensors
//   seg128: 8 lanes per cell, one 128-byte aligned line per cell and instruction (8 cells per instruction)
// hipcc --offload-arch=gfx950 -O3 read_pattern.hip -o read_pattern.exe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int CELL = 320, TILE = 64 * CELL;
__device__ float g_sink;
template <int MODE>
__global__ __launch_bounds__(512) void k(const char* __restrict__ a, const char* __restrict__ b, long long ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long long t = (long long)blockIdx.x * nw + wave; t < ntiles; t += (long long)gridDim.x * nw) {
        const char* ta = a + t * TILE;
        const char* tb = b + t * TILE;
        if (MODE == 0) {
#pragma unroll
            for (int it = 0; it < 20; ++it) {
                acc += *reinterpret_cast<const f32x4*>(ta + it * 1024 + lane * 16);
                acc += *reinterpret_cast<const f32x4*>(tb + it * 1024 + lane * 16);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int p = 0; p < 5; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) {                 // 4 instructions x 16 cells = the tile's 64 cells
                    const int cell = 16 * q + (lane >> 2);
                    acc += *reinterpret_cast<const f32x4*>(ta + cell * CELL + p * 64 + (lane & 3) * 16);
                    acc += *reinterpret_cast<const f32x4*>(tb + cell * CELL + p * 64 + (lane & 3) * 16);
                }
        } else {
            // 20 KB tile = 160 lines of 128 B; instruction i reads lines 8 i .. 8 i + 7 but in the order "line j of every 2.5-line cell
            // row": here simply 8 lanes per line, lines strided by 5 (a cell pair = 5 lines), so one instruction touches 8 cell pairs
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int pair = 8 * q + (lane >> 3);    // 32 cell pairs per tile
                    acc += *reinterpret_cast<const f32x4*>(ta + pair * 640 + j * 128 + (lane & 7) * 16);
                    acc += *reinterpret_cast<const f32x4*>(tb + pair * 640 + j * 128 + (lane & 7) * 16);
                }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) g_sink = acc.x;
}
// The same patterns with at most 16 KB per wave in flight (the ring kernel's LDS budget: 8 waves x 16 KB per CU): a group of 16 load
// instructions (8 per tensor) is issued, then consumed (s_waitcnt vmcnt(0)) before the next group goes out; `work` dependent FMAs per
// group stand in for the sample arithmetic.
template <int MODE>
__global__ __launch_bounds__(512) void k16(const char* __restrict__ a, const char* __restrict__ b, long long ntiles, int work) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float w = 1.0f;
    for (long long t = (long long)blockIdx.x * nw + wave; t < ntiles; t += (long long)gridDim.x * nw) {
        const char* ta = a + t * TILE;
        const char* tb = b + t * TILE;
        // 20 KB per tensor and tile = 2.5 groups of 8 KB: groups of 8 instructions, the last group of a tile has 4
#pragma unroll
        for (int grp = 0; grp < 3; ++grp) {
            f32x4 va[8], vb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int it = grp * 8 + i;                   // instruction index 0..19 of the tile
                if (it >= 20) { va[i] = f32x4{0.f, 0.f, 0.f, 0.f}; vb[i] = va[i]; continue; }
                int off;
                if (MODE == 0) off = it * 1024 + lane * 16;
                else if (MODE == 1) { const int p = it >> 2, q = it & 3; off = (16 * q + (lane >> 2)) * CELL + p * 64 + (lane & 3) * 16; }
                else { const int j = it >> 2, q = it & 3; off = (8 * q + (lane >> 3)) * 640 + j * 128 + (lane & 7) * 16; }
                va[i] = *reinterpret_cast<const f32x4*>(ta + off);
                vb[i] = *reinterpret_cast<const f32x4*>(tb + off);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { acc += va[i]; acc += vb[i]; }
            for (int i = 0; i < work; ++i) w = fmaf(w, 1.0000001f, acc.x);
            asm volatile("" : "+v"(w));
        }
    }
    if (acc.x + acc.y + acc.z + acc.w + w == 12345.678f) g_sink = acc.x;
}
// The same three patterns through the LDS-DMA path the ring kernel uses (global_load_lds_dwordx4: 16 bytes per lane straight into LDS),
// 16 KB per wave in flight: a group of 16 instructions, s_waitcnt vmcnt(0), next group.
template <int MODE>
__global__ __launch_bounds__(512) void kdma(const char* __restrict__ a, const char* __restrict__ b, long long ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float* ring = lds + wave * 4096;
    for (long long t = (long long)blockIdx.x * nw + wave; t < ntiles; t += (long long)gridDim.x * nw) {
        const char* ta = a + t * TILE;
        const char* tb = b + t * TILE;
#pragma unroll
        for (int grp = 0; grp < 3; ++grp) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int it = grp * 8 + i;
                if (it >= 20) continue;
                int off;
                if (MODE == 0) off = it * 1024 + lane * 16;
                else if (MODE == 1) { const int p = it >> 2, q = it & 3; off = (16 * q + (lane >> 2)) * CELL + p * 64 + (lane & 3) * 16; }
                else { const int j = it >> 2, q = it & 3; off = (8 * q + (lane >> 3)) * 640 + j * 128 + (lane & 7) * 16; }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ta + off),
                                                 (__attribute__((address_space(3))) void*)(ring + i * 256), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + off),
                                                 (__attribute__((address_space(3))) void*)(ring + 2048 + i * 256), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (ring[lane] == 12345.678f) g_sink = ring[lane];
}
int main() {
    const long long ntiles = 250000;
    char *a, *b;
    hipMalloc(&a, ntiles * TILE); hipMalloc(&b, ntiles * TILE);
    hipMemset(a, 0, ntiles * TILE); hipMemset(b, 0, ntiles * TILE);
    hipFuncSetAttribute((const void*)kdma<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
    hipFuncSetAttribute((const void*)kdma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
    hipFuncSetAttribute((const void*)kdma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"seq   ", "seg64 ", "seg128"};
    for (int blocks : {256, 512, 1024}) for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, a, b, ntiles);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, a, b, ntiles);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, a, b, ntiles);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
        }
        printf("%4d blocks x 8 waves  %s : %.3f ms  %.0f GB/s\n", blocks, names[mode], best, 2.0 * ntiles * TILE / (best * 1e-3) / 1e9);
    }
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(kdma<0>, dim3(256), dim3(512), 8 * 16384, 0, a, b, ntiles);
            if (mode == 1) hipLaunchKernelGGL(kdma<1>, dim3(256), dim3(512), 8 * 16384, 0, a, b, ntiles);
            if (mode == 2) hipLaunchKernelGGL(kdma<2>, dim3(256), dim3(512), 8 * 16384, 0, a, b, ntiles);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
        }
        printf("LDS-DMA (global_load_lds_dwordx4), 16 KB per wave in flight  %s : %.3f ms  %.0f GB/s\n", names[mode], best, 2.0 * ntiles * TILE / (best * 1e-3) / 1e9);
    }
    for (int work : {0, 400, 800}) for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k16<0>, dim3(256), dim3(512), 0, 0, a, b, ntiles, work);
            if (mode == 1) hipLaunchKernelGGL(k16<1>, dim3(256), dim3(512), 0, 0, a, b, ntiles, work);
            if (mode == 2) hipLaunchKernelGGL(k16<2>, dim3(256), dim3(512), 0, 0, a, b, ntiles, work);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
        }
        printf("16 KB per wave in flight, %3d dependent FMAs per group  %s : %.3f ms  %.0f GB/s\n", work, names[mode], best, 2.0 * ntiles * TILE / (best * 1e-3) / 1e9);
    }
    return 0;
}
