// T1: pure mixture VMP (GMM: reference models/gmm.py:25-269; SMM: models/smm.py:25-245) for gfx950.
//
// One streaming "pass" kernel.  A wave processes tiles of 64 data rows in two bodies of 32 rows = 4 groups of 8 rows.
// Lane l = (i16 = l & 15, kk = l >> 4) owns mixture component k = i16 (+16 per extra component tile) and the data
// rows n8+kk and n8+4+kk of a group:
//   E-part  (packed fp32 VALU): q = ||W_k (x_n - m_k)||^2 with W_k, m_k resident in the lane's VGPRs (loaded once
//            per kernel), the two rows of the lane packed into v_pk_fma_f32; softmax over k = all-reduce over the
//            16 lanes of a DPP row (v_*_dpp row_ror), no scalar loads in the loop;
//   M-part  (matrix pipe): sum_n w_nk * [1 | x_n | x_n x_n^T] as a GEMM with the data row as inner index.  fp32 MFMA
//            runs on the VALU's issue slots on gfx950 (round-1 measurement), so the operands are split into three
//            bf16 terms each (v = h + m + l, 8 bits per term, |residual| < 2^-26 |v|) and the six products of order
//            <= 2 go to v_mfma_f32_16x16x32_bf16: exact products, fp32 accumulation - the accuracy of the fp32 chain
//            it replaces (tools/ubench/mfma_split_numerics.hip) on the otherwise idle XDL pipe.  The lane's weights
//            and feature products of a group fill slot t = 2u + h of its 8 k-slots (the same slot <-> row map for
//            A and B); one burst of 18 MFMAs per body.  The h h products and the five corrections accumulate
//            separately (fp32), flushed to fp64 registers after every tile.
// r_nk leaves the E-part in a layout whose 64 lanes cover 4 consecutive rows x 16 components = contiguous memory,
// so it is stored coalesced without a transpose.  The x tile is staged once per tile in LDS as [d][position] with the
// two rows of a lane adjacent (one ds_read_b64 each, see ppos below).  Per-block fp64 partials go to the workspace
// and are reduced in a fixed order by the finalize kernel (deterministic, no atomics).
// See vmp_common.h for the packed-fp32 op_sel erratum the bf16 MFMAs expose.
#include "vmp_common.h"
#include <stdlib.h>
#include <type_traits>

using namespace vmp;

namespace {

constexpr int TR = 64;        // data rows per wave tile
constexpr int LS = 68;        // LDS stride (floats) between value-rows of the x image (68 = 4 mod 64: see below)
constexpr int MAX_NW = 8;     // waves per block (K > 16)
constexpr int MAX_NW1 = 8;    // waves per block when K <= 16 (12 waves = 3 per SIMD measured no faster and caps the VGPRs at 168)
constexpr int max_nw(int KT) { return KT == 1 ? MAX_NW1 : MAX_NW; }
constexpr int MAX_BLOCKS = 1024;   // upper bound (workspace sizing); the plan uses tuned_blocks

struct PassArgs {
    const float* x;
    const float* r_in;
    const float* u_in;
    const uint8_t* mask;
    const float* pivot;   // D floats subtracted from x before anything is formed (NULL: none)
    const float* pack;
    float* r_out;
    float* u_out;
    float* logr_out;
    double* partials;
    long long N;
    long long rpw;        // rows per wave (multiple of 8): wave g owns rows [g*rpw, min(N, (g+1)*rpw))
    long long rpw_b;      // != rpw: the first half of a block's waves own rpw rows each, the second half rpw_b (see make_plan)
    int K;
    int vec_ok;       // x pointer 16-byte aligned (vector row loads allowed)
    int par_reduce;   // LDS holds one fp64 slab per wave: reduce the waves in one parallel step
#ifdef VMP_DEBUG_TS
    long long* dbg_t; // exploration builds only (make EXTRA=-DVMP_DEBUG_TS): timestamps of blocks 0 and 100
#endif
};
// In-kernel time stamps (tools/pass_ts.py) exist only in -DVMP_DEBUG_TS builds: the shipped library has no debug
// exports and no process-global state.
#ifdef VMP_DEBUG_TS
#define PASS_TS(i) do { if (a.dbg_t && (blockIdx.x == 0 || blockIdx.x == 100) && (threadIdx.x & 63) == 0) { a.dbg_t[(blockIdx.x ? 64 : 0) + (threadIdx.x >> 6) * 8 + (i)] = clock64(); \
    if ((i) == 0 && threadIdx.x == 0) a.dbg_t[(blockIdx.x ? 64 : 0) + 7] = wall_clock64();          /* 100 MHz, comparable ACROSS kernels: launch gaps */ \
    if ((i) == 5 && threadIdx.x == 64) a.dbg_t[(blockIdx.x ? 64 : 0) + 8 + 7] = wall_clock64(); } } while (0)
#else
#define PASS_TS(i) do { } while (0)
#endif

template <int D>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&o)[D], bool vec) {
    if constexpr (D % 4 == 0) {
        if (vec) {
#pragma unroll
            for (int j = 0; j < D / 4; ++j) {
                float4 v = reinterpret_cast<const float4*>(p)[j];
                o[4 * j] = v.x; o[4 * j + 1] = v.y; o[4 * j + 2] = v.z; o[4 * j + 3] = v.w;
            }
            return;
        }
    } else if constexpr (D % 2 == 0) {
        if (vec) {
#pragma unroll
            for (int j = 0; j < D / 2; ++j) {
                float2 v = reinterpret_cast<const float2*>(p)[j];
                o[2 * j] = v.x; o[2 * j + 1] = v.y;
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) o[j] = p[j];
}

// all-reduce over the 16 lanes of a DPP row (= the 16 components of one data row) by row rotations.
// Hand-written DPP: one instruction per butterfly step.  The two wait states a DPP read needs after a VALU
// write of the same VGPR are explicit (hipcc's hazard recogniser does not look inside asm).
#define VMP_DPP2(OP, CTRL)                                                   \
    "v_" OP "_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"  \
    "v_" OP "_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"  \
    "s_nop 0\n\t"
// two independent reductions interleaved: the partner's instruction is one of the two wait states
// (the first step reads the inputs and writes fresh registers: no copies to keep the un-reduced values alive)
#define VMP_DPP2_FIRST(OP, CTRL)                                             \
    "v_" OP "_f32_dpp %0, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"  \
    "v_" OP "_f32_dpp %1, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"  \
    "s_nop 0\n\t"
__device__ __forceinline__ v2f row16_max2(v2f v) {
    float a, b;
    asm("s_nop 1\n\t" VMP_DPP2_FIRST("max", "row_ror:8") VMP_DPP2("max", "row_ror:4") VMP_DPP2("max", "row_ror:2")
        VMP_DPP2("max", "row_ror:1")
        : "=&v"(a), "=&v"(b) : "v"(v.x), "v"(v.y));
    return v2f{a, b};
}
__device__ __forceinline__ v2f row16_sum2(v2f v) {
    float a, b;
    asm("s_nop 1\n\t" VMP_DPP2_FIRST("add", "row_ror:8") VMP_DPP2("add", "row_ror:4") VMP_DPP2("add", "row_ror:2")
        VMP_DPP2("add", "row_ror:1")
        : "=&v"(a), "=&v"(b) : "v"(v.x), "v"(v.y));
    return v2f{a, b};
}

constexpr int MOM_TERMS = 3;
#ifndef VMP_MOM_FLUSH
#define VMP_MOM_FLUSH 2       // tiles of 64 rows between two fp32 -> fp64 flushes of the moment accumulators (both pass kernels)
#endif

__device__ __forceinline__ double readlane_d(double v, int src_lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}

// Block reduction of the fp64 moment accumulators (waves summed in a fixed order) and the per-block partial: shared by
// the two pass kernels below.
template <int D, int KT, int FLAV>
__device__ __forceinline__ void pass_epilogue(const PassArgs& a, float* smem, const double (&dacc)[KT][Geo<D>::FT][4],
                                              const double (&dn)[KT][4], int lane, int wave, int nw) {
    using G = Geo<D>;
    constexpr int FT = G::FT;
    constexpr bool SMM = (FLAV == VMP_SMM);
    const int K = a.K;
    {
        // ---- block reduction in fp64 (waves summed in a fixed order), then one partial per block
        __syncthreads();
        PASS_TS(3);
        double* sc = reinterpret_cast<double*>(smem);          // [KT][FT+1][4][64]
        constexpr int SLAB = KT * (FT + 1) * 4 * WAVE;
        if (a.par_reduce) {
            // every wave drops its accumulators into its own slab, then each thread sums the nw slabs of its
            // elements: one LDS round trip instead of nw dependent read-modify-write rounds
            double* mine = sc + wave * SLAB;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int ft = 0; ft < FT; ++ft) mine[((kt * (FT + 1) + ft) * 4 + c) * WAVE + lane] = dacc[kt][ft][c];
                    mine[((kt * (FT + 1) + FT) * 4 + c) * WAVE + lane] = SMM ? dn[kt][c] : dacc[kt][0][c];
                }
            __syncthreads();
            // all slab values of a thread's elements are requested before the first addition (a rolled loop of dependent
            // read-add steps cost ~1.6 k cycles here); the additions keep the wave order
            constexpr int EPT = 2, NWB = max_nw(KT);               // elements per thread and round; bound on nw
            for (int e0 = threadIdx.x; e0 < SLAB; e0 += EPT * blockDim.x) {
                double vals[EPT][NWB];
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int e = e0 + q * blockDim.x;
#pragma unroll
                    for (int w = 0; w < NWB; ++w) vals[q][w] = (e < SLAB && w < nw) ? sc[w * SLAB + e] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int e = e0 + q * blockDim.x;
                    double t2 = vals[q][0];
#pragma unroll
                    for (int w = 1; w < NWB; ++w) t2 += vals[q][w];
                    if (e < SLAB) sc[e] = t2;
                }
            }
            __syncthreads();
        } else {
            for (int w = 0; w < nw; ++w) {
                if (wave == w) {
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
#pragma unroll
                            for (int ft = 0; ft < FT; ++ft) {
                                const int idx = ((kt * (FT + 1) + ft) * 4 + c) * WAVE + lane;
                                sc[idx] = (w == 0 ? 0.0 : sc[idx]) + dacc[kt][ft][c];
                            }
                            const int idn = ((kt * (FT + 1) + FT) * 4 + c) * WAVE + lane;
                            sc[idn] = (w == 0 ? 0.0 : sc[idn]) + (SMM ? dn[kt][c] : dacc[kt][0][c]);
                        }
                }
                __syncthreads();
            }
        }
        PASS_TS(4);
        // partials[k][block][PX]: one component's rows of all blocks are contiguous (what a finalize block streams);
        // slot PF of every row = this block's sum_k N_k (the Dirichlet normaliser needs the total count)
        constexpr int PX = G::PF + 1;
        double* out = a.partials;
        for (int e = threadIdx.x; e < KT * (FT + 1) * 4 * WAVE; e += blockDim.x) {
            const int l = e & 63, c = (e >> 6) & 3, tf = (e >> 8) % (FT + 1), kt = (e >> 8) / (FT + 1);
            const int k = kt * 16 + (l >> 4) * 4 + c;
            if (k >= K) continue;
            double* row = out + ((long long)k * MAX_BLOCKS + blockIdx.x) * PX;
            if (tf < FT) {
                const int f = tf * 16 + (l & 15);
                if (f < G::F) row[f] = sc[e];
            } else if ((l & 15) == 0) {
                row[G::F] = sc[e];                             // Nk = sum_n r_nk
            }
        }
        if (wave == 0) {                                       // lane k holds N_k; fixed-order sum; lane k writes row k's slot
            double ntot = 0.0;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int k = kt * 16 + (lane & 15), l = ((k & 15) >> 2) * 16, c = k & 3;
                const double nk = (lane < 16 && k < K) ? sc[((kt * (FT + 1) + FT) * 4 + c) * WAVE + l] : 0.0;
#pragma unroll
                for (int j = 0; j < 16; ++j) ntot += readlane_d(nk, j);
            }
            for (int k = lane; k < K; k += WAVE) out[((long long)k * MAX_BLOCKS + blockIdx.x) * PX + G::PF] = ntot;
        }
        PASS_TS(5);
    }
}

template <int D, int KT, int FLAV, bool ESTEP, bool STATS, bool MASK>
__global__ __launch_bounds__(max_nw(KT) * WAVE) void pass_kernel(PassArgs a) {
    using G = Geo<D>;
    constexpr int FT = G::FT;
    constexpr bool SMM = (FLAV == VMP_SMM);
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int K = a.K;
    PASS_TS(0);
#ifdef VMP_DEBUG_TS
    if (a.dbg_t && (blockIdx.x == 0 || blockIdx.x == 100) && (threadIdx.x & 63) == 0)
        a.dbg_t[(blockIdx.x ? 64 : 0) + (threadIdx.x >> 6) * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
#endif
    float* xl = smem + wave * (G::XROWS * LS);     // [XROWS][LS]: (x - pivot) columns, ones, zeros
    constexpr int ONE = D, ZERO = D + 1;
    xl[ONE * LS + lane] = 1.0f;
    xl[ZERO * LS + lane] = 0.0f;
    // Within a value-row the 64 data rows of a tile are stored so that the two rows one lane works on in a group of 8
    // (n8 + kk and n8 + 4 + kk) are ADJACENT: position of tile row t = 8 (t / 8) + 2 (t % 4) + (t / 4) % 2.  A lane then
    // fetches both with one ds_read_b64 (half the LDS cycles of ds_read2_b32).  Banks of ds_read_b64 are dword-address
    // mod 64: with LS = 4 (mod 64) the feature reads of a half-wave (value-row ra = 0..9 by lane, kk in {0,1} or {2,3})
    // fall on banks 4 ra + {0..3} (+ 4): all distinct; the E-part reads are two broadcast addresses per half-wave.
    const int ppos = (lane & ~7) + 2 * (lane & 3) + ((lane >> 2) & 1);

    // (The first rows are requested before the parameters, so that both are in flight together.)
    // Each wave owns a CONTIGUOUS row range of the same length (a multiple of 8 rows): with whole 64-row tiles dealt
    // round-robin the waves of a full chip get 5 or 6 tiles each at N=1e6 and everyone waits for the 6s.
    const bool vec = a.vec_ok != 0;
    long long lo, hi;
    if (a.rpw_b == a.rpw) {
        lo = ((long long)blockIdx.x * nw + wave) * a.rpw;
        hi = lo + a.rpw;
    } else {
        const int hw = nw >> 1;
        const long long base = (long long)blockIdx.x * hw * (a.rpw + a.rpw_b);
        lo = wave < hw ? base + wave * a.rpw : base + hw * a.rpw + (wave - hw) * a.rpw_b;
        hi = lo + (wave < hw ? a.rpw : a.rpw_b);
    }
    if (hi > a.N) hi = a.N;
    float xr[D];
    {
        const long long n = lo + lane;
#pragma unroll
        for (int j = 0; j < D; ++j) xr[j] = 0.f;
        if (n < hi) load_row<D>(a.x + n * D, xr, vec);
    }
    const int i16 = lane & 15, kk = lane >> 4;
    // NOTE on every "cond ? load : 0" below: a load under a per-element condition compiles to a branch plus a full
    // s_waitcnt per element (serialised round trips).  Loads are therefore issued unconditionally from an address
    // that is always valid, and the condition selects the VALUE afterwards.
    float pv[D];
    {
        const float* __restrict__ pp = a.pivot ? a.pivot : a.x;      // a.x: any valid address
        const bool hasp = a.pivot != nullptr;
#pragma unroll
        for (int j = 0; j < D; ++j) { const float v = pp[j]; pv[j] = hasp ? v : 0.f; }
    }

    // ---- this lane's component parameters, resident for the whole kernel
    constexpr int MP = (D + 1) / 2, WP = (G::TRI + 1) / 2;
    v2f pm2[KT][MP], pw2[KT][WP], pch[KT];
    float pua[KT], pub[KT];
    if constexpr (ESTEP) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int k = kt * 16 + i16;
            const bool on = k < K;
            const float* __restrict__ p = a.pack + (on ? k : 0) * G::PACK;
            float raw[G::PACK];
            if (G::PACK % 4 == 0 && (reinterpret_cast<uintptr_t>(a.pack) & 15) == 0) {
                // 16-byte loads: a quarter of the requests.  (The ~5 k cycles a wave spends before its first row is processed -
                // tools/pass_ts.py - did not move with this nor with the order of the requests: first-touch latency.)
#pragma unroll
                for (int j = 0; j < G::PACK / 4; ++j) {
                    const float4 q = reinterpret_cast<const float4*>(p)[j];
                    raw[4 * j] = q.x; raw[4 * j + 1] = q.y; raw[4 * j + 2] = q.z; raw[4 * j + 3] = q.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < G::PACK; ++j) raw[j] = p[j];              // all loads in flight together
            }
#pragma unroll
            for (int j = 0; j < 2 * MP; ++j) pm2[kt][j >> 1][j & 1] = (on && j < D) ? raw[j < D ? j : 0] - pv[j < D ? j : 0] : 0.f;
#pragma unroll
            for (int j = 0; j < 2 * WP; ++j) pw2[kt][j >> 1][j & 1] = (on && j < G::TRI) ? raw[D + (j < G::TRI ? j : 0)] : 0.f;
            pch[kt].x = on ? raw[D + G::TRI] : -INFINITY;     // log2-domain constant; -inf switches the lane off
            pch[kt].y = on ? raw[D + G::TRI + 1] : 0.f;
            pua[kt] = on ? raw[D + G::TRI + 2] : 0.f;
            pub[kt] = on ? raw[D + G::TRI + 3] : 1.f;
        }
    }

    // ---- MFMA B-operand addressing: lane = (feature column i16, inner index kk = data row n0+kk)
    int offA[FT], offB[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
        const int f = ft * 16 + i16;
        int ra = ZERO, rb = ZERO;
        if (f == 0) { ra = ONE; rb = ONE; }
        else if (f <= D) { ra = f - 1; rb = ONE; }
        else if (f < G::F) {
            int p = f - D - 1, d = 0;
            while (p >= D - d) { p -= D - d; ++d; }
            ra = d; rb = d + p;
        }
        offA[ft] = ra * LS + 2 * kk;
        offB[ft] = rb * LS + 2 * kk;
    }

    f32x4 acc[KT][FT], acs[KT][FT];
    f32x4 nacc[KT], nacs[KT];
    double dacc[KT][FT][4];
    double dn[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        nacc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        nacs[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) dn[kt][c] = 0.0;
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
            acc[kt][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
            acs[kt][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) dacc[kt][ft][c] = 0.0;
        }
    }

#ifdef VMP_DEBUG_TS
    if (a.dbg_t) {                                                        // parameters and first rows have arrived
        float chk = xr[0] + pv[0];
        if constexpr (ESTEP) chk += pm2[0][0][0] + pch[0].x;
        asm volatile("" :: "v"(chk));
        PASS_TS(1);
    }
#endif
    for (long long row0 = lo; row0 < hi; row0 += TR) {
        const int trows = (hi - row0 < TR) ? (int)(hi - row0) : TR;      // rows of this (possibly partial) tile
        // stage this tile's rows (transposed, pivot-shifted), then prefetch the next tile's row
        {
            const bool valid = row0 + lane < hi;
#pragma unroll
            for (int j = 0; j < D; ++j) xl[j * LS + ppos] = valid ? xr[j] - pv[j] : 0.f;
            const long long n2 = row0 + TR + lane;
#pragma unroll
            for (int j = 0; j < D; ++j) xr[j] = 0.f;
            if (n2 < hi) load_row<D>(a.x + n2 * D, xr, vec);
        }
        __builtin_amdgcn_wave_barrier();

        // this lane's element of r/u/logr for (row0 + kk, component i16) - everything else is a 32-bit offset
        const long long tbase = (row0 + kk) * K + i16;
        const int K4 = 4 * K;
        // A tile is walked in bodies of 32 rows = 4 groups of 8 rows.  One group = the E-part for this lane's two rows
        // (n8 + kk, n8 + 4 + kk); its weights and feature products are split into bf16 (hi, lo) pairs and parked in the
        // operand registers of the 16x16x32 bf16 MFMA (slot t = 2u + h of lane group kk <-> row 8u + 4h + kk, the same
        // for A and B), so that the moment GEMM of the body is 3 products (hi hi + hi lo + lo hi) per feature tile on the
        // MATRIX pipe instead of 8 fp32 MFMAs per tile on the vector pipe (fp32 MFMA shares the VALU issue slots on
        // gfx950).  FULL bodies (32 valid rows, K a multiple of 16) skip every predicate.
#pragma unroll 1
        for (int n0 = 0; n0 < trows; n0 += 32) {
            unsigned As[KT][3][4], Rs[KT][3][4], Bs[FT][3][4];          // [term h/m/l][group u]
            auto group = [&](auto full_c, auto u_c) __attribute__((always_inline)) {
                constexpr bool FULL = decltype(full_c)::value;
                constexpr int u = decltype(u_c)::value;
                const int n8 = n0 + 8 * u;
                if (!FULL && n8 >= trows) {                                   // wave-uniform: nothing left in this body
                    if constexpr (STATS) {
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
#pragma unroll
                            for (int kt = 0; kt < KT; ++kt) { As[kt][t][u] = 0u; Rs[kt][t][u] = 0u; }
#pragma unroll
                            for (int ft = 0; ft < FT; ++ft) Bs[ft][t][u] = 0u;
                        }
                    }
                    return;
                }
                const long long ra = row0 + n8 + kk, rb = ra + 4;             // this lane's two data rows
                const bool va = FULL || ra < hi, vb = FULL || rb < hi;
                const int so = n8 * K;                                        // wave-uniform
                v2f w[KT], rr[KT];
                // the LDS reads of the moment features are issued here, ahead of the E-part, so that their latency is
                // covered by its ~350 cycles of arithmetic instead of being waited for three times at the end of the group
                v2f fa[FT], fb[FT];
                auto load_features = [&]() __attribute__((always_inline)) {
#pragma unroll
                    for (int ft = 0; ft < FT; ++ft) {
                        fa[ft] = *reinterpret_cast<const v2f*>(&xl[offA[ft] + n8]);
                        fb[ft] = *reinterpret_cast<const v2f*>(&xl[offB[ft] + n8]);
                    }
                };
                if constexpr (ESTEP) {
                    v2f xv[D];
#pragma unroll
                    for (int j = 0; j < D; ++j) xv[j] = *reinterpret_cast<const v2f*>(&xl[j * LS + n8 + 2 * kk]);
                    v2f keep[D];
                    if constexpr (MASK) {
#pragma unroll
                        for (int j = 0; j < D; ++j)
                            keep[j] = v2f{(va && a.mask[ra * D + j] != 0) ? 0.f : 1.f, (vb && a.mask[rb * D + j] != 0) ? 0.f : 1.f};
                    }
                    v2f lg[KT], uu[KT];
                    v2f mx;
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        v2f dv[D];
#pragma unroll
                        for (int j = 0; j < D; ++j) {
                            dv[j] = pk_sub_b(xv[j], pm2[kt][j >> 1], j & 1);
                            if constexpr (MASK) dv[j] = dv[j] * keep[j];
                        }
                        // y = W (x - m), W lower triangular packed row-major; walked by COLUMNS so that the D
                        // accumulators form independent dependency chains (a row-wise walk is latency-bound)
                        v2f y[D];
#pragma unroll
                        for (int i = 0; i < D; ++i) y[i] = pk_mul_b(dv[0], pw2[kt][(i * (i + 1) / 2) >> 1], (i * (i + 1) / 2) & 1);
#pragma unroll
                        for (int j = 1; j < D; ++j)
#pragma unroll
                            for (int i = j; i < D; ++i) {
                                const int e = i * (i + 1) / 2 + j;
                                y[i] = pk_fma_b(dv[j], pw2[kt][e >> 1], y[i], e & 1);
                            }
                        v2f q = y[0] * y[0], q1 = v2f{0.f, 0.f};
#pragma unroll
                        for (int i = 1; i < D; ++i) {
                            if (i & 1) q1 = __builtin_elementwise_fma(y[i], y[i], q1);
                            else q = __builtin_elementwise_fma(y[i], y[i], q);
                        }
                        q += q1;
                        lg[kt] = pk_const_minus_scaled(q, pch[kt]);           // log2 rho
                        if (kt == 0) mx = lg[0];
                        else mx = v2f{fmaxf(mx.x, lg[kt].x), fmaxf(mx.y, lg[kt].y)};
                        if constexpr (SMM) uu[kt] = v2f{pua[kt] * __builtin_amdgcn_rcpf(q.x + pub[kt]), pua[kt] * __builtin_amdgcn_rcpf(q.y + pub[kt])};
                    }
                    mx = row16_max2(mx);
                    v2f ssum;
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        lg[kt] = v2f{__builtin_amdgcn_exp2f(lg[kt].x - mx.x), __builtin_amdgcn_exp2f(lg[kt].y - mx.y)};
                        if (kt == 0) ssum = lg[0];
                        else ssum += lg[kt];
                    }
                    ssum = row16_sum2(ssum);
                    v2f inv = v2f{__builtin_amdgcn_rcpf(ssum.x), __builtin_amdgcn_rcpf(ssum.y)};
                    if constexpr (!FULL) inv = v2f{va ? inv.x : 0.f, vb ? inv.y : 0.f};
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        const int k = kt * 16 + i16;
                        rr[kt] = lg[kt] * inv;
                        w[kt] = SMM ? rr[kt] * uu[kt] : rr[kt];
                        if constexpr (FULL) {
                            // K == 16 KT here: every offset below the tile base is a compile-time constant
                            float* __restrict__ ro = a.r_out + tbase + kt * 16 + (long long)n0 * (16 * KT);
                            // streaming stores: r (and u) are written once per iteration and never read by these kernels
                            __builtin_nontemporal_store(rr[kt].x, ro + 8 * u * 16 * KT);
                            __builtin_nontemporal_store(rr[kt].y, ro + (8 * u + 4) * 16 * KT);
                            if constexpr (SMM) {
                                float* __restrict__ uo = a.u_out + tbase + kt * 16 + (long long)n0 * (16 * KT);
                                __builtin_nontemporal_store(uu[kt].x, uo + 8 * u * 16 * KT);
                                __builtin_nontemporal_store(uu[kt].y, uo + (8 * u + 4) * 16 * KT);
                            }
                            if (a.logr_out) {
                                float* __restrict__ lo = a.logr_out + tbase + kt * 16 + (long long)n0 * (16 * KT);
                                lo[8 * u * 16 * KT] = logf(rr[kt].x);
                                lo[(8 * u + 4) * 16 * KT] = logf(rr[kt].y);
                            }
                        } else {
                            const bool sa = va && k < K, sb = vb && k < K;
                            float* __restrict__ ro = a.r_out + tbase + kt * 16;
                            if (sa) ro[so] = rr[kt].x;
                            if (sb) ro[so + K4] = rr[kt].y;
                            if constexpr (SMM) {
                                float* __restrict__ uo = a.u_out + tbase + kt * 16;
                                if (sa) uo[so] = uu[kt].x;
                                if (sb) uo[so + K4] = uu[kt].y;
                            }
                            if (a.logr_out) {
                                float* __restrict__ lo = a.logr_out + tbase + kt * 16;
                                if (sa) lo[so] = logf(rr[kt].x);
                                if (sb) lo[so + K4] = logf(rr[kt].y);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        const int k = kt * 16 + i16;
                        const bool on = k < K;
                        const float* __restrict__ ri = a.r_in + tbase + kt * 16;
                        const float r0v = *((on && va) ? ri + so : a.r_in), r1v = *((on && vb) ? ri + so + K4 : a.r_in);
                        rr[kt] = v2f{(on && va) ? r0v : 0.f, (on && vb) ? r1v : 0.f};
                        if constexpr (SMM) {
                            const float* __restrict__ ui = a.u_in + tbase + kt * 16;
                            const float u0v = *((on && va) ? ui + so : a.u_in), u1v = *((on && vb) ? ui + so + K4 : a.u_in);
                            const v2f u2 = v2f{(on && va) ? u0v : 0.f, (on && vb) ? u1v : 0.f};
                            w[kt] = rr[kt] * u2;
                        } else {
                            w[kt] = rr[kt];
                        }
                    }
                }

                if constexpr (STATS) {
                    load_features();
                    unsigned t3[3];
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        split_bf16<MOM_TERMS>(w[kt], t3);
#pragma unroll
                        for (int t = 0; t < 3; ++t) As[kt][t][u] = t3[t];
                        if constexpr (SMM) {
                            split_bf16<MOM_TERMS>(rr[kt], t3);
#pragma unroll
                            for (int t = 0; t < 3; ++t) Rs[kt][t][u] = t3[t];
                        }
                    }
#pragma unroll
                    for (int ft = 0; ft < FT; ++ft) {
                        split_bf16<MOM_TERMS>(fa[ft] * fb[ft], t3);
#pragma unroll
                        for (int t = 0; t < 3; ++t) Bs[ft][t][u] = t3[t];
                    }
                }
            };
            using std::integral_constant;
            const bool full = (K == 16 * KT) && (n0 + 32 <= trows);
            if (full) {
                group(integral_constant<bool, true>{}, integral_constant<int, 0>{});
                group(integral_constant<bool, true>{}, integral_constant<int, 1>{});
                group(integral_constant<bool, true>{}, integral_constant<int, 2>{});
                group(integral_constant<bool, true>{}, integral_constant<int, 3>{});
            } else {
                group(integral_constant<bool, false>{}, integral_constant<int, 0>{});
                group(integral_constant<bool, false>{}, integral_constant<int, 1>{});
                group(integral_constant<bool, false>{}, integral_constant<int, 2>{});
                group(integral_constant<bool, false>{}, integral_constant<int, 3>{});
            }

            if constexpr (STATS) {
                bf16x8 b[FT][3];
#pragma unroll
                for (int ft = 0; ft < FT; ++ft)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        b[ft][t] = __builtin_bit_cast(bf16x8, u32x4{Bs[ft][t][0], Bs[ft][t][1], Bs[ft][t][2], Bs[ft][t][3]});
                // sum over (ta, tb) with ta + tb < MOM_TERMS; within one (ta, tb) the FT MFMAs hit different accumulators,
                // so back-to-back MFMAs never share one
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    bf16x8 av[3], rv[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        av[t] = __builtin_bit_cast(bf16x8, u32x4{As[kt][t][0], As[kt][t][1], As[kt][t][2], As[kt][t][3]});
                        if constexpr (SMM) rv[t] = __builtin_bit_cast(bf16x8, u32x4{Rs[kt][t][0], Rs[kt][t][1], Rs[kt][t][2], Rs[kt][t][3]});
                    }
                    // The h h products (full magnitude) and the five corrections (<= 2^-8 of it) go to SEPARATE fp32
                    // accumulators: a correction added to a large accumulator loses most of its bits, and six roundings at
                    // the large magnitude per body cost ~3x the accuracy of the fp32 chain this replaces (measured on the
                    // 60-row golden SMM case: C_k 2.2e-6 vs 6.8e-7).  Both are summed in fp64 at the tile flush.
#pragma unroll
                    for (int ta = 0; ta < MOM_TERMS; ++ta) {
#pragma unroll
                        for (int tb = 0; tb + ta < MOM_TERMS; ++tb) {
#pragma unroll
                            for (int ft = 0; ft < FT; ++ft) {
                                if (ta + tb == 0) acc[kt][ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ta], b[ft][tb], acc[kt][ft], 0, 0, 0);
                                else acs[kt][ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ta], b[ft][tb], acs[kt][ft], 0, 0, 0);
                            }
                        }
                        // N_k = sum r: column 0 of feature tile 0 is the constant 1 (exact in bf16: only its h term is non-zero)
                        if constexpr (SMM) {
                            if (ta == 0) nacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rv[ta], b[0][0], nacc[kt], 0, 0, 0);
                            else nacs[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rv[ta], b[0][0], nacs[kt], 0, 0, 0);
                        }
                    }
                }
            }
        }

        if constexpr (STATS) {
            // fp32 accumulators -> fp64 every VMP_MOM_FLUSH-th tile and after the wave's last tile (as pass_xdl_kernel)
            if (((row0 - lo) / TR) % VMP_MOM_FLUSH == VMP_MOM_FLUSH - 1 || row0 + TR >= hi) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if constexpr (SMM) dn[kt][c] += (double)nacc[kt][c] + (double)nacs[kt][c];
#pragma unroll
                        for (int ft = 0; ft < FT; ++ft) dacc[kt][ft][c] += (double)acc[kt][ft][c] + (double)acs[kt][ft][c];
                    }
                    nacc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    nacs[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ft = 0; ft < FT; ++ft) { acc[kt][ft] = f32x4{0.f, 0.f, 0.f, 0.f}; acs[kt][ft] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    PASS_TS(2);
    if constexpr (STATS) pass_epilogue<D, KT, FLAV>(a, smem, dacc, dn, lane, wave, nw);
}


// ---------------------------------------------------------------------------------------------------------
// The same pass with the E-part's quadratic forms ON THE XDL PIPE (round 3; K <= 16, no missing-data mask).
//
// Round-2 counters: ~100 of the 153 VALU instructions an 8-row group costs are the sixteen packed-fp32 forms
// q_nk = ||W_k (x_n - m_k)||^2, and the kernel is VALU-issue bound.  Here  y = W_k x' + b_k  (x' = x - pivot,
// b_k = -W_k (m_k - pivot), formed in fp64 once per kernel) is a GEMM on v_mfma_f32_16x16x32_bf16:
//     M = data row (16 per MFMA),  N = component k (lane & 15),  one MFMA tile per output coordinate i of y,
//     contraction = the D coordinates of x' - only 8 of the 32 k-slots - so the slots carry SEVERAL bf16 TERMS of the
//     same values (v = h + m + l, 8 bits each): lane group g = lane >> 4 of the two MFMAs of a tile multiplies
//         MFMA 1:  x'_h W_h | (1,1,1) (b_h,b_m,b_l) | x'_h W_m | x'_m W_h        MFMA 2:  x'_h W_l | x'_l W_h | x'_m W_m | 0
//     (round 5 order: the bias that cancels most of the leading products sits in the SAME MFMA as they do, see below)
//     i.e. the six products of order <= 2 (dropped: <= 2^-24 relative, the fp32 rounding level) plus the bias.
// No expanded quadratic form x^T Theta x is ever evaluated: y is formed exactly as the fp32 chain formed it (same
// cancellation between W x' and W m'), then q = sum_i y_i^2 costs 8 FMAs per cell in the accumulator registers.
// Operands: the x' tile is split ONCE per row at staging time (lane = row) into an LDS image [row][x_h | x_m | x_l | 1 | 0]
// (16 B each) that the A operands are fetched from with one ds_read_b128 per MFMA; W's terms live in 16 D VGPRs.
// The accumulator layout (lane = (k, g), register v <-> tile row 4 g + v) is mapped to DATA rows 4 v + g, so that one
// store instruction covers 4 consecutive rows x 16 components (256 B) and the moment GEMM's k-slots (tile, v) take
// the responsibilities straight from the softmax registers.
// Measured against the packed-fp32 kernel above on one box: see DESIGN.md section 6.
// ---------------------------------------------------------------------------------------------------------
// u32 per row of the bf16 A image: x_h(4) x_m(4) x_l(4) ones(4) zeros(4) | (x_h|x_h)(4) (x_m|x_h)(4) (x_l|x_m)(4) of the
// coordinates 0..3 (operands of the one-MFMA tiles, see below) | pad; 144 B = 9 16-byte slots: 16 rows hit 16 bank groups
constexpr int AIS = 36;

struct FinArgs {
    const double* partials;    // [K][MAX_BLOCKS][PF + 1]   (src == 0)
    const double* stats_in;    // [K][SW]         (src == 1)
    int nblk, K, flavour, src, do_post;
    const float *alpha0, *beta0, *m0, *C0, *v0, *kappa;
    const float* pivot;        // the shift the pass kernel applied to x (src == 0 only; NULL: none)
    float *alpha, *beta, *m, *C, *v, *xbar, *S, *pi, *pack;
    double* pack64;            // the same E-step pack in fp64 (vmp_mix_estep_accurate), or NULL
    double* stats_out;
    // one-launch data-parallel form (vmp_mix_finalize_exchange): peer[g] = rank g's exchange buffer; nranks = 0: no exchange
    double* peer[VMP_EXCH_MAX_RANKS];
    int nranks, rank;
    unsigned long long iter;
    int* status;
#ifdef VMP_DEBUG_TS
    long long* dbg_t;          // exploration builds only: 8 timestamps of block 0 / thread 0
#endif
};

#ifdef VMP_DEBUG_TS
#define FIN_TS(i) do { if (a.dbg_t && blockIdx.x == 0 && tid == 0) { a.dbg_t[i] = clock64(); if ((i) == 0) a.dbg_t[6] = wall_clock64(); if ((i) == 5) a.dbg_t[7] = wall_clock64(); } } while (0)
#else
#define FIN_TS(i) do { } while (0)
#endif

template <int D>
__device__ void finalize_block(const FinArgs& a, const int k, const int tid, const int nthreads);   // defined below the pass kernels

template <int D> constexpr int xdl_wave_floats() { return Geo<D>::XROWS * LS + TR * AIS; }

// MT: bf16 terms per operand of the MOMENT GEMM.  3 (six products: fp32-equivalent products) below VMP_MOM2_ROWS rows; 2 (three
// products, 2^-17 relative per product, unbiased) from there on: every moment is then a sum over >= 1e4 rows per component whose
// per-term rounding errors average out (relative error of a moment ~ 2^-17 / sqrt(rows of the component) < 1e-7), while the splits
// and MFMAs they save are a fifth of the kernel's issue time (round 5).  The E-part (one value per row, nothing averages) keeps 3.
template <int D, int FLAV, bool STATS, int MT = 3>
__device__ __forceinline__ void pass_xdl_body(const PassArgs& a) {
    using G = Geo<D>;
    constexpr int FT = G::FT, KT = 1;
    constexpr bool SMM = (FLAV == VMP_SMM);
    constexpr int DP = (D + 1) / 2;                          // coordinate pairs
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int K = a.K;
    PASS_TS(0);
    float* xl = smem + wave * xdl_wave_floats<D>();          // [XROWS][LS] fp32 (x - pivot), ones, zeros: moment features
    unsigned* ai = reinterpret_cast<unsigned*>(xl + G::XROWS * LS);   // [64][AIS] bf16 terms of x - pivot: E-part A operands
    constexpr int ONE = D, ZERO = D + 1;
    // position of tile row t in a value-row of the fp32 image: the 4 rows {g, 4+g, 8+g, 12+g} of a 16-row sub-tile that one
    // lane group owns are adjacent (one ds_read_b128 per feature factor)
    const int ppos = (lane & ~15) + 4 * (lane & 3) + ((lane >> 2) & 3);
    xl[ONE * LS + lane] = 1.0f;
    xl[ZERO * LS + lane] = 0.0f;
    {
        u32x4 ones = u32x4{0x3F803F80u, 0x00003F80u, 0u, 0u}, zeros = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(ai + lane * AIS + 12) = ones;
        *reinterpret_cast<u32x4*>(ai + lane * AIS + 16) = zeros;
    }

    const bool vec = a.vec_ok != 0;
    long long lo, hi;
    if (a.rpw_b == a.rpw) {
        lo = ((long long)blockIdx.x * nw + wave) * a.rpw;
        hi = lo + a.rpw;
    } else {
        const int hw = nw >> 1;
        const long long base = (long long)blockIdx.x * hw * (a.rpw + a.rpw_b);
        lo = wave < hw ? base + wave * a.rpw : base + hw * a.rpw + (wave - hw) * a.rpw_b;
        hi = lo + (wave < hw ? a.rpw : a.rpw_b);
    }
    if (hi > a.N) hi = a.N;
    float xr[D];
    {
        const long long n = lo + lane;
#pragma unroll
        for (int j = 0; j < D; ++j) xr[j] = 0.f;
        if (n < hi) load_row<D>(a.x + n * D, xr, vec);
    }
    const int i16 = lane & 15, kk = lane >> 4;
    float pv[D];
    {
        const float* __restrict__ pp = a.pivot ? a.pivot : a.x;
        const bool hasp = a.pivot != nullptr;
#pragma unroll
        for (int j = 0; j < D; ++j) { const float v = pp[j]; pv[j] = hasp ? v : 0.f; }
    }

    // ---- B operands of the y GEMM: lane (k = i16, g = kk) holds the terms of W_k its lane group multiplies (see the header)
    constexpr int NS = D < 4 ? D : 4;                        // output coordinates done by ONE MFMA (contraction over x'_0..3)
    constexpr int NB = D > 4 ? D - 4 : 0;                    // output coordinates 4.. : two MFMAs (contraction over x'_0..7)
    u32x4 Bsm[NS], B1[NB > 0 ? NB : 1], B2[NB > 0 ? NB : 1];
    v2f pch;
    float pua, pub;
    {
        const bool on = i16 < K;
        const float* __restrict__ p = a.pack + (on ? i16 : 0) * G::PACK;
        float raw[G::PACK];
        if (G::PACK % 4 == 0 && (reinterpret_cast<uintptr_t>(a.pack) & 15) == 0) {
#pragma unroll
            for (int j = 0; j < G::PACK / 4; ++j) {
                const float4 q = reinterpret_cast<const float4*>(p)[j];
                raw[4 * j] = q.x; raw[4 * j + 1] = q.y; raw[4 * j + 2] = q.z; raw[4 * j + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < G::PACK; ++j) raw[j] = p[j];
        }
        pch.x = on ? raw[D + G::TRI] : -INFINITY;            // log2-domain constant; -inf switches the lane off
        pch.y = on ? raw[D + G::TRI + 1] : 0.f;
        pua = on ? raw[D + G::TRI + 2] : 0.f;
        pub = on ? raw[D + G::TRI + 3] : 1.f;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double bi = 0.0;                                 // b_i = -sum_j W_ij (m_j - pivot_j), fp64
#pragma unroll
            for (int j = 0; j <= i; ++j) bi -= (double)raw[D + i * (i + 1) / 2 + j] * ((double)raw[j] - (double)pv[j]);
            unsigned tb[3];
            split_bf16<3>(v2f{(float)bi, 0.f}, tb);          // low halves: b_h, b_m, b_l
            const unsigned bias0 = (tb[0] & 0xffffu) | (tb[1] << 16), bias1 = tb[2] & 0xffffu;   // slots (b_h, b_m), (b_l, 0)
            unsigned th[4] = {0u, 0u, 0u, 0u}, tm[4] = {0u, 0u, 0u, 0u}, tl[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int q = 0; q < DP; ++q) {
                const int j0 = 2 * q, j1 = 2 * q + 1;
                const float wa = j0 <= i ? raw[D + i * (i + 1) / 2 + j0] : 0.f;
                const float wb = (j1 <= i && j1 < D) ? raw[D + i * (i + 1) / 2 + (j1 <= i ? j1 : 0)] : 0.f;
                unsigned t3[3];
                split_bf16<3>(v2f{wa, wb}, t3);
                th[q] = t3[0]; tm[q] = t3[1]; tl[q] = t3[2];
            }
            if (i < NS) {
                // (W_h|W_m), (W_h|W_l), (W_h|W_m), bias  over the coordinate pairs (0,1), (2,3)
                const u32x4 w = kk == 0 ? u32x4{th[0], th[1], tm[0], tm[1]}
                              : kk == 1 ? u32x4{th[0], th[1], tl[0], tl[1]}
                              : kk == 2 ? u32x4{th[0], th[1], tm[0], tm[1]}
                                        : u32x4{bias0, bias1, 0u, 0u};
                Bsm[i] = on ? w : u32x4{0u, 0u, 0u, 0u};
            } else {
                const int ib = i - 4;
                // MFMA 1: W_h | bias | W_m | W_h  (x x'_h | ones | x'_h | x'_m: the leading products, the bias that cancels most of them and the
                //          first-order corrections)      MFMA 2: W_l | W_h | W_m | 0  (x x'_h | x'_l | x'_m: the second-order corrections).
                // The MFMA rounds its sum ONCE, at the exponent of its LARGEST term (tools/ubench/mfma_cancel_numerics.hip): with the
                // bias beside the leading products the first MFMA leaves y itself (|y| ~ 3 where |W x'|, |b| ~ 15) and the second one adds
                // terms of 2^-16 of that to it - one rounding at the scale of the large terms instead of two (round 4 had the bias in the
                // second MFMA); tools/r5_smm_error_budget.py: max |r - r_fp64| of the SMM 6.1e-6 -> 2.4e-6 in emulation.
                const u32x4 w1 = kk == 0 ? u32x4{th[0], th[1], th[2], th[3]} : kk == 1 ? u32x4{bias0, bias1, 0u, 0u}
                               : kk == 2 ? u32x4{tm[0], tm[1], tm[2], tm[3]} : u32x4{th[0], th[1], th[2], th[3]};
                const u32x4 w2 = kk == 0 ? u32x4{tl[0], tl[1], tl[2], tl[3]} : kk == 1 ? u32x4{th[0], th[1], th[2], th[3]}
                               : kk == 2 ? u32x4{tm[0], tm[1], tm[2], tm[3]} : u32x4{0u, 0u, 0u, 0u};
                B1[ib < 0 ? 0 : ib] = on ? w1 : u32x4{0u, 0u, 0u, 0u};
                B2[ib < 0 ? 0 : ib] = on ? w2 : u32x4{0u, 0u, 0u, 0u};
            }
        }
    }
    // A operands: lane (m = i16, g = kk) reads term T[g] of tile row rho(m) = 4 (m & 3) + (m >> 2)
    const int rho = 4 * (i16 & 3) + (i16 >> 2);
    const int offA1 = rho * AIS + (kk == 1 ? 12 : (kk == 3 ? 4 : 0));                    // x_h | ones | x_h | x_m
    const int offA2 = rho * AIS + (kk == 0 ? 0 : (kk == 1 ? 8 : (kk == 2 ? 4 : 16)));    // x_h | x_l | x_m | zeros
    const int offA3 = rho * AIS + (kk == 3 ? 12 : 20 + 4 * kk);                           // (x_h|x_h) | (x_m|x_h) | (x_l|x_m) | ones

    // ---- moment GEMM B-operand addressing: lane = (feature column i16, k-slot group kk)
    int offA[FT], offB[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
        const int f = ft * 16 + i16;
        int ra = ZERO, rb = ZERO;
        if (f == 0) { ra = ONE; rb = ONE; }
        else if (f <= D) { ra = f - 1; rb = ONE; }
        else if (f < G::F) {
            int p = f - D - 1, d = 0;
            while (p >= D - d) { p -= D - d; ++d; }
            ra = d; rb = d + p;
        }
        offA[ft] = ra * LS + 4 * kk;
        offB[ft] = rb * LS + 4 * kk;
    }

    f32x4 acc[KT][FT], acs[KT][FT];
    double dacc[KT][FT][4];
    double dn[KT][4];
    // SMM: N_k = sum_n r_nk (the moment GEMM carries w = r u).  The lane owns component i16, so its rows' r are summed on the
    // VALU (fp32 per tile, fp64 across tiles) and brought into the accumulator layout once, after the last row - instead of
    // a second split of r and three more MFMAs per body.
    float nsum = 0.f;
    double dnl = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) dn[0][c] = 0.0;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
        acc[0][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
        acs[0][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) dacc[0][ft][c] = 0.0;
    }

#ifdef VMP_DEBUG_TS
    asm volatile("" :: "v"(xr[0]), "v"(pv[0]), "v"(pch.x), "v"(B1[0][0]));      // the stamp below is taken when rows, pivot and pack have ARRIVED
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    PASS_TS(1);
    for (long long row0 = lo; row0 < hi; row0 += TR) {
        const int trows = (hi - row0 < TR) ? (int)(hi - row0) : TR;
        // ---- stage this tile: lane = tile row.  fp32 image for the moment features, bf16 term image for the y GEMM
        {
            const bool valid = row0 + lane < hi;
            float xs[2 * DP];
#pragma unroll
            for (int j = 0; j < 2 * DP; ++j) xs[j] = (valid && j < D) ? xr[j < D ? j : 0] - pv[j < D ? j : 0] : 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) xl[j * LS + ppos] = xs[j];
            unsigned th[4] = {0u, 0u, 0u, 0u}, tm[4] = {0u, 0u, 0u, 0u}, tl[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int q = 0; q < DP; ++q) {
                unsigned t3[3];
                split_bf16<3>(v2f{xs[2 * q], xs[2 * q + 1]}, t3);
                th[q] = t3[0]; tm[q] = t3[1]; tl[q] = t3[2];
            }
            *reinterpret_cast<u32x4*>(ai + lane * AIS) = u32x4{th[0], th[1], th[2], th[3]};
            *reinterpret_cast<u32x4*>(ai + lane * AIS + 4) = u32x4{tm[0], tm[1], tm[2], tm[3]};
            *reinterpret_cast<u32x4*>(ai + lane * AIS + 8) = u32x4{tl[0], tl[1], tl[2], tl[3]};
            *reinterpret_cast<u32x4*>(ai + lane * AIS + 20) = u32x4{th[0], th[1], th[0], th[1]};
            *reinterpret_cast<u32x4*>(ai + lane * AIS + 24) = u32x4{tm[0], tm[1], th[0], th[1]};
            *reinterpret_cast<u32x4*>(ai + lane * AIS + 28) = u32x4{tl[0], tl[1], tm[0], tm[1]};
            const long long n2 = row0 + TR + lane;
#pragma unroll
            for (int j = 0; j < D; ++j) xr[j] = 0.f;
            if (n2 < hi) load_row<D>(a.x + n2 * D, xr, vec);
        }
        __builtin_amdgcn_wave_barrier();

        const long long tbase = (row0 + kk) * K + i16;       // this lane's element of r/u for (row0 + kk, component i16)
#pragma unroll 1
        for (int n0 = 0; n0 < trows; n0 += 32) {
            const bool full = (K == 16) && (n0 + 32 <= trows);
            // One instance of the whole body (two sub-tiles + the moment MFMAs) per path: the operand registers of the MFMAs
            // are written and consumed inside the same instance, so no register tuple has to be re-assembled where the two
            // paths would merge (the first version paid 64 v_mov per body for that).
            auto body = [&](auto full_c) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_c)::value;
            unsigned As[3][4], Bs[FT][3][4];                 // [term h/m/l][k-slot pair: (sub-tile jj, v pair)]
            auto subtile = [&](auto jj_c) __attribute__((always_inline)) {
                constexpr int jj = decltype(jj_c)::value;
                const int n16 = n0 + 16 * jj;                // first tile row of this 16-row sub-tile
                if (!FULL && n16 >= trows) {                 // wave-uniform: nothing left
                    if constexpr (STATS) {
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            As[t][2 * jj] = 0u; As[t][2 * jj + 1] = 0u;
#pragma unroll
                            for (int ft = 0; ft < FT; ++ft) { Bs[ft][t][2 * jj] = 0u; Bs[ft][t][2 * jj + 1] = 0u; }
                        }
                    }
                    return;
                }
                // ---- y = W x' + b on the XDL pipe, q = |y|^2 in the accumulator registers
                f32x4 y[D];
                {
                    const bf16x8 a3 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ai + n16 * AIS + offA3));
#pragma unroll
                    for (int i = 0; i < NS; ++i)
                        y[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, __builtin_bit_cast(bf16x8, Bsm[i]), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                }
                if constexpr (NB > 0) {
                    const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ai + n16 * AIS + offA1));
                    const bf16x8 a2 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ai + n16 * AIS + offA2));
#pragma unroll
                    for (int i = 0; i < NB; ++i)
                        y[4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8, B1[i]), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < NB; ++i)
                        y[4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, __builtin_bit_cast(bf16x8, B2[i]), y[4 + i], 0, 0, 0);
                }
                f32x4 q4 = y[0] * y[0], q4b = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 1; i < D; ++i) {
                    if (i & 1) q4b = __builtin_elementwise_fma(y[i], y[i], q4b);
                    else q4 = __builtin_elementwise_fma(y[i], y[i], q4);
                }
                q4 += q4b;
                // register v <-> data row n16 + 4 v + kk
                v2f lg[2], uu[2], rr[2], w[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const v2f qh = v2f{q4[2 * h], q4[2 * h + 1]};
                    lg[h] = pk_const_minus_scaled(qh, pch);                 // log2 rho
                    if constexpr (SMM) uu[h] = v2f{pua * __builtin_amdgcn_rcpf(qh.x + pub), pua * __builtin_amdgcn_rcpf(qh.y + pub)};
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const v2f mx = row16_max2(lg[h]);
                    lg[h] = v2f{__builtin_amdgcn_exp2f(lg[h].x - mx.x), __builtin_amdgcn_exp2f(lg[h].y - mx.y)};
                    const v2f ssum = row16_sum2(lg[h]);
                    v2f inv = v2f{__builtin_amdgcn_rcpf(ssum.x), __builtin_amdgcn_rcpf(ssum.y)};
                    if constexpr (!FULL) {
                        const bool va = row0 + n16 + 8 * h + kk < hi, vb = row0 + n16 + 8 * h + 4 + kk < hi;
                        inv = v2f{va ? inv.x : 0.f, vb ? inv.y : 0.f};
                    }
                    rr[h] = lg[h] * inv;
                    w[h] = SMM ? rr[h] * uu[h] : rr[h];
                }
                // ---- stores: for fixed v the 64 lanes cover rows n16 + 4 v .. + 3 x 16 components = 256 contiguous bytes
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float rv = rr[v >> 1][v & 1];
                    if constexpr (FULL) {
                        float* __restrict__ ro = a.r_out + tbase + (long long)(n16 + 4 * v) * 16;
                        __builtin_nontemporal_store(rv, ro);
                        if constexpr (SMM) __builtin_nontemporal_store(uu[v >> 1][v & 1], a.u_out + tbase + (long long)(n16 + 4 * v) * 16);
                        if (a.logr_out) a.logr_out[tbase + (long long)(n16 + 4 * v) * 16] = logf(rv);
                    } else {
                        const bool sv = (row0 + n16 + 4 * v + kk < hi) && i16 < K;
                        const long long o = tbase + (long long)(n16 + 4 * v) * K;
                        if (sv) {
                            a.r_out[o] = rv;
                            if constexpr (SMM) a.u_out[o] = uu[v >> 1][v & 1];
                            if (a.logr_out) a.logr_out[o] = logf(rv);
                        }
                    }
                }
                if constexpr (STATS) {
                    unsigned t3[3];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        split_bf16<MT>(w[h], t3);
#pragma unroll
                        for (int t = 0; t < 3; ++t) As[t][2 * jj + h] = t3[t];
                    }
                    if constexpr (SMM) nsum += (rr[0].x + rr[0].y) + (rr[1].x + rr[1].y);
#pragma unroll
                    for (int ft = 0; ft < FT; ++ft) {
                        const f32x4 fa = *reinterpret_cast<const f32x4*>(&xl[offA[ft] + n16]);
                        const f32x4 fb = *reinterpret_cast<const f32x4*>(&xl[offB[ft] + n16]);
                        const f32x4 pr = fa * fb;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            split_bf16<MT>(v2f{pr[2 * h], pr[2 * h + 1]}, t3);
#pragma unroll
                            for (int t = 0; t < 3; ++t) Bs[ft][t][2 * jj + h] = t3[t];
                        }
                    }
                }
            };
            using std::integral_constant;
            subtile(integral_constant<int, 0>{});
            subtile(integral_constant<int, 1>{});

            if constexpr (STATS) {
                bf16x8 b[FT][3];
#pragma unroll
                for (int ft = 0; ft < FT; ++ft)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        b[ft][t] = __builtin_bit_cast(bf16x8, u32x4{Bs[ft][t][0], Bs[ft][t][1], Bs[ft][t][2], Bs[ft][t][3]});
                bf16x8 av[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) av[t] = __builtin_bit_cast(bf16x8, u32x4{As[t][0], As[t][1], As[t][2], As[t][3]});
                // h h products and the five corrections in separate fp32 accumulators (see pass_kernel)
#pragma unroll
                for (int ta = 0; ta < MT; ++ta) {
#pragma unroll
                    for (int tb = 0; tb + ta < MT; ++tb) {
#pragma unroll
                        for (int ft = 0; ft < FT; ++ft) {
                            if (ta + tb == 0) acc[0][ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ta], b[ft][tb], acc[0][ft], 0, 0, 0);
                            else acs[0][ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ta], b[ft][tb], acs[0][ft], 0, 0, 0);
                        }
                    }
                }
            }
            };   // body
            if (full) body(std::integral_constant<bool, true>{});
            else body(std::integral_constant<bool, false>{});
        }

        if constexpr (STATS) {
            // fp32 accumulators -> fp64 every VMP_MOM_FLUSH-th tile (default 2 = 128 rows; the fp64 conversions and additions run at a fraction of
            // the fp32 rate: 48 of them per tile were ~5 % of the kernel) and after the wave's last tile
            if (((row0 - lo) / TR) % VMP_MOM_FLUSH == VMP_MOM_FLUSH - 1 || row0 + TR >= hi) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int ft = 0; ft < FT; ++ft) dacc[0][ft][c] += (double)acc[0][ft][c] + (double)acs[0][ft][c];
                }
                if constexpr (SMM) { dnl += (double)nsum; nsum = 0.f; }
#pragma unroll
                for (int ft = 0; ft < FT; ++ft) { acc[0][ft] = f32x4{0.f, 0.f, 0.f, 0.f}; acs[0][ft] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    PASS_TS(2);
    if constexpr (STATS && SMM) {
        // N_k into the accumulator layout the epilogue reads: lane (g, column 0), register c <-> component 4 g + c
        double tot = dnl;
        {
            int lo = __double2loint(tot), hi = __double2hiint(tot);
            tot += __hiloint2double(__shfl_xor(hi, 32), __shfl_xor(lo, 32));
            lo = __double2loint(tot); hi = __double2hiint(tot);
            tot += __hiloint2double(__shfl_xor(hi, 16), __shfl_xor(lo, 16));          // (kk0 + kk2) + (kk1 + kk3): same on every lane
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int src = 4 * kk + c;                                               // a lane whose i16 is component 4 kk + c
            const int lo = __shfl(__double2loint(tot), src), hi = __shfl(__double2hiint(tot), src);
            dn[0][c] = __hiloint2double(hi, lo);
        }
    }
    if constexpr (STATS) pass_epilogue<D, 1, FLAV>(a, smem, dacc, dn, lane, wave, nw);
}

template <int D, int FLAV, bool STATS, int MT = 3>
__global__ __launch_bounds__(MAX_NW1 * WAVE) void pass_xdl_kernel(PassArgs a) {
    pass_xdl_body<D, FLAV, STATS, MT>(a);
}

// ---------------------------------------------------------------------------------------------------------
// K-sized posterior update (one block per component), fp64.
// ---------------------------------------------------------------------------------------------------------
__device__ double digamma_d(double x) {
    double r = 0.0;
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x
           - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760))))));
}

template <int D>
__device__ void write_pack(float* pack, int k, const double* m, const double* W /*lower, row-major full DxD*/,
                           double c, double h, double ua, double ub) {
    using G = Geo<D>;
    float* p = pack + k * G::PACK;
    int idx = D;
    for (int j = 0; j < D; ++j) p[j] = (float)m[j];
    for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) p[idx++] = (float)W[i * D + j];
    const double LOG2E = 1.4426950408889634074;          // the pass kernel evaluates 2^(c - h q)
    p[idx++] = (float)(c * LOG2E); p[idx++] = (float)(h * LOG2E); p[idx++] = (float)ua; p[idx++] = (float)ub;
}

// Cholesky of SPD A (DxD, row-major) -> lower L (in place, upper zeroed).  Returns false if not SPD.
// Fully unrolled so that the matrix lives in registers (runtime-indexed local arrays would go to scratch).
template <int D>
__device__ __forceinline__ bool chol_lower(double (&A)[D * D]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double s = A[j * D + j];
#pragma unroll
        for (int p = 0; p < j; ++p) s -= A[j * D + p] * A[j * D + p];
        ok = ok && (s > 0.0);
        const double d = sqrt(s);
        const double rd = 1.0 / d;
        A[j * D + j] = d;
#pragma unroll
        for (int i = j + 1; i < D; ++i) {
            double t = A[i * D + j];
#pragma unroll
            for (int p = 0; p < j; ++p) t -= A[i * D + p] * A[j * D + p];
            A[i * D + j] = t * rd;
        }
#pragma unroll
        for (int i = 0; i < j; ++i) A[i * D + j] = 0.0;
    }
    return ok;
}

// inverse of lower-triangular L -> Li (lower)
template <int D>
__device__ __forceinline__ void tri_inv_lower(const double (&L)[D * D], double (&Li)[D * D]) {
#pragma unroll
    for (int i = 0; i < D * D; ++i) Li[i] = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        Li[j * D + j] = 1.0 / L[j * D + j];
#pragma unroll
        for (int i = j + 1; i < D; ++i) {
            double s = 0.0;
#pragma unroll
            for (int p = j; p < i; ++p) s += L[i * D + p] * Li[p * D + j];
            Li[i * D + j] = -s / L[i * D + i];
        }
    }
}



// 1/sqrt(x) for a positive, normal x: the hardware estimate (v_rsq_f64, ~2^-26) and two Newton steps - 8 dependent
// operations on the serial chain of the factorisation instead of the library routine's range handling.
__device__ __forceinline__ double fast_rsqrt_d(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}

// Wave-parallel factorisation of the SPD matrix A (DxD in LDS): lane i owns row i.  Cholesky A = L L^T by
// columns (finished entries are broadcast with v_readlane, so the code is uniform across lanes), then lane c
// solves L X = e_c, i.e. holds column c of L^{-1}.  Returns X (column `lane`), sum_i log L_ii and SPD-ness.
template <int D>
__device__ __forceinline__ void wave_chol_inverse(const double* A, int lane, double (&X)[D], double& sumlog, bool& ok) {
    double row[D], rd[D];
#pragma unroll
    for (int e = 0; e < D; ++e)
        row[e] = lane < D ? 0.5 * (A[lane * D + e] + A[e * D + lane]) : (e == lane ? 1.0 : 0.0);
    ok = true;
    double mydiag = 1.0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double t = row[j];
#pragma unroll
        for (int p = 0; p < j; ++p) t -= row[p] * readlane_d(row[p], j);
        const double sj = readlane_d(t, j);              // A_jj - sum_p L_jp^2   (the serial chain: keep it short)
        ok = ok && (sj > 0.0);
        rd[j] = fast_rsqrt_d(sj);                        // 1 / L_jj
        if (lane == j) mydiag = sj;
        row[j] = lane >= j ? t * rd[j] : 0.0;            // lane j: sj / sqrt(sj) = L_jj
    }
    // sum_j log L_jj = 0.5 sum_j log s_j : one log per lane, off the serial chain
    double lg = lane < D ? 0.5 * log(mydiag) : 0.0;
    sumlog = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) sumlog += readlane_d(lg, j);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double acc = (i == lane) ? 1.0 : 0.0;
#pragma unroll
        for (int p = 0; p < i; ++p) acc -= readlane_d(row[p], i) * X[p];
        X[i] = acc * rd[i];
    }
}

// Shared tail: expected log-dets, E log pi, constants and the pack, given C_k's Cholesky-derived W.
template <int D>
__device__ void estep_constants(int k, int flavour, double alpha_k, double alpha_sum, double beta_k, double v_k,
                                double logdetP, double kap, double& c, double& h, double& ua, double& ub, double& elp) {
    const double LOG2 = 0.69314718055994530942, PI = 3.14159265358979323846;
    elp = digamma_d(alpha_k) - digamma_d(alpha_sum);
    double sdg = 0.0;
    if (flavour == VMP_GMM) {
        for (int i = 0; i < D; ++i) sdg += digamma_d(0.5 * (v_k + 1.0 + i));      // gmm.py:128-129
        // gmm.py:120-121: log det P replaced by 0 when det P <= 1e-20
        const double ld = (logdetP > log(1e-20)) ? logdetP : 0.0;
        const double eld = sdg + D * LOG2 + ld;
        c = elp + 0.5 * eld - 0.5 * (D / beta_k);
        h = 0.5; ua = 1.0; ub = 1.0;
    } else {
        for (int i = 0; i < D; ++i) sdg += digamma_d(0.5 * (v_k + i));            // smm.py:107-108
        const double eld = sdg + D * LOG2 + logdetP;                                // smm.py:102 (no guard)
        h = 0.5 * (D + kap);
        // smm.py:122-124 (note the precedence of line 124: ... - (0.5 (D+kappa) m - log kappa))
        c = lgamma(0.5 * (D + kap)) - lgamma(0.5 * kap) - 0.5 * D * log(kap * PI) + elp + 0.5 * eld
            - h * (D / beta_k) + log(kap);
        ua = D + kap;                                                               // smm.py:134-137
        ub = D / beta_k + kap;
    }
}

constexpr int FIN_THREADS = 1024;
constexpr int FIN_MAX_GROUPS = FIN_THREADS / 64;

__device__ __forceinline__ void st_pack(const FinArgs& a, int idx, double v) {
    a.pack[idx] = (float)v;
    if (a.pack64) a.pack64[idx] = v;
}

// One block per component (k), `nthreads` threads (>= 192, a multiple of 64; the stand-alone kernel: 1024, the head of the
// one-launch step: the streaming block's 512).  Phase A: all threads reduce the per-block partials in a fixed order.
// Phase B: 64 lanes build S_k, C_k element-wise.  Phase C: wave 0 factorises C_k while lanes 64.. evaluate
// the digamma / lgamma terms.  Phase D: constants + pack.
template <int D>
__device__ void finalize_block(const FinArgs& a, const int k, const int tid, const int nthreads) {
    using G = Geo<D>;
    __shared__ double part[FIN_MAX_GROUPS][64];
    __shared__ double st[G::SW];           // canonical: Nk, Wk, sx[D], sxx[D*D]
    __shared__ double ntot;                // sum_j N_j over all components
    __shared__ double Ck[D * D], mk[D], sp[D + 4], scal[8];
    __shared__ double alpha0s[VMP_MAX_K];
    const int K = a.K;
    const int FIN_GROUPS = nthreads >> 6;
    // issue every small prior load up front so that its latency overlaps the partial-sum loads below
    const bool post = a.do_post != 0;
    const bool smm = a.flavour == VMP_SMM;
    const double beta0 = post ? (double)a.beta0[k] : 0.0, v0 = post ? (double)a.v0[k] : 0.0;
    const double alpha0 = post ? (double)a.alpha0[k] : 0.0;
    const double kap = (post && smm) ? (double)a.kappa[k] : 0.0;
    bool shifted = (a.src == 0) && (a.pivot != nullptr);
    double m0d = 0.0, m0e = 0.0, C0de = 0.0, cd = 0.0, ce = 0.0;
    if (tid < D * D) {
        const int d = tid / D, e = tid % D;
        if (post) { m0d = a.m0[k * D + d]; m0e = a.m0[k * D + e]; C0de = a.C0[(k * D + d) * D + e]; }
        if (shifted) { cd = a.pivot[d]; ce = a.pivot[e]; }
    }
    if (post && tid < K) alpha0s[tid] = a.alpha0[tid];

    FIN_TS(0);
    if (a.src == 0) {
        constexpr int PX = G::PF + 1;
        const int f = tid & 63, g = tid >> 6;
        // all loads of a chunk are issued before the first add (fixed summation order: b ascending).  The rows of this
        // component are contiguous: partials[k][b][PX].  The order of the additions does not depend on the thread count: there are
        // always FIN_MAX_GROUPS LOGICAL groups (group lg sums blocks lg, lg + 16, ..), a block of fewer waves takes several each -
        // the head of the one-launch step (512 threads) is bit-identical to the stand-alone kernel (1024).
        const double* __restrict__ mine = a.partials + (long long)k * MAX_BLOCKS * PX;
        for (int lg = g; lg < FIN_MAX_GROUPS; lg += FIN_GROUPS) {
            double s = 0.0;
            for (int b0 = lg; b0 < a.nblk; b0 += FIN_MAX_GROUPS * 16) {
                double v1[16];
                // unconditional loads from clamped (always valid) addresses, masked afterwards: a load under a
                // per-element condition becomes a branch + full wait per element (32 serialised round trips)
                const int fc = f < PX ? f : PX - 1;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int b = b0 + i * FIN_MAX_GROUPS;
                    const int bc = b < a.nblk ? b : a.nblk - 1;
                    v1[i] = mine[(long long)bc * PX + fc];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool in = b0 + i * FIN_MAX_GROUPS < a.nblk;
                    v1[i] = (in && f < PX) ? v1[i] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) s += v1[i];
            }
            part[lg][f] = s;
        }
        __syncthreads();
        FIN_TS(1);
        if (tid < 64) {
            double t1 = 0.0;
            for (int gg = 0; gg < FIN_MAX_GROUPS; ++gg) t1 += part[gg][tid];
            part[0][tid] = t1;
        }
        __syncthreads();
        if (tid == 0) { st[0] = part[0][G::F]; st[1] = part[0][0]; ntot = part[0][G::PF]; }
        if (tid < D) st[2 + tid] = part[0][1 + tid];
        if (tid < D * D) {
            const int d = tid / D, e = tid % D;
            const int lo = d < e ? d : e, hi = d < e ? e : d;
            st[2 + D + tid] = part[0][1 + D + lo * D - lo * (lo - 1) / 2 + (hi - lo)];
        }
    } else {
        for (int i = tid; i < G::SW; i += nthreads) st[i] = a.stats_in[(long long)k * G::SW + i];
        if (tid == 0) {
            double t = 0.0;
            for (int j = 0; j < K; ++j) t += a.stats_in[(long long)j * G::SW];
            ntot = t;
        }
    }
    __syncthreads();
    // st holds the moments of the SHIFTED data x - c (c = pivot, 0 if none); the public layout is un-shifted:
    //   sum w x = sx' + W c,   sum w x x^T = sxx' + c sx'^T + sx' c^T + W c c^T        (fp64)
    if (a.nranks >= 1) {
        // ---- sum over ranks inside this launch (see include/vmp_hip.h, vmp_mix_finalize_exchange): this block owns
        // component k on its rank and exchanges with the blocks that own component k on the other ranks
        constexpr int SWP = (G::SW + 1 + 7) & ~7;
        const int par = (int)(a.iter & 1), G_ = a.nranks;
        double val = 0.0;
        if (tid < G::SW) {
            val = st[tid];
            if (shifted && tid >= 2) {
                const double W = st[1];
                if (tid < 2 + D) {
                    val += W * (double)a.pivot[tid - 2];
                } else {
                    const int d = (tid - 2 - D) / D, e = (tid - 2 - D) % D;
                    const double pd = a.pivot[d], pe = a.pivot[e];
                    val += pd * st[2 + e] + st[2 + d] * pe + W * pd * pe;
                }
            }
        } else if (tid == G::SW) {
            val = ntot;
        }
        if (tid <= G::SW) {
            for (int g = 0; g < G_; ++g) {
                double* slot = a.peer[g] + (((size_t)par * G_ + a.rank) * K + k) * SWP;
                __hip_atomic_store(slot + tid, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            // EVERY storing wave (SW + 1 = 75 doubles at D = 8: waves 0 and 1) releases and drains its OWN slot stores
            // before the barrier: a workgroup barrier does not wait for another wave's outstanding vector stores
            // (s_waitcnt lgkmcnt(0); s_barrier), so the publisher's vmcnt(0) below covers wave 0 only
            // (tests/test_abi.py::test_peer_exchange_release_covers_every_storing_wave checks the shipped code object)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                 // system scope: the slots are visible before the words
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int g = 0; g < G_; ++g) {
                unsigned long long* fl = reinterpret_cast<unsigned long long*>(a.peer[g] + (size_t)2 * G_ * K * SWP);
                __hip_atomic_store(fl + ((size_t)par * G_ + a.rank) * K + k, a.iter + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        double* mine = a.peer[a.rank];
        if (tid < G_) {
            const unsigned long long* fl = reinterpret_cast<const unsigned long long*>(mine + (size_t)2 * G_ * K * SWP)
                                           + ((size_t)par * G_ + tid) * K + k;
            const long long t0 = wall_clock64();                          // 100 MHz
            bool ok = false;
            while (!ok) {
                ok = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == a.iter + 1ull;
                if (!ok) {
                    if (wall_clock64() - t0 > 400000000ll) break;        // ~4 s: a peer never arrived
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            if (!ok && a.status) *a.status = 1;
        }
        __syncthreads();
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        __syncthreads();
        if (tid <= G::SW) {
            double s2 = 0.0;
            for (int g = 0; g < G_; ++g)                                  // fixed rank order: identical on every rank
                s2 += __hip_atomic_load(mine + (((size_t)par * G_ + g) * K + k) * SWP + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (tid < G::SW) st[tid] = s2; else ntot = s2;
        }
        shifted = false;                                                  // st now holds the GLOBAL moments in the caller's coordinates
        cd = 0.0; ce = 0.0;
        __syncthreads();
    }
    if (a.stats_out) {
        for (int i = tid; i < G::SW; i += nthreads) {
            double val = st[i];
            if (shifted && i >= 2) {
                const double W = st[1];
                if (i < 2 + D) {
                    val += W * (double)a.pivot[i - 2];
                } else {
                    const int d = (i - 2 - D) / D, e = (i - 2 - D) % D;
                    const double pd = a.pivot[d], pe = a.pivot[e];
                    val += pd * st[2 + e] + st[2 + d] * pe + W * pd * pe;
                }
            }
            a.stats_out[(long long)k * G::SW + i] = val;
        }
    }
    if (!a.do_post) return;

    const double Nk = st[0], Wk = smm ? st[1] : st[0];
    const double* sx = st + 2;
    const double* sxx = st + 2 + D;
    // x_k: gmm.py:30-36 (NaN -> un-normalised when N_k == 0);  smm.py:32-38 (eps = 1e-20)
    const double den = smm ? (Wk + 1e-20) : Wk;
    const bool empty = (!smm) && !(Wk != 0.0);
    const double alpha_k = alpha0 + Nk;                              // gmm.py:49-51 / smm.py:53-55
    const double beta_k = beta0 + Wk;                                // gmm.py:54-56 / smm.py:58-60
    const double v_k = smm ? (v0 + Nk) : (v0 + Nk + 1.0);            // smm.py:73-76 / gmm.py:79-81 (+1 quirk)
    FIN_TS(2);
    // ---- phase B: element (d,e) per lane
    if (tid < D * D) {
        const int d = tid / D, e = tid % D;
        const double xd = empty ? sx[d] : sx[d] / den, xe = empty ? sx[e] : sx[e] / den;
        // S_k = sum_n w (x - x_k)(x - x_k)^T / W_k from raw moments (gmm.py:39-46, smm.py:41-50)
        const double cen = sxx[d * D + e] - xd * sx[e] - sx[d] * xe + Wk * xd * xe;
        const double Sde = empty ? cen : cen / den;
        // x_k in the caller's coordinates (an empty component keeps the reference's un-normalised 0)
        const double xrd = empty ? sx[d] + Wk * cd : xd + cd, xre = empty ? sx[e] + Wk * ce : xe + ce;
        const double q0d = xrd - m0d, q0e = xre - m0e;
        const double Cde = C0de + Wk * Sde + (beta0 * Wk / beta_k) * q0d * q0e;                        // gmm.py:71-76
        Ck[d * D + e] = Cde;
        if (a.S) a.S[(k * D + d) * D + e] = (float)Sde;
        if (a.C) a.C[(k * D + d) * D + e] = (float)Cde;
        if (e == 0) {
            const double md = (beta0 * m0d + Wk * xrd) / beta_k;                     // gmm.py:59-68
            mk[d] = md;
            if (a.m) a.m[k * D + d] = (float)md;
            if (a.xbar) a.xbar[k * D + d] = (float)xrd;
        }
    }
    if (tid == 0) {
        if (a.alpha) a.alpha[k] = (float)alpha_k;
        if (a.beta) a.beta[k] = (float)beta_k;
        if (a.v) a.v[k] = (float)v_k;
    }
    __syncthreads();
    FIN_TS(3);
    // ---- phase C: factorisation (thread 0)  ||  special functions (threads 64..)
    // P_k = inv(C_k) (gmm.py:260) is never formed: with C = Lc Lc^T,
    //   v (x-m)^T P (x-m) = || sqrt(v) Lc^{-1} (x-m) ||^2   and   log det P = -2 sum log diag Lc.
    if (tid < 64) {
        double X[D], sumlog;
        bool ok;
        wave_chol_inverse<D>(Ck, tid, X, sumlog, ok);
        const double sv = ok ? sqrt(v_k) : nan("");
        if (a.pack && tid < D) {
            const int p = k * G::PACK + D;
#pragma unroll
            for (int i = 0; i < D; ++i)
                if (i >= tid) st_pack(a, p + i * (i + 1) / 2 + tid, X[i] * sv);      // W = sqrt(v) L^{-1}, lower
        }
        if (tid == 0) { scal[0] = -2.0 * sumlog; scal[1] = ok ? 1.0 : 0.0; }
    } else if (tid >= 64 && tid < 64 + D + 2) {
        // all digammas in ONE wave and ONE code path (different branches of a wave run one after the other: four
        // special-function evaluations in sequence were most of this phase)
        const int i = tid - 64;
        double arg = 0.5 * (v_k + (smm ? 0.0 : 1.0) + i);             // gmm.py:128-129 / smm.py:107-108
        if (i == D) arg = alpha_k;
        if (i == D + 1) {
            double asum = 0.0;
            for (int j = 0; j < K; ++j) asum += alpha0s[j];
            arg = asum + ntot;
        }
        sp[i] = digamma_d(arg);
    } else if (tid == 128 && smm) {
        sp[D + 2] = lgamma(0.5 * (D + kap)) - lgamma(0.5 * kap);
    }
    __syncthreads();
    FIN_TS(4);
    // ---- phase D
    if (tid == 0) {
        const double LOG2 = 0.69314718055994530942, PI = 3.14159265358979323846;
        const double elp = sp[D] - sp[D + 1];
        double sdg = 0.0;
        for (int i = 0; i < D; ++i) sdg += sp[i];
        double c, h = 0.5, ua = 1.0, ub = 1.0;
        const double logdetP = scal[0];
        if (!smm) {
            // gmm.py:120-121: log det P replaced by 0 when det P <= 1e-20
            const double ld = (logdetP > log(1e-20)) ? logdetP : 0.0;
            c = elp + 0.5 * (sdg + D * LOG2 + ld) - 0.5 * (D / beta_k);
        } else {
            h = 0.5 * (D + kap);
            // smm.py:122-124 (note the precedence of line 124: ... - (0.5 (D+kappa) m - log kappa))
            c = sp[D + 2] - 0.5 * D * log(kap * PI) + elp + 0.5 * (sdg + D * LOG2 + logdetP) - h * (D / beta_k) + log(kap);
            ua = D + kap;                                                                   // smm.py:134-137
            ub = D / beta_k + kap;
        }
        if (scal[1] == 0.0) c = nan("");
        if (a.pi) a.pi[k] = (float)exp(elp);
        if (a.pack) {
            const int p = k * G::PACK + D + G::TRI;
            const double LOG2E = 1.4426950408889634074;  // the pass kernel evaluates 2^(c - h q)
            st_pack(a, p, c * LOG2E); st_pack(a, p + 1, h * LOG2E); st_pack(a, p + 2, ua); st_pack(a, p + 3, ub);
        }
    }
    if (a.pack && tid < D) st_pack(a, k * G::PACK + tid, mk[tid]);
    FIN_TS(5);
}

template <int D>
__global__ __launch_bounds__(FIN_THREADS) void finalize_kernel(FinArgs a) {
    finalize_block<D>(a, blockIdx.x, threadIdx.x, FIN_THREADS);
}


// E-step pack from explicit (alpha, beta, m, P, v): gmm.e_step / smm.e_step signature.
struct PackArgs {
    int K, flavour;
    const float *alpha, *beta, *m, *P, *v, *kappa;
    float *pack, *pi;
};

template <int D>
__global__ void pack_kernel(PackArgs a) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.K) return;
    double asum = 0.0;
    for (int j = 0; j < a.K; ++j) asum += a.alpha[j];
    const double v_k = a.v[k];
    // v P = R R^T (lower R);  q = || R^T d ||^2.  The kernel wants a LOWER-triangular W with q = ||W d||^2:
    // factor the reversed matrix J P J = U_r U_r^T ... simpler: upper-Cholesky via reversal permutation.
    double A[D * D], mk[D];
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            const int ri = D - 1 - i, rj = D - 1 - j;               // reversal: A = J (sym P) J
            A[i * D + j] = 0.5 * ((double)a.P[(k * D + ri) * D + rj] + (double)a.P[(k * D + rj) * D + ri]);
        }
    double c = 0, h = 0.5, ua = 1, ub = 1, elp = 0;
    double W[D * D];
    if (chol_lower<D>(A)) {
        // A = L L^T  =>  P = (J L J)(J L J)^T with J L J upper-triangular U; q = d^T U U^T d = ||U^T d||^2,
        // and U^T = J L^T J is LOWER triangular.
        double ld = 0.0;
        for (int i = 0; i < D; ++i) ld += log(A[i * D + i]);
        const double sv = sqrt(v_k);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) W[i * D + j] = sv * A[(D - 1 - j) * D + (D - 1 - i)];
        const double kap = a.flavour == VMP_SMM ? (double)a.kappa[k] : 0.0;
        estep_constants<D>(k, a.flavour, a.alpha[k], asum, a.beta[k], v_k, 2.0 * ld, kap, c, h, ua, ub, elp);
    } else {
        for (int i = 0; i < D * D; ++i) W[i] = nan("");
        c = nan("");
    }
    for (int d = 0; d < D; ++d) mk[d] = a.m[k * D + d];
    if (a.pi) a.pi[k] = (float)exp(elp);
    write_pack<D>(a.pack, k, mk, W, c, h, ua, ub);
}


// Pivot for the moment accumulation: mean of up to 4096 evenly strided rows of x (deterministic, one block).
// Any fixed vector is a valid pivot (the finalize kernel un-shifts exactly in fp64); a vector near the data mean
// keeps the fp32 products x_d x_e small, which is what makes raw-moment accumulation as accurate as the
// reference's two-pass centred form (gmm.py:39-46).
struct PivotArgs { const float* x; long long N; int D; float* out; };

__global__ __launch_bounds__(1024) void pivot_kernel(PivotArgs a) {
    __shared__ double red[1024];
    const int tid = threadIdx.x;
    const long long S = a.N < 4096 ? a.N : 4096;
    const long long step = a.N / S;
    for (int d = 0; d < a.D; ++d) {
        double s = 0.0;
        for (long long i = tid; i < S; i += 1024) s += (double)a.x[(i * step) * a.D + d];
        red[tid] = s;
        __syncthreads();
        for (int w = 512; w > 0; w >>= 1) {
            if (tid < w) red[tid] += red[tid + w];
            __syncthreads();
        }
        if (tid == 0) a.out[d] = (float)(red[0] / (double)S);
        __syncthreads();
    }
}

// Raw moments of a SMALL batch: small_stats_component (vmp_common.h), one block per component.
__global__ __launch_bounds__(SMALL_STATS_GROUPS * 80) void small_stats_kernel(SmallStatsArgs a) {
    __shared__ double part[SMALL_STATS_GROUPS][80];
    const int SW = 2 + a.D + a.D * a.D, i = threadIdx.x % 80;
    const double t = small_stats_component(a, blockIdx.x, part);
    if (threadIdx.x < 80 && i < SW) a.stats[(long long)blockIdx.x * SW + i] = t;
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
struct Plan {
    int nw, blocks, par_reduce;
    size_t lds;
    long long rpw, rpw_b;
};

Plan make_plan(long long N, int D, int K, int flavour, bool stats, bool xdl = false) {
    Plan p;
    constexpr int tuned_blocks = 256;      // one 8-wave block per CU
    constexpr int tuned_nw = MAX_NW;
    const size_t wreg = (size_t)((D + 2) * LS + (xdl ? TR * AIS : 0)) * sizeof(float);
    int nw = tuned_nw;
    if (nw > max_nw((K + 15) / 16)) nw = max_nw((K + 15) / 16);
    if (nw < 1) nw = 1;
    const long long ntiles = (N + TR - 1) / TR;
    if ((long long)nw > ntiles) nw = (int)ntiles;
    long long blocks = (ntiles + nw - 1) / nw;
    if (blocks > tuned_blocks) blocks = tuned_blocks;
    if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
    // equal contiguous row ranges, at least one full tile each
    long long rpw = (N + blocks * nw - 1) / (blocks * nw);
    rpw = (rpw + 7) / 8 * 8;
    if (rpw < TR) rpw = TR;
    blocks = ((N + rpw - 1) / rpw + nw - 1) / nw;
    p.rpw_b = rpw;
    // Two waves share a SIMD (w and w + 4 of an 8-wave block) and the sequencer serves the OLDER one first: with equal
    // shares, clock64 stamps (tools/pass_ts.py) show waves 0-3 done at 76 k cycles and waves 4-7 at 107 k - the last 30 %
    // of the kernel runs one wave per SIMD with nothing to hide its latencies behind (s_setprio does not change it).  The
    // older waves therefore get the larger share, so that both mates finish together.  Ranges stay contiguous and fixed:
    // results remain deterministic.
#ifndef VMP_SPLIT_GMM
#define VMP_SPLIT_GMM 64
#endif
#ifndef VMP_SPLIT_SMM
#define VMP_SPLIT_SMM 62
#endif
    const int split = flavour == VMP_SMM ? VMP_SPLIT_SMM : VMP_SPLIT_GMM;   // % of a pair's rows for the older wave: measured optima (N = 3e5 .. 1e7)
    if (nw == 8 && split != 50 && rpw >= 2 * TR) {
        long long cap = tuned_blocks < MAX_BLOCKS ? tuned_blocks : MAX_BLOCKS;
        long long pr = ((N + 4 * cap - 1) / (4 * cap) + 7) / 8 * 8;           // rows of a SIMD pair, all blocks in use
        if (pr < 2 * TR) pr = 2 * TR;
        const long long ra = (pr * split / 100 + 4) / 8 * 8, rb = pr - ra;
        if (rb >= TR) {
            rpw = ra;
            p.rpw_b = rb;
            blocks = (N + 4 * pr - 1) / (4 * pr);
        }
    }
    p.rpw = rpw;
    p.nw = nw;
    p.blocks = (int)blocks;
    const int FTn = (1 + D + D * (D + 1) / 2 + 15) / 16, KTn = (K + 15) / 16;
    const size_t scratch = stats ? (size_t)(KTn <= 2 ? KTn : 4) * (FTn + 1) * 4 * WAVE * sizeof(double) : 0;
    p.lds = wreg * nw;
    p.par_reduce = 0;
    if (scratch * nw <= 64 * 1024) {          // one slab per wave fits: parallel block reduction
        p.par_reduce = 1;
        if (scratch * nw > p.lds) p.lds = scratch * nw;
    } else if (scratch > p.lds) {
        p.lds = scratch;
    }
    return p;
}

template <int D, int KT>
int launch_pass_dk(const PassArgs& a, const Plan& p, int flavour, bool estep, bool stats, bool mask, hipStream_t s) {
    dim3 grid(p.blocks), block(p.nw * WAVE);
#define VMP_LAUNCH(FL, E, S, M) \
    hipLaunchKernelGGL((pass_kernel<D, KT, FL, E, S, M>), grid, block, p.lds, s, a)
    if (flavour == VMP_GMM) {
        if (estep && stats) VMP_LAUNCH(VMP_GMM, true, true, false);
        else if (estep && mask) VMP_LAUNCH(VMP_GMM, true, false, true);
        else if (estep) VMP_LAUNCH(VMP_GMM, true, false, false);
        else VMP_LAUNCH(VMP_GMM, false, true, false);
    } else {
        if (estep && stats) VMP_LAUNCH(VMP_SMM, true, true, false);
        else if (estep) VMP_LAUNCH(VMP_SMM, true, false, false);
        else VMP_LAUNCH(VMP_SMM, false, true, false);
    }
#undef VMP_LAUNCH
    return check_launch("pass_kernel");
}

#ifndef VMP_T1_XDL
#define VMP_T1_XDL 1          // 0: build without the XDL E-part (A/B measurements: tools/build_variant.sh)
#endif
// E-part on the XDL pipe: E-step launches with K <= 16 and no missing-data mask
inline bool use_xdl(int K, bool estep, bool mask) { return VMP_T1_XDL && estep && !mask && K <= 16; }

#ifndef VMP_MOM2_ROWS
#define VMP_MOM2_ROWS (1ll << 16)      // rows from which the moment GEMM of the XDL pass multiplies 2-term operands (pass_xdl_body, MT)
#endif
template <int D>
int launch_pass_xdl(const PassArgs& a, const Plan& p, int flavour, bool stats, hipStream_t s) {
    dim3 grid(p.blocks), block(p.nw * WAVE);
#define VMP_LAUNCH_X(FL, S, M) do { \
        if (p.lds > 64 * 1024) { if (const int rc_ = set_dyn_lds(reinterpret_cast<const void*>(pass_xdl_kernel<D, FL, S, M>), p.lds, "pass_xdl_kernel")) return rc_; } \
        hipLaunchKernelGGL((pass_xdl_kernel<D, FL, S, M>), grid, block, p.lds, s, a); } while (0)
    const bool m2 = stats && a.N >= VMP_MOM2_ROWS;
    if (flavour == VMP_GMM) { if (m2) VMP_LAUNCH_X(VMP_GMM, true, 2); else if (stats) VMP_LAUNCH_X(VMP_GMM, true, 3); else VMP_LAUNCH_X(VMP_GMM, false, 3); }
    else { if (m2) VMP_LAUNCH_X(VMP_SMM, true, 2); else if (stats) VMP_LAUNCH_X(VMP_SMM, true, 3); else VMP_LAUNCH_X(VMP_SMM, false, 3); }
#undef VMP_LAUNCH_X
    return check_launch("pass_xdl_kernel");
}

template <int D>
int launch_pass_d(const PassArgs& a, const Plan& p, int flavour, bool estep, bool stats, bool mask, hipStream_t s) {
    if (use_xdl(a.K, estep, mask)) return launch_pass_xdl<D>(a, p, flavour, stats, s);
    const int KT = (a.K + 15) / 16;
    if (KT == 1) return launch_pass_dk<D, 1>(a, p, flavour, estep, stats, mask, s);
    if (KT == 2) return launch_pass_dk<D, 2>(a, p, flavour, estep, stats, mask, s);
    return launch_pass_dk<D, 4>(a, p, flavour, estep, stats, mask, s);
}

#define VMP_DISPATCH_D(D, CALL)            \
    switch (D) {                            \
        case 1: { constexpr int DD = 1; CALL; } break; \
        case 2: { constexpr int DD = 2; CALL; } break; \
        case 3: { constexpr int DD = 3; CALL; } break; \
        case 4: { constexpr int DD = 4; CALL; } break; \
        case 5: { constexpr int DD = 5; CALL; } break; \
        case 6: { constexpr int DD = 6; CALL; } break; \
        case 7: { constexpr int DD = 7; CALL; } break; \
        case 8: { constexpr int DD = 8; CALL; } break; \
        default: break;                     \
    }

int check_dims(int64_t N, int D, int K) {
    if (N <= 0) { set_error("N must be positive (got %lld)", (long long)N); return VMP_E_BADARG; }
    if (D < 1 || D > VMP_MAX_D) { set_error("D=%d outside compiled range 1..%d", D, VMP_MAX_D); return VMP_E_DIM; }
    if (K < 1 || K > VMP_MAX_K) { set_error("K=%d outside compiled range 1..%d", K, VMP_MAX_K); return VMP_E_DIM; }
    return 0;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#ifdef VMP_DEBUG_TS
static long long* g_dbg_pass = nullptr;
static long long* g_dbg_t = nullptr;
#endif
int run_pass(PassArgs a, int D, int flavour, bool estep, bool stats, bool mask, hipStream_t s) {
    Plan p = make_plan(a.N, D, a.K, flavour, stats, use_xdl(a.K, estep, mask));
    a.rpw = p.rpw;
    a.rpw_b = p.rpw_b;
    a.par_reduce = p.par_reduce;
#ifdef VMP_DEBUG_TS
    a.dbg_t = g_dbg_pass;
#endif

    int rc = -1;
    VMP_DISPATCH_D(D, rc = launch_pass_d<DD>(a, p, flavour, estep, stats, mask, s));
    return rc;
}

int run_finalize(FinArgs f, int D, hipStream_t s) {
    int rc = -1;
#ifdef VMP_DEBUG_TS
    f.dbg_t = g_dbg_t;
#endif

    VMP_DISPATCH_D(D, {
        hipLaunchKernelGGL((finalize_kernel<DD>), dim3(f.K), dim3(FIN_THREADS), 0, s, f);
        rc = check_launch("finalize_kernel");
    });
    return rc;
}

// ---------------------------------------------------------------------------------------------------------
// Opt-in ACCURATE E-part (round 6): the whole cell arithmetic in fp64, from an fp64 copy of the pack.
// Why it exists: the SMM's log rho_nk = c_k - (D + kappa)/2 * q_nk (smm.py:119-128, linear in the expected Mahalanobis distance q with
// a factor 6.5 at D = 8, kappa = 5) reaches 1e2..1e3 for rows without a close component; an fp32 q carries an absolute error of
// 1e-7 q, i.e. up to 6e-5 in log rho and 1..4e-5 in r_nk (same-input r at C5: 9e-6 / 8e-6 / 4e-5), and an fp32 log rho cannot
// even represent the difference.  Here d = x - m, y = W d, q = |y|^2, log2 rho = c' - h' q, the row max, the exponentials and their
// sum are fp64; r (and u = ua / (q + ub), smm.py:131-137) are rounded once, on the way out.  One thread per data row, the K pack
// rows in LDS; two sweeps over k (max, then normalise) recompute q instead of holding K doubles per thread.
// ---------------------------------------------------------------------------------------------------------
struct AccArgs { const float* x; const double* pack; float* r; float* u; float* logr; long long N; int K; };
template <int D, bool SMM>
__global__ __launch_bounds__(256) void estep_f64_kernel(AccArgs a) {
    using G = Geo<D>;
    extern __shared__ double spk[];                          // [K][PACK]
    for (int e = threadIdx.x; e < a.K * G::PACK; e += blockDim.x) spk[e] = a.pack[e];
    __syncthreads();
    const double LN2 = 0.69314718055994530942;
    for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < a.N; n += (long long)gridDim.x * blockDim.x) {
        double xd[D];
#pragma unroll
        for (int j = 0; j < D; ++j) xd[j] = (double)a.x[n * D + j];
        auto quad = [&](int k) {
            const double* __restrict__ p = spk + k * G::PACK;
            double q = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double y = 0.0;
#pragma unroll
                for (int j = 0; j <= i; ++j) y = fma(p[D + i * (i + 1) / 2 + j], xd[j] - p[j], y);
                q = fma(y, y, q);
            }
            return q;
        };
        double mx = -INFINITY;
        for (int k = 0; k < a.K; ++k) {
            const double* __restrict__ p = spk + k * G::PACK;
            const double l2 = p[D + G::TRI] - p[D + G::TRI + 1] * quad(k);
            mx = l2 > mx ? l2 : mx;                           // (NaN packs - a failed factorisation - propagate through the sum below)
        }
        double ssum = 0.0;
        for (int k = 0; k < a.K; ++k) {
            const double* __restrict__ p = spk + k * G::PACK;
            ssum += exp2(p[D + G::TRI] - p[D + G::TRI + 1] * quad(k) - mx);
        }
        const double inv = 1.0 / ssum, l2s = log2(ssum);
        for (int k = 0; k < a.K; ++k) {
            const double* __restrict__ p = spk + k * G::PACK;
            const double q = quad(k);
            const double l2 = p[D + G::TRI] - p[D + G::TRI + 1] * q - mx;
            a.r[n * a.K + k] = (float)(exp2(l2) * inv);
            if (a.logr) a.logr[n * a.K + k] = (float)((l2 - l2s) * LN2);
            if constexpr (SMM) a.u[n * a.K + k] = (float)(p[D + G::TRI + 2] / (q + p[D + G::TRI + 3]));
        }
    }
}

// fp64 M-pass of the accurate mode: sum_n w_nk [1 | x' | x' x'^T (upper)] and N_k = sum_n r_nk with every product and every sum in
// fp64 (x' = x - pivot, w = r (GMM) / r u (SMM): products of fp32 values are exact in fp64), into the same per-block partial rows
// the fused pass leaves for finalize_kernel.  Why: the SMM's log rho = c - 6.5 q with q up to 1e2..1e3 turns a 5e-8 relative
// error of P_k (the fp32-product moments' level) into 1e-4 on log rho of rows without a close component.
// Block b owns a contiguous row range; thread t owns the (component, feature) items t, t + blockDim, ..; rows are staged
// through LDS RS at a time.
constexpr int ACC_RS = 128, ACC_THREADS = 1024, ACC_ITEMS = 3;   // 64 components x 46 features <= 3 x 1024
struct AccStatArgs { const float *x, *r, *u, *pivot; double* partials; long long N, rows_per_block; int K; };
template <int D>
__global__ __launch_bounds__(ACC_THREADS) void stats_f64_kernel(AccStatArgs a) {
    using G = Geo<D>;
    constexpr int FP = G::F + 1;                             // features + the N_k column
    constexpr int XS = D + 1;                                // staged row: x' and a constant 1
    extern __shared__ float sm[];
    const int K = a.K;
    float* xs = sm;                                          // [RS][XS]
    float* ws_ = sm + ACC_RS * XS;                           // [RS][K]  w = r (u)
    float* us_ = ws_ + ACC_RS * K;                           // [RS][K]  u (SMM) - the N_k column needs r alone
    const bool smm = a.u != nullptr;
    int ik[ACC_ITEMS], ia[ACC_ITEMS], ib[ACC_ITEMS], isn[ACC_ITEMS];
    double acc[ACC_ITEMS];
#pragma unroll
    for (int q = 0; q < ACC_ITEMS; ++q) {
        const int it = (int)threadIdx.x + q * ACC_THREADS;
        const bool on = it < K * FP;
        const int k = on ? it / FP : 0, f = on ? it - k * FP : 0;
        ik[q] = on ? k : -1; isn[q] = (f == G::F);
        int a_ = D, b_ = D;                                  // f = 0 (W_k) and f = F (N_k): feature 1 * 1
        if (f >= 1 && f <= D) a_ = f - 1;
        else if (f > D && f < G::F) {
            int lo = 0, rem = f - 1 - D;
            while (rem >= D - lo) { rem -= D - lo; ++lo; }
            a_ = lo; b_ = lo + rem;
        }
        ia[q] = a_; ib[q] = b_; acc[q] = 0.0;
    }
    const long long lo_row = (long long)blockIdx.x * a.rows_per_block;
    long long hi_row = lo_row + a.rows_per_block;
    if (hi_row > a.N) hi_row = a.N;
    for (long long base = lo_row; base < hi_row; base += ACC_RS) {
        const int nr = (int)((hi_row - base) < ACC_RS ? (hi_row - base) : ACC_RS);
        __syncthreads();
        for (int e = threadIdx.x; e < nr * XS; e += blockDim.x) {
            const int n = e / XS, d = e - n * XS;
            xs[e] = d < D ? a.x[(base + n) * D + d] - (a.pivot ? a.pivot[d] : 0.f) : 1.f;    // (x - pivot in fp32, as the fused pass shifts it;
        }                                                                                   //  finalize un-shifts with the same fp32 pivot in fp64)
        for (int e = threadIdx.x; e < nr * K; e += blockDim.x) {
            ws_[e] = a.r[base * K + e];
            if (smm) us_[e] = a.u[base * K + e];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < ACC_ITEMS; ++q) {
            if (ik[q] < 0) continue;
            const int k = ik[q];
            double s = 0.0;
            for (int n = 0; n < nr; ++n) {
                const double rr = (double)ws_[n * K + k];
                const double w = (smm && !isn[q]) ? rr * (double)us_[n * K + k] : rr;
                s = fma(w, (double)xs[n * XS + ia[q]] * (double)xs[n * XS + ib[q]], s);
            }
            acc[q] += s;
        }
    }
    // N_k of this block -> LDS, block total in a fixed order; then the partial rows
    __syncthreads();
    double* nk = reinterpret_cast<double*>(sm);              // [K] (+1: total)
#pragma unroll
    for (int q = 0; q < ACC_ITEMS; ++q)
        if (ik[q] >= 0 && isn[q]) nk[ik[q]] = acc[q];
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < K; ++k) t += nk[k];
        nk[K] = t;
    }
    __syncthreads();
    constexpr int PX = G::PF + 1;
#pragma unroll
    for (int q = 0; q < ACC_ITEMS; ++q) {
        if (ik[q] < 0) continue;
        const int it = (int)threadIdx.x + q * ACC_THREADS, f = it - ik[q] * FP;
        double* row = a.partials + ((long long)ik[q] * MAX_BLOCKS + blockIdx.x) * PX;
        row[f] = acc[q];                                     // f = F is the N_k slot
        if (f == 0) row[G::PF] = nk[K];
    }
}

// workspace layout: [per-block partials | reserved words (zeroed by vmp_mix_stats_ws) | status word]
constexpr size_t WS_TPACK_WORDS = 16 * (VMP_MAX_D + VMP_MAX_D * (VMP_MAX_D + 1) / 2 + 4);
inline size_t ws_partial_bytes(int D, int K) { return (size_t)MAX_BLOCKS * K * (partial_words(D) + 1) * sizeof(double); }
inline unsigned long long* ws_seq(void* ws, int D, int K) { return reinterpret_cast<unsigned long long*>(static_cast<char*>(ws) + ws_partial_bytes(D, K)); }

}  // namespace

// =========================================================================================================
// C ABI
// =========================================================================================================
extern "C" {

#ifdef VMP_DEBUG_TS
void vmp_debug_set_finalize_timestamps(long long* p) { g_dbg_t = p; }     // exploration builds only (tools/pass_ts.py)
void vmp_debug_set_pass_timestamps(long long* p) { g_dbg_pass = p; }
#endif

int vmp_mix_pack_words(int D) { return pack_words(D); }
int vmp_mix_stats_words(int D) { return stats_words(D); }

size_t vmp_mix_workspace_bytes(int64_t N, int D, int K) {
    (void)N;
    // [K][MAX_BLOCKS][PF + 1] per-block partials | reserved words + status
    return ws_partial_bytes(D, K) + (WS_TPACK_WORDS + 2) * sizeof(unsigned long long);
}

int vmp_mix_pivot(const float* x, int64_t N, int D, float* pivot_out, void* stream) {
    int rc = check_dims(N, D, 1);
    if (rc) return rc;
    if (!x || !pivot_out) { set_error("vmp_mix_pivot: null pointer"); return VMP_E_BADARG; }
    PivotArgs a{x, N, D, pivot_out};
    hipLaunchKernelGGL(pivot_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("pivot_kernel");
}

int vmp_mix_stats(const float* x, const float* r, const float* u, const float* pivot, int64_t N, int D, int K,
                  double* stats, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_dims(N, D, K);
    if (rc) return rc;
    if (!x || !r || !stats || !ws) { set_error("vmp_mix_stats: null pointer"); return VMP_E_BADARG; }
    if (ws_bytes < vmp_mix_workspace_bytes(N, D, K)) { set_error("vmp_mix_stats: workspace too small"); return VMP_E_WS; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N <= SMALL_STATS_MAX_N && !pivot) {                 // small batch, no shift requested: direct fp64 sums, one launch
        SmallStatsArgs sa{x, r, u, stats, (int)N, D, K};
        hipLaunchKernelGGL(small_stats_kernel, dim3(K), dim3(SMALL_STATS_GROUPS * 80), 0, s, sa);
        return check_launch("small_stats_kernel");
    }
    const int flavour = u ? VMP_SMM : VMP_GMM;
    PassArgs a{};
    a.x = x; a.r_in = r; a.u_in = u; a.pivot = pivot; a.N = N; a.K = K; a.partials = static_cast<double*>(ws);
    a.vec_ok = aligned16(x) && aligned16(r) && (!u || aligned16(u));
    rc = run_pass(a, D, flavour, false, true, false, s);
    if (rc) return rc;
    FinArgs f{};
    f.partials = a.partials; f.nblk = make_plan(N, D, K, flavour, true).blocks; f.K = K; f.flavour = flavour;
    f.src = 0; f.do_post = 0; f.stats_out = stats; f.pivot = pivot;
    return run_finalize(f, D, s);
}

int vmp_mix_finalize(const double* stats, int D, int K, int flavour, const float* alpha0, const float* beta0,
                     const float* m0, const float* C0, const float* v0, const float* kappa, float* alpha, float* beta,
                     float* m, float* C, float* v, float* xbar, float* S, float* pi, float* pack, void* stream) {
    int rc = check_dims(1, D, K);
    if (rc) return rc;
    if (!stats || !alpha0 || !beta0 || !m0 || !C0 || !v0) { set_error("vmp_mix_finalize: null pointer"); return VMP_E_BADARG; }
    if (flavour == VMP_SMM && !kappa) { set_error("vmp_mix_finalize: SMM needs kappa"); return VMP_E_BADARG; }
    FinArgs f{};
    f.stats_in = stats; f.nblk = 0; f.K = K; f.flavour = flavour; f.src = 1; f.do_post = 1;
    f.alpha0 = alpha0; f.beta0 = beta0; f.m0 = m0; f.C0 = C0; f.v0 = v0; f.kappa = kappa;
    f.alpha = alpha; f.beta = beta; f.m = m; f.C = C; f.v = v; f.xbar = xbar; f.S = S; f.pi = pi; f.pack = pack;
    return run_finalize(f, D, static_cast<hipStream_t>(stream));
}

int vmp_mix_pack_from_params(int D, int K, int flavour, const float* alpha, const float* beta, const float* m,
                             const float* P, const float* v, const float* kappa, float* pack, float* pi, void* stream) {
    int rc = check_dims(1, D, K);
    if (rc) return rc;
    if (!alpha || !beta || !m || !P || !v || !pack) { set_error("vmp_mix_pack_from_params: null pointer"); return VMP_E_BADARG; }
    if (flavour == VMP_SMM && !kappa) { set_error("vmp_mix_pack_from_params: SMM needs kappa"); return VMP_E_BADARG; }
    PackArgs a{K, flavour, alpha, beta, m, P, v, kappa, pack, pi};
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = -1;
    VMP_DISPATCH_D(D, {
        hipLaunchKernelGGL((pack_kernel<DD>), dim3((K + 63) / 64), dim3(64), 0, s, a);
        rc = check_launch("pack_kernel");
    });
    return rc;
}

int vmp_mix_estep(const float* x, int64_t N, int D, int K, int flavour, const float* pack, const uint8_t* miss_mask,
                  float* r_out, float* u_out, float* logr_out, const float* pivot, double* stats_out, void* ws,
                  size_t ws_bytes, void* stream) {
    int rc = check_dims(N, D, K);
    if (rc) return rc;
    if (!x || !pack || !r_out) { set_error("vmp_mix_estep: null pointer"); return VMP_E_BADARG; }
    if (flavour == VMP_SMM && !u_out) { set_error("vmp_mix_estep: SMM needs u_out"); return VMP_E_BADARG; }
    if (flavour != VMP_GMM && flavour != VMP_SMM) { set_error("vmp_mix_estep: bad flavour %d", flavour); return VMP_E_BADARG; }
    if (miss_mask && (flavour != VMP_GMM || stats_out)) {
        set_error("vmp_mix_estep: miss_mask only with GMM flavour and without fused stats");
        return VMP_E_BADARG;
    }
    if (stats_out && (!ws || ws_bytes < vmp_mix_workspace_bytes(N, D, K))) {
        set_error("vmp_mix_estep: workspace too small for fused stats");
        return VMP_E_WS;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    PassArgs a{};
    a.x = x; a.mask = miss_mask; a.pack = pack; a.r_out = r_out; a.u_out = u_out; a.logr_out = logr_out;
    a.pivot = pivot; a.N = N; a.K = K; a.partials = static_cast<double*>(ws);
    a.vec_ok = aligned16(x) && aligned16(r_out) && (!u_out || aligned16(u_out)) && (!logr_out || aligned16(logr_out));
    rc = run_pass(a, D, flavour, true, stats_out != nullptr, miss_mask != nullptr, s);
    if (rc || !stats_out) return rc;
    FinArgs f{};
    f.partials = a.partials; f.nblk = make_plan(N, D, K, flavour, true).blocks; f.K = K; f.flavour = flavour;
    f.src = 0; f.do_post = 0; f.stats_out = stats_out; f.pivot = pivot;
    return run_finalize(f, D, s);
}

int vmp_mix_estep_fused(const float* x, int64_t N, int D, int K, int flavour, const float* pack, float* r_out,
                        float* u_out, float* logr_out, const float* pivot, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_dims(N, D, K);
    if (rc) return rc;
    if (!x || !pack || !r_out || !ws) { set_error("vmp_mix_estep_fused: null pointer"); return VMP_E_BADARG; }
    if (flavour != VMP_GMM && flavour != VMP_SMM) { set_error("vmp_mix_estep_fused: bad flavour %d", flavour); return VMP_E_BADARG; }
    if (flavour == VMP_SMM && !u_out) { set_error("vmp_mix_estep_fused: SMM needs u_out"); return VMP_E_BADARG; }
    if (ws_bytes < vmp_mix_workspace_bytes(N, D, K)) { set_error("vmp_mix_estep_fused: workspace too small"); return VMP_E_WS; }
    PassArgs a{};
    a.x = x; a.pack = pack; a.r_out = r_out; a.u_out = u_out; a.logr_out = logr_out; a.pivot = pivot;
    a.N = N; a.K = K; a.partials = static_cast<double*>(ws);
    a.vec_ok = aligned16(x) && aligned16(r_out) && (!u_out || aligned16(u_out)) && (!logr_out || aligned16(logr_out));
    return run_pass(a, D, flavour, true, true, false, static_cast<hipStream_t>(stream));
}

int vmp_mix_stats_ws(const float* x, const float* r, const float* u, const float* pivot, int64_t N, int D, int K,
                     void* ws, size_t ws_bytes, void* stream) {
    int rc = check_dims(N, D, K);
    if (rc) return rc;
    if (!x || !r || !ws) { set_error("vmp_mix_stats_ws: null pointer"); return VMP_E_BADARG; }
    if (ws_bytes < vmp_mix_workspace_bytes(N, D, K)) { set_error("vmp_mix_stats_ws: workspace too small"); return VMP_E_WS; }
    PassArgs a{};
    a.x = x; a.r_in = r; a.u_in = u; a.pivot = pivot; a.N = N; a.K = K; a.partials = static_cast<double*>(ws);
    a.vec_ok = aligned16(x) && aligned16(r) && (!u || aligned16(u));
    // this call seeds the workspace of an iteration loop: the reserved words and the status word start at zero
    hipError_t e = hipMemsetAsync(ws_seq(ws, D, K), 0, (WS_TPACK_WORDS + 2) * sizeof(unsigned long long), static_cast<hipStream_t>(stream));
    if (e != hipSuccess) { set_error("vmp_mix_stats_ws: hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
    return run_pass(a, D, u ? VMP_SMM : VMP_GMM, false, true, false, static_cast<hipStream_t>(stream));
}

int vmp_mix_stats_ws_accurate(const float* x, const float* r, const float* u, const float* pivot, int64_t N, int D, int K,
                              void* ws, size_t ws_bytes, void* stream) {
    int rc = check_dims(N, D, K);
    if (rc) return rc;
    if (!x || !r || !ws) { set_error("vmp_mix_stats_ws_accurate: null pointer"); return VMP_E_BADARG; }
    if (ws_bytes < vmp_mix_workspace_bytes(N, D, K)) { set_error("vmp_mix_stats_ws_accurate: workspace too small"); return VMP_E_WS; }
    // the block count finalize_kernel will read (vmp_mix_finalize_ws / _ws64 derive it from the same plan)
    const int blocks = make_plan(N, D, K, u ? VMP_SMM : VMP_GMM, true).blocks;
    AccStatArgs a{x, r, u, pivot, static_cast<double*>(ws), N, (N + blocks - 1) / blocks, K};
    const size_t lds = (size_t)ACC_RS * (D + 1 + 2 * K) * sizeof(float);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(ws_seq(ws, D, K), 0, (WS_TPACK_WORDS + 2) * sizeof(unsigned long long), s);
    if (e != hipSuccess) { set_error("vmp_mix_stats_ws_accurate: hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
    rc = -1;
    VMP_DISPATCH_D(D, {
        if (lds > 48 * 1024) { if ((rc = set_dyn_lds(reinterpret_cast<const void*>(stats_f64_kernel<DD>), lds, "stats_f64_kernel")) != 0) return rc; }
        hipLaunchKernelGGL((stats_f64_kernel<DD>), dim3(blocks), dim3(ACC_THREADS), lds, s, a);
        rc = check_launch("stats_f64_kernel");
    });
    return rc;
}

int vmp_mix_finalize_exchange(const void* ws, const float* pivot, int64_t N, int D, int K, int flavour, const float* alpha0,
                              const float* beta0, const float* m0, const float* C0, const float* v0, const float* kappa,
                              float* alpha, float* beta, float* m, float* C, float* v, float* xbar, float* S, float* pi,
                              float* pack, double* stats_out, void* const* peers, int nranks, int rank,
                              unsigned long long iteration, int* status, void* stream) {
    int rc = check_dims(N, D, K);
    if (rc) return rc;
    if (!ws || !alpha0 || !beta0 || !m0 || !C0 || !v0 || !peers) { set_error("vmp_mix_finalize_exchange: null pointer"); return VMP_E_BADARG; }
    if (nranks < 1 || nranks > VMP_EXCH_MAX_RANKS || rank < 0 || rank >= nranks) { set_error("vmp_mix_finalize_exchange: bad rank / nranks"); return VMP_E_BADARG; }
    if (flavour == VMP_SMM && !kappa) { set_error("vmp_mix_finalize_exchange: SMM needs kappa"); return VMP_E_BADARG; }
    FinArgs f{};
    f.partials = static_cast<const double*>(ws);
    f.nblk = make_plan(N, D, K, flavour, true).blocks;
    f.K = K; f.flavour = flavour; f.src = 0;
    f.do_post = (alpha || beta || m || C || v || xbar || S || pi || pack) ? 1 : 0;
    f.alpha0 = alpha0; f.beta0 = beta0; f.m0 = m0; f.C0 = C0; f.v0 = v0; f.kappa = kappa;
    f.alpha = alpha; f.beta = beta; f.m = m; f.C = C; f.v = v; f.xbar = xbar; f.S = S; f.pi = pi; f.pack = pack;
    f.stats_out = stats_out; f.pivot = pivot;
    for (int g = 0; g < nranks; ++g) {
        if (!peers[g]) { set_error("vmp_mix_finalize_exchange: peers[%d] is null", g); return VMP_E_BADARG; }
        f.peer[g] = static_cast<double*>(peers[g]);
    }
    f.nranks = nranks; f.rank = rank; f.iter = iteration; f.status = status;
    return run_finalize(f, D, static_cast<hipStream_t>(stream));
}

int vmp_mix_finalize_ws(const void* ws, const float* pivot, int64_t N, int D, int K, int flavour, const float* alpha0, const float* beta0,
                        const float* m0, const float* C0, const float* v0, const float* kappa, float* alpha,
                        float* beta, float* m, float* C, float* v, float* xbar, float* S, float* pi, float* pack,
                        double* stats_out, void* stream) {
    int rc = check_dims(N, D, K);
    if (rc) return rc;
    if (!ws || !alpha0 || !beta0 || !m0 || !C0 || !v0) { set_error("vmp_mix_finalize_ws: null pointer"); return VMP_E_BADARG; }
    if (flavour == VMP_SMM && !kappa) { set_error("vmp_mix_finalize_ws: SMM needs kappa"); return VMP_E_BADARG; }
    FinArgs f{};
    f.partials = static_cast<const double*>(ws);
    f.nblk = make_plan(N, D, K, flavour, true).blocks;
    f.K = K; f.flavour = flavour; f.src = 0;
    f.do_post = (alpha || beta || m || C || v || xbar || S || pi || pack) ? 1 : 0;   // stats_out only: reduction only
    f.alpha0 = alpha0; f.beta0 = beta0; f.m0 = m0; f.C0 = C0; f.v0 = v0; f.kappa = kappa;
    f.alpha = alpha; f.beta = beta; f.m = m; f.C = C; f.v = v; f.xbar = xbar; f.S = S; f.pi = pi; f.pack = pack;
    f.stats_out = stats_out; f.pivot = pivot;
    return run_finalize(f, D, static_cast<hipStream_t>(stream));
}

int vmp_mix_finalize_ws64(const void* ws, const float* pivot, int64_t N, int D, int K, int flavour, const float* alpha0, const float* beta0,
                          const float* m0, const float* C0, const float* v0, const float* kappa, float* alpha,
                          float* beta, float* m, float* C, float* v, float* xbar, float* S, float* pi, float* pack, double* pack64,
                          double* stats_out, void* stream) {
    int rc = check_dims(N, D, K);
    if (rc) return rc;
    if (!ws || !alpha0 || !beta0 || !m0 || !C0 || !v0 || !pack || !pack64) { set_error("vmp_mix_finalize_ws64: null pointer"); return VMP_E_BADARG; }
    if (flavour == VMP_SMM && !kappa) { set_error("vmp_mix_finalize_ws64: SMM needs kappa"); return VMP_E_BADARG; }
    FinArgs f{};
    f.partials = static_cast<const double*>(ws);
    f.nblk = make_plan(N, D, K, flavour, true).blocks;
    f.K = K; f.flavour = flavour; f.src = 0; f.do_post = 1;
    f.alpha0 = alpha0; f.beta0 = beta0; f.m0 = m0; f.C0 = C0; f.v0 = v0; f.kappa = kappa;
    f.alpha = alpha; f.beta = beta; f.m = m; f.C = C; f.v = v; f.xbar = xbar; f.S = S; f.pi = pi; f.pack = pack; f.pack64 = pack64;
    f.stats_out = stats_out; f.pivot = pivot;
    return run_finalize(f, D, static_cast<hipStream_t>(stream));
}

int vmp_mix_estep_accurate(const float* x, int64_t N, int D, int K, int flavour, const double* pack64, float* r_out, float* u_out,
                           float* logr_out, void* stream) {
    int rc = check_dims(N, D, K);
    if (rc) return rc;
    if (!x || !pack64 || !r_out) { set_error("vmp_mix_estep_accurate: null pointer"); return VMP_E_BADARG; }
    if (flavour != VMP_GMM && flavour != VMP_SMM) { set_error("vmp_mix_estep_accurate: bad flavour %d", flavour); return VMP_E_BADARG; }
    if (flavour == VMP_SMM && !u_out) { set_error("vmp_mix_estep_accurate: SMM needs u_out"); return VMP_E_BADARG; }
    AccArgs a{x, pack64, r_out, u_out, logr_out, N, K};
    long long blocks = (N + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const size_t lds = (size_t)K * pack_words(D) * sizeof(double);
    hipStream_t s = static_cast<hipStream_t>(stream);
    VMP_DISPATCH_D(D, {
        if (flavour == VMP_SMM) hipLaunchKernelGGL((estep_f64_kernel<DD, true>), dim3((int)blocks), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((estep_f64_kernel<DD, false>), dim3((int)blocks), dim3(256), lds, s, a);
    });
    return check_launch("estep_f64_kernel");
}

int vmp_mix_iterate(const float* x, int64_t N, int D, int K, int flavour, const float* alpha0, const float* beta0,
                    const float* m0, const float* C0, const float* v0, const float* kappa, const float* pivot,
                    float* r, float* u, float* alpha, float* beta, float* m, float* C, float* v, float* xbar, float* S,
                    float* pi, float* pack, void* ws, size_t ws_bytes, int iterations, void* stream) {
    if (iterations < 0 || !pack) { set_error("vmp_mix_iterate: bad argument"); return VMP_E_BADARG; }
    for (int it = 0; it < iterations; ++it) {
        int rc = vmp_mix_finalize_ws(ws, pivot, N, D, K, flavour, alpha0, beta0, m0, C0, v0, kappa, alpha, beta, m, C, v,
                                     xbar, S, pi, pack, nullptr, stream);
        if (rc) return rc;
        rc = vmp_mix_estep_fused(x, N, D, K, flavour, pack, r, u, nullptr, pivot, ws, ws_bytes, stream);
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"
