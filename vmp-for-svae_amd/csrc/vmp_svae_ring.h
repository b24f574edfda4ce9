// T2 backward, LDS-ring form (round 3; round 4: any 8 <= K <= 16 and the Student-t theta).
// Replaces TF's autodiff through models/svae.py:14-119 and the two per-sample densities of compute_elbo(_smm)
// (svae.py:229-252, 265-322; distributions/gaussian.py:74-105, student_t.py:7-39).
//
// What bounded the generic kernel (vmp_svae.hip; measured): with the sample arithmetic REMOVED it still
// takes 3.2 ms at C3, with the loads removed 1.9 ms - the time is the access pattern, not the arithmetic: every lane fetches
// its own 64-byte piece of a 320-byte cell row, so each load instruction touches 64 different cache lines and the vector L1
// spends its cycles on tag look-ups (TCP pending-stall 62 %).  Here the two sample rows (x, dL/dx) of a sample PAIR of the
// wave's cells are brought in by global_load_lds_dwordx4 with FOUR ADJACENT LANES PER CELL (one 64-byte segment per lane
// quad: 16 look-ups per instruction instead of 64) into a per-wave ring of two stages; a stage is drained to registers and
// re-requested for the pair after next BEFORE the arithmetic of its pair starts, so two pairs (16 KB per wave, 128 KB per CU)
// are always in flight and no VGPR is spent on prefetch.  The LDS this needs comes from the per-component accumulators: the
// rows of a tile are summed across lanes first (fixed order), so a wave keeps [rows][16] floats instead of a lane-private
// [rows][64] column set.  One block per CU.  Even L >= 4 and even S >= 4 (a sample pair is L/2 16-byte pieces).
//
// Tile = RPT whole data rows = CT = RPT*K consecutive cells, one cell per lane:
//   K = 16   RPT = 4, lane = cell index: a data row is one 16-lane DPP row, lanes l, l^16, l^32, l^48 own the same component - row
//            sums by DPP rotations, component sums by v_permlane32_swap / v_permlane16_swap;
//   8 <= K < 16 (round 4; C2/C4's K = 10: 60 of 64 lanes busy - a 16-lane-per-row layout with masked lanes would run the
//            VALU-bound cell arithmetic at K/16 of the lanes and lose to the generic kernel): RPT = 64 / K in {4..8} rows.
//            Round 5: tile rows 0..3 sit in columns 0..K-1 of the four DPP rows - already the K = 16 arrangement - and the cells of
//            rows 4..RPT-1 fill the 16 - K spare lanes of each DPP row (the map is spelled out at `cell_of_slot` below).  A value that
//            is summed across lanes costs ONE ds_bpermute (a main lane pulls the value of the cell four rows below it) or none
//            (K >= 13) before the very same DPP / permlane reductions apply.  Round 4 laid the cells out as lane = r K + k, which no
//            DPP row matches, and gathered every value twice (~130 ds_bpermute and their selects per tile at K = 10).
// Row sums (d eta of the encoder): a reduce-scatter over the 16 lanes of a data row leaves value c in lane c; one store per tile.
// Student-t theta (round 4; BASELINE configs[4]): the theta term of T' is (nu+L)/2 log1p(delta^2/nu), so the sample loop
// scales W^T W (x - m) by c_s = (nu+L)/(nu+delta_s^2), and theta/mu_k, theta/L_k are trainable: per cell
//   cy = sum_s g_s y_s,  Qy = sum_s g_s y_s d_s^T (lower),  g_s = gT c_s / S,  d = x - m,  y = W d
// stay in registers over the cell's S samples (the generic kernel does 44 LDS read-modify-writes per SAMPLE), then
// dL/dm = -W^T cy and dL/dW = Qy join the tile's cross-lane sums and the wave's LDS accumulators ([2(L+TRI+1)][16] floats
// instead of [L+TRI+1][16]: at L = 8 seven waves per CU fit the 160 KB instead of eight).
#pragma once
#include "vmp_svae_cell.h"

using namespace vmp;

namespace {

// LDS table of the Student-t theta parameters (per component): rows of the lower-triangular W padded to whole 16-byte
// pieces, then m and h (padded to LP)
template <int L>
struct SvRingTab {
    __host__ __device__ static constexpr int rlen(int i) { return ((i + 1) + 3) & ~3; }
    __host__ __device__ static constexpr int roff(int i) { return i == 0 ? 0 : roff(i - 1) + rlen(i - 1); }
    static constexpr int WTOT = roff(L);
    static constexpr int LP = (L + 3) & ~3;
    static constexpr int RAW = WTOT + 2 * LP;
    static constexpr int TST = ((RAW / 4) & 1) ? RAW : RAW + 4;      // odd number of 16-byte pieces
};

// Sixteen values per lane, summed over the 16 lanes of every DPP row: on return lane c of the row holds the total of value c
// (v[0]).  Recursive halving with the lane pairings xor 8 (row_ror:8), xor 7 (row_half_mirror), xor 2 and xor 1 (quad_perm): at
// every stage a lane keeps the half of its values whose index bit matches its own lane bit and adds its partner's partial sums of
// that half - 15 exchanges instead of 16 all-reduces of 4 rotations each, and the result is already spread one value per lane,
// which is what ONE coalesced store of the row sums needs.
__device__ __forceinline__ float row16_reduce_scatter(float (&v)[16], int c) {
#define VMP_RS(N, BIT, CTRL)                                                                                            \
    _Pragma("unroll") for (int j = 0; j < N; ++j) {                                                                      \
        const bool up = (c & BIT) != 0;                                                                                  \
        const float keep = up ? v[j + N] : v[j], send = up ? v[j] : v[j + N];                                            \
        v[j] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), CTRL, 0xf, 0xf, true));        \
    }
    VMP_RS(8, 8, 0x128) VMP_RS(4, 4, 0x141) VMP_RS(2, 2, 0x4E) VMP_RS(1, 1, 0xB1)
#undef VMP_RS
    return v[0];
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the counter takes an immediate): exact for n <= 40, else drains
__device__ __forceinline__ void wait_vmcnt(int n) {
#define VMP_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        VMP_W(1) VMP_W(2) VMP_W(3) VMP_W(4) VMP_W(5) VMP_W(6) VMP_W(7) VMP_W(8) VMP_W(9) VMP_W(10) VMP_W(11) VMP_W(12) VMP_W(13)
        VMP_W(14) VMP_W(15) VMP_W(16) VMP_W(17) VMP_W(18) VMP_W(19) VMP_W(20) VMP_W(21) VMP_W(22) VMP_W(23) VMP_W(24) VMP_W(25)
        VMP_W(26) VMP_W(27) VMP_W(28) VMP_W(29) VMP_W(30) VMP_W(31) VMP_W(32) VMP_W(33) VMP_W(34) VMP_W(35) VMP_W(36) VMP_W(37)
        VMP_W(38) VMP_W(39) VMP_W(40)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef VMP_W
}

// NSTG = 2: eight waves per CU (two per SIMD), two stages per wave (rounds 3 / 4).
// NSTG = 4: FOUR waves per CU - one per SIMD, 512 registers - and four stages per wave: the same 128 KB per CU in flight,
//           requested four pairs ahead; no form is register-starved (the Student-t theta keeps W and all its sums in registers and
//           runs the packed sample loop), and every memory operation inside the tile loop sits on a COUNTED wait: the tile's small
//           inputs come by DMA one tile ahead, the row sums leave with one store per tile (see below).
#ifndef VMP_RING_NT
#define VMP_RING_NT 0        // A/B: 2 = the sample DMA carries nt
#endif
#ifndef VMP_RING_INPUTS_AHEAD
#define VMP_RING_INPUTS_AHEAD 0   // A/B: 1 = the tile's small inputs are loaded one tile ahead into registers (round 5: no gain, see below)
#endif
#ifdef VMP_DEBUG_TS
// exploration builds: clock64 stamps of ONE tile (the 9th of block 0, wave 0) - tools/ring_ts.py
#define RG_TS(i) do { if (a.dbg_t && blockIdx.x == 0 && wave == 0 && tile_it == 8 && lane == 0) a.dbg_t[i] = clock64(); } while (0)
#define RG_USE(v) asm volatile("" :: "v"(v))
#else
#define RG_TS(i) do { } while (0)
#define RG_USE(v) do { } while (0)
#endif
// KS = 16: K = 16.  KS = 0: any 8 <= K <= 15 at run time (spare-lane cells in linear order, one ds_bpermute per summed value).
// KS in {8, 10, 11, 12}: K = KS at compile time, spare-lane cells in DPP-rotation order (no ds_bpermute at all) - see the lane map.
template <int KS> struct SvRingRot {
    static constexpr int K = KS, E = 16 - K, XR = 64 / K - 4;          // spare lanes per DPP row; tile rows beyond the fourth
    static constexpr int CPR = XR > 0 ? 4 / XR : 1;                     // DPP rows that share one extra tile row (its "chunks")
    static constexpr int U = (K + CPR - 1) / CPR;                       // components per chunk
    static constexpr bool ok = XR > 0 && (XR == 1 || XR == 2 || XR == 4) && U <= E;
};
template <int N, int ROWS> __device__ __forceinline__ float row_ror_masked(float v) {
    // rows of ROWS: the value of the lane N columns to the left (rotating inside the 16-lane row); other rows: -0.0
    // (-0.0, the identity of the float add, in the rows left out: lets the compiler fold move + add into one v_add_f32_dpp)
    return __int_as_float(__builtin_amdgcn_update_dpp((int)0x80000000u, __float_as_int(v), 0x120 + N, ROWS, 0xf, false));
}
template <int L, int KS, bool STUDENT, int NSTG>
__global__ __launch_bounds__((NSTG == 2 ? SVR_NW : 4) * WAVE) void svae_estep_bwd_ring_kernel(EBwdArgs a, int nblk_abi) {
    constexpr bool K16 = KS == 16;
    constexpr bool KR = KS != 16 && KS != 0;                 // compile-time K < 16, rotation layout
    using RT = SvRingRot<KR ? KS : 8>;
    static_assert(!KR || RT::ok, "no rotation layout for this K");
    constexpr bool SP = STUDENT && NSTG == 4;                // Student-t, parameters in registers, packed sample loop
    constexpr bool ST = STUDENT && NSTG == 2;                // Student-t, parameters in an LDS table, scalar sample loop
    constexpr bool PF = NSTG == 4;                           // small inputs of a tile prefetched by DMA
    constexpr int TRI = SvGeo<L>::TRI;
    constexpr int PW = 2 * (L + TRI + 1);
    constexpr int TH = L + TRI + 1;                          // phi-side sums; theta-side sums (Student-t) follow at TH
    constexpr int PWa = STUDENT ? PW : TH;                   // accumulator rows in use
    constexpr int PP = L / 2;                                // 16-byte pieces of one sample pair of one cell
    constexpr int STG = svr_stage_floats<L>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: tile indices and DMA bases stay scalar
    const int K = KS ? KS : a.K, S = a.S;
    const int LSn = L * S, NP = S >> 1;
    const int RPT = KS ? WAVE / KS : WAVE / K, CT = RPT * K;
    constexpr int PSTR = TRI | 1;
    // Student-t: the theta parameters of a component live in an LDS table instead of 52 VGPRs per lane (with the 44 theta-side
    // sums of the cell the L = 8 sample loop spilled): [W rows, each padded to whole 16-byte pieces | m | h], stride = odd
    // number of 16-byte pieces (16 components -> 16 different bank groups; equal components read one address)
    constexpr int TST = ST ? SvRingTab<L>::TST : 0;
    const int tab = ((K * PSTR + 3) & ~3) + K * TST;
    float* pk_lds = smem;                                    // [K][PSTR]  lower triangle of P_k
    float* th_lds = smem + ((K * PSTR + 3) & ~3);            // [K][TST]   Student-t only
    float* ring = smem + tab + wave * (NSTG * STG);          // NSTG stages: [x pair: 64 cells x 2L | dx pair: 64 cells x 2L]
    float* accw = smem + tab + nw * (NSTG * STG) + wave * (PWa * 16);   // this wave's per-component sums [PWa][16]
    constexpr int PFW = PF ? 5 * WAVE : 0;                   // [eta1 rows | eta2d rows | dL/dT' | dL/dlog z | log z] of the next tile
    float* pfb = smem + tab + nw * (NSTG * STG) + nw * (PWa * 16) + wave * PFW;
    // ---- lane <-> cell map of a tile (round 5 for K < 16).  DPP row dr = lanes 16 dr .. 16 dr + 15, column col = lane & 15:
    //   columns 0 .. K-1   MAIN cells: component col of tile row dr - rows 0..3 sit in the K = 16 arrangement as they are;
    //   columns K .. 15    the E = 16 - K spare lanes of the four DPP rows take the cells of tile rows 4 .. RPT-1 in linear order:
    //                      spare x = dr E + (col - K)  <->  cell (4 + x / K, x % K)   (4 E >= (RPT - 4) K for every 8 <= K <= 15).
    // A value that is summed across lanes needs ONE ds_bpermute per value ("pull": main lane (dr, col) fetches the value of cell
    // (4 + dr, col) and adds it to its own) or none (K >= 13: RPT = 4) before the DPP / permlane reductions of the K = 16 kernel;
    // the round-4 layout lane = r K + k needed two gathers per value (90 of them per tile for the component sums at K = 10).
    // The DMA slot of a cell is its lane, so only the per-lane source offsets (cbyte) know about the map.
    const int dr = lane >> 4, col = lane & 15;
    const int E = 16 - K, XR = RPT - 4;                     // spare lanes per DPP row; tile rows beyond the fourth
    // K known at compile time (KS = 8, 10, 11, 12; the reference's K = 10): the spare lanes are filled in DPP-ROTATION order instead -
    // an extra tile row 4 + xr is cut into CPR = 4 / (RPT - 4) chunks of U components, chunk c of it sits in columns K .. K+U-1 of DPP
    // row xr CPR + c - so that ONE masked row rotation per chunk (v_add_f32 with row_ror:(16 - K + c U) row_mask) drops every spare
    // value onto the main lane of its component: no ds_bpermute, no LDS round trip; the row sums of an extra row are the sums over
    // the spare lanes of its CPR DPP rows (one permlane swap + add per doubling).
    auto cell_of_slot = [&](int sl) -> int {                // tile cell index held by lane / DMA slot sl, or -1
        if constexpr (K16) return sl;
        const int d_ = sl >> 4, c_ = sl & 15;
        if (c_ < K) return d_ * K + c_;
        if constexpr (KR) {
            const int j_ = c_ - K, cp_ = (d_ % RT::CPR) * RT::U + j_;
            return (j_ < RT::U && cp_ < K) ? (4 + d_ / RT::CPR) * K + cp_ : -1;
        } else {
            const int x_ = d_ * E + (c_ - K);
            return x_ < XR * K ? 4 * K + x_ : -1;
        }
    };
    const bool mainl = K16 ? true : col < K;
    const int own = cell_of_slot(lane);
    const bool lane_on = own >= 0;
    const int r = (K16 || mainl) ? dr : (lane_on ? own / K : RPT), k = (K16 || mainl) ? col : (lane_on ? own - (own / K) * K : 0);
    // pull: main lane (dr, col) takes the value of cell (4 + dr, col) = spare x' = dr K + col
    const bool pv = !K16 && !KR && mainl && dr < XR;
    const int xq_ = dr * K + col;
    const int pa = 4 * (pv ? (xq_ / (E > 0 ? E : 1)) * 16 + K + xq_ % (E > 0 ? E : 1) : 0);      // byte address for ds_bpermute
    // (the exchange is executed by ALL lanes and the result selected afterwards: inside `pv ? bpermute(..) : 0` it would run
    //  under the exec mask of the pv lanes only, and a ds_bpermute that reads from a disabled lane returns 0)
    auto pull = [&](float v) { const float t_ = __uint_as_float(__builtin_amdgcn_ds_bpermute(pa, __float_as_uint(v))); return pv ? t_ : 0.f; };
    // rotation layout: the spare lanes' values of e (zero elsewhere) dropped onto the main lanes of their components
    auto rot_in = [&](float e) -> float {
        if constexpr (KR) {
            float t_ = row_ror_masked<RT::E, RT::CPR == 1 ? 0xF : RT::CPR == 2 ? 0x5 : 0x1>(e);
            if constexpr (RT::CPR >= 2) t_ += row_ror_masked<RT::E + RT::U, RT::CPR == 2 ? 0xA : 0x2>(e);
            if constexpr (RT::CPR == 4) { t_ += row_ror_masked<RT::E + 2 * RT::U, 0x4>(e); t_ += row_ror_masked<RT::E + 3 * RT::U, 0x8>(e); }
            return t_;
        } else return 0.f;
    };
    // ... and a per-DPP-row quantity of the spare lanes summed over the CPR DPP rows that share an extra tile row (all of them get it)
    auto rows_total = [&](float t_) -> float {
        if constexpr (KR) {
            if constexpr (RT::CPR >= 2) {
                const auto s_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(t_), __float_as_uint(t_), false, false);
                t_ = __uint_as_float(s_[0]) + __uint_as_float(s_[1]);
            }
            if constexpr (RT::CPR == 4) {
                const auto s_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(t_), __float_as_uint(t_), false, false);
                t_ = __uint_as_float(s_[0]) + __uint_as_float(s_[1]);
            }
        }
        return t_;
    };
    // The TH values a cell contributes to its component's sums -> this wave's LDS accumulator rows [row0, row0 + TH).
    // Four values at a time are summed over the tile's rows by a REDUCE-SCATTER across the four DPP rows: v_permlane32_swap of two
    // values leaves [a.lo | b.lo], [a.hi | b.hi], whose sum holds a's half-sums in lanes 0..31 and b's in lanes 32..63 (one swap + one
    // add for TWO values); v_permlane16_swap of two such results does the same across odd / even rows.  Three swaps + three adds per four
    // values (an all-reduce per value was two swaps + two adds EACH), and the result register holds, in DPP row dr, the total of value
    // 4 j + {0, 2, 1, 3}[dr] for component col: all 64 lanes then update the accumulators - 12 LDS reads and 12 writes per call instead
    // of 45 + 45 by 16 lanes (stage stamps, profiles/r04_ring_stage_stamps.txt: this step was 3.3 k of a tile's 24.5 k cycles at K = 16,
    // 6.8 k of 31.9 k at K = 10, and ran twice per tile for the Student-t theta).  K < 16: the values are first gathered into the K = 16
    // arrangement (ds_bpermute, all exchanges of a chunk issued together, one wait).
    auto acc_batches = [&](const float (&vals)[TH], int row0) {
        constexpr int NG = (TH + 3) / 4;                                   // groups of four values
        constexpr int GC = 3;                                              // groups per chunk (12 values)
        const int vm = ((dr & 1) << 1) | (dr >> 1);                        // {0, 2, 1, 3}[dr]
#pragma unroll
        for (int g0 = 0; g0 < NG; g0 += GC) {
            float w[4 * GC];
            if constexpr (K16) {
#pragma unroll
                for (int u = 0; u < 4 * GC; ++u) w[u] = (4 * g0 + u < TH) ? vals[4 * g0 + u < TH ? 4 * g0 + u : 0] : 0.f;
            } else if constexpr (KR) {
#pragma unroll
                for (int u = 0; u < 4 * GC; ++u) {
                    const float v_ = (4 * g0 + u < TH) ? vals[4 * g0 + u < TH ? 4 * g0 + u : 0] : 0.f;
                    w[u] = (4 * g0 + u < TH) ? (mainl ? v_ : 0.f) + rot_in((!mainl && lane_on) ? v_ : 0.f) : 0.f;
                }
            } else {
                float t1[4 * GC];
#pragma unroll
                for (int u = 0; u < 4 * GC; ++u) {
                    t1[u] = 0.f;
                    w[u] = (4 * g0 + u < TH && mainl) ? vals[4 * g0 + u < TH ? 4 * g0 + u : 0] : 0.f;
                    if (4 * g0 + u < TH && RPT > 4)
                        t1[u] = __uint_as_float(__builtin_amdgcn_ds_bpermute(pa, __float_as_uint(vals[4 * g0 + u < TH ? 4 * g0 + u : 0])));
                }
                if (RPT > 4) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int u = 0; u < 4 * GC; ++u) { asm volatile("" : "+v"(t1[u])); w[u] += pv ? t1[u] : 0.f; }
                }
            }
            float R[GC], oldv[GC];
            int addr[GC];
#pragma unroll
            for (int g = 0; g < GC; ++g) {
                if (g0 + g < NG) {
                    const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[4 * g]), __float_as_uint(w[4 * g + 1]), false, false);
                    const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[4 * g + 2]), __float_as_uint(w[4 * g + 3]), false, false);
                    const float P = __uint_as_float(s1[0]) + __uint_as_float(s1[1]), Q = __uint_as_float(s2[0]) + __uint_as_float(s2[1]);
                    const auto s3 = __builtin_amdgcn_permlane16_swap(__float_as_uint(P), __float_as_uint(Q), false, false);
                    R[g] = __uint_as_float(s3[0]) + __uint_as_float(s3[1]);
                    const int vi = 4 * (g0 + g) + vm;
                    addr[g] = vi < TH ? (row0 + vi) * 16 + col : -1;
                    oldv[g] = accw[addr[g] < 0 ? 0 : addr[g]];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int g = 0; g < GC; ++g)
                if (g0 + g < NG) {
                    asm volatile("" : "+v"(oldv[g]));
                    if (addr[g] >= 0) accw[addr[g]] = oldv[g] + R[g];
                }
        }
    };
    // where a spare lane finds the sum of ITS tile row r >= 4 after the pulled row sums: any lane of DPP row r - 4
    const int back = 4 * (((mainl ? dr : r - 4) & 3) * 16);

    for (int e = threadIdx.x; e < K * TRI; e += blockDim.x) {
        const int kk = e / TRI, idx = e - kk * TRI;
        int i = 0;
        while (tri(i + 1, 0) <= idx) ++i;
        const int j = idx - tri(i, 0);
        pk_lds[kk * PSTR + idx] = a.Pk[(kk * L + i) * L + j];
    }
    for (int e = lane; e < PWa * 16; e += WAVE) accw[e] = 0.f;
    if constexpr (ST) {
        using T = SvRingTab<L>;
        for (int e = threadIdx.x; e < K * TST; e += blockDim.x) {
            const int kk = e / TST, f = e - kk * TST;
            float v = 0.f;
            if (f < T::WTOT) {
                int i = 0;
                while (i + 1 < L && T::roff(i + 1) <= f) ++i;
                const int j = f - T::roff(i);
                if (j <= i) v = a.Wk[(kk * L + i) * L + j];
            } else if (f < T::WTOT + T::LP) {
                if (f - T::WTOT < L) v = a.mk[kk * L + (f - T::WTOT)];
            } else if (f < T::WTOT + 2 * T::LP) {
                if (f - T::WTOT - T::LP < L) v = a.hk[kk * L + (f - T::WTOT - T::LP)];
            }
            th_lds[e] = v;
        }
    }
    __syncthreads();

    const int kc = lane_on ? k : 0;
    // Gaussian theta: the component's parameters stay in this lane's registers for the whole kernel, W in the padded pair
    // layout of the sample loop (below)
    constexpr int TPRk = ((L + 1) * (L + 1)) / 4;
    float hkk[ST ? 1 : L];
    v2f m2[ST ? 1 : L / 2], W2[ST ? 1 : TPRk];
    if constexpr (!ST) {
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const float hv = a.hk[kc * L + i], mv = a.mk[kc * L + i];
            hkk[i] = lane_on ? hv : 0.f;
            m2[i / 2][i & 1] = lane_on ? mv : 0.f;
#pragma unroll
            for (int j = 0; j < 2 * ((i + 2) / 2); ++j) {
                const float wv = a.Wk[(kc * L + i) * L + (j <= i ? j : i)];
                W2[((i + 1) * (i + 1)) / 4 + j / 2][j & 1] = (lane_on && j <= i) ? wv : 0.f;
            }
        }
    }
    float nuk = 1.f;
    if constexpr (STUDENT) { const float nv = a.nu[kc]; nuk = lane_on ? nv : 1.f; }
    const int th_off = kc * TST;                             // this lane's row of the theta table

    const long long ntiles = (a.N + RPT - 1) / RPT;
    const long long tstride = (long long)gridDim.x * nw;
    const float invS = 1.0f / (float)S;
    // DMA slot walk: slot g = j*64 + lane of instruction j holds piece (g % PP) of tile cell (g / PP); for L = 8 the piece
    // index is XOR-swizzled with bits 2-3 of the cell so that the 16 lanes a ds_read_b128 serves together hit 16 banks
    int dcell[PP], dpiece[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int g = j * WAVE + lane;
        dcell[j] = g / PP;
        const int pc = g - dcell[j] * PP;
        dpiece[j] = (PP == 4) ? (pc ^ ((dcell[j] >> 2) & 3)) : pc;
    }
    const int sw = (PP == 4) ? ((lane >> 2) & 3) : 0;       // read side of the same swizzle (this lane's cell = lane)
    // L = 8: a sample pair of a cell is 64 bytes, a cell row S/2 of them.  With S/2 odd every second cell row starts in the
    // middle of a 128-byte L2 line, and that line holds the LAST pair of the cell before it: requested S/2 - 1 pair-times
    // apart, the line has left the L2 in between and crosses the fabric twice (TCC_MISS 28.8 M for 20 M lines at N = 250 000,
    // profiles/r04_t2_pmc_summary.txt).  Such cells take their pairs in ROTATED order (1, 2, .., S/2 - 1, 0; nothing in the
    // arithmetic depends on the order of a cell's samples): its pair 0 is then asked for by the SAME DMA instruction as its
    // neighbour's last pair - one 128-byte request.
    const bool rotate = (PP == 4) && (NP & 1);
    // Per-lane part of the DMA source address, fixed for the whole kernel: byte offset of (cell, piece) inside a FULL tile and the
    // cell's parity.  A request then costs one add and a three-instruction pair rotation per DMA instruction on top of a wave-uniform
    // base (the stage stamps showed 670-1900 cycles per re-request: ~85 VALU instructions of 64-bit address arithmetic per 8 DMAs).
    unsigned cbyte[PP], cpar[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int dc = (PP == 4) ? ((j * WAVE + lane) >> 2) : dcell[j];
        const int dp = (PP == 4) ? ((lane & 3) ^ ((dc >> 2) & 3)) : dpiece[j];
        const int ci = cell_of_slot(dc);
        const int cc = ci >= 0 ? ci : CT - 1;                // K < 16: lanes without a cell re-fetch the tile's last cell
        cbyte[j] = (unsigned)(cc * LSn + 4 * dp) * 4u;
        cpar[j] = (unsigned)cc & 1u;
    }
    auto issue = [&](long long tt, int pp, float* stage) {
        // pair pp of tile tt -> stage; cells past the end of the tile / of the tensor are clamped to the tile's last valid cell
        const long long cells_left = (a.N - tt * RPT) * K;
        const long long tile0 = tt * (long long)CT * LSn;
        const unsigned par0 = (unsigned)((tt * CT) & 1);     // parity of the tile's first cell
        const char* __restrict__ xb = reinterpret_cast<const char*>(a.x + tile0);      // wave-uniform bases
        const char* __restrict__ gb = reinterpret_cast<const char*>(a.Gx + tile0);
        if (cells_left >= CT) {                              // full tile (all but the last one)
#pragma unroll
            for (int j = 0; j < PP; ++j) {
                unsigned pe = (unsigned)pp;
                if (rotate) { pe = (unsigned)pp + (cpar[j] ^ par0); pe = pe >= (unsigned)NP ? pe - (unsigned)NP : pe; }
                const unsigned ob = cbyte[j] + pe * (unsigned)(2 * L * 4);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + ob),
                                                 (__attribute__((address_space(3))) void*)(stage + j * (4 * WAVE)), 16, 0, VMP_RING_NT);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + ob),
                                                 (__attribute__((address_space(3))) void*)(stage + WAVE * 2 * L + j * (4 * WAVE)), 16, 0, VMP_RING_NT);
            }
            return;
        }
        const int ncell = (int)cells_left;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int dc = (PP == 4) ? ((j * WAVE + lane) >> 2) : dcell[j];
            const int dp = (PP == 4) ? ((lane & 3) ^ ((dc >> 2) & 3)) : dpiece[j];
            const int ci = cell_of_slot(dc);
            const int cc = (ci >= 0 && ci < ncell) ? ci : ncell - 1;
            int pe = pp;
            if (rotate) { pe = pp + ((cc + (int)par0) & 1); pe = pe >= NP ? pe - NP : pe; }
            const long long off = tile0 + (long long)cc * LSn + pe * 2 * L + 4 * dp;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.x + off),
                                             (__attribute__((address_space(3))) void*)(stage + j * (4 * WAVE)), 16, 0, VMP_RING_NT);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.Gx + off),
                                             (__attribute__((address_space(3))) void*)(stage + WAVE * 2 * L + j * (4 * WAVE)), 16, 0, VMP_RING_NT);
        }
    };

    // the tile's small inputs, one tile ahead (NSTG = 4): five DMA instructions, issued right BEFORE the request of the tile's pair 0
    auto prefetch = [&](long long tt) {
        if constexpr (PF) {
            const long long rows_left = a.N - tt * RPT;
            const int nrow = rows_left < RPT ? (int)rows_left : RPT;
            const int ne = nrow * L, nc = nrow * K;
            const int le = lane < ne ? lane : ne - 1, lc = lane < nc ? lane : nc - 1;      // clamped: always a valid element
            const long long e0 = tt * (long long)RPT * L, c0 = tt * (long long)CT;
#define VMP_PF(SRC, SLOT) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), \
                                                           (__attribute__((address_space(3))) void*)(pfb + (SLOT) * WAVE), 4, 0, 0)
            VMP_PF(a.eta1 + e0 + le, 0); VMP_PF(a.eta2d + e0 + le, 1); VMP_PF(a.GT + c0 + lc, 2); VMP_PF(a.Glz + c0 + lc, 3);
            VMP_PF(a.lz + c0 + lc, 4);
#undef VMP_PF
        }
    };
    // Requests are a FIFO over the wave's pairs (tile t0, pairs 0 .. NP-1; tile t0 + tstride, ...), one stage each, slot = sequence
    // number mod NSTG.  `outst` requests are outstanding when a pair is about to be consumed; the one it needs is the OLDEST, so
    // everything issued after it may stay in flight:  (outst - 1) stage requests of 2 PP DMA instructions, plus the five prefetch
    // instructions that precede a pair-0 request if one of those younger requests is a pair 0 (at most one: NP >= NSTG).  The row-sum
    // stores of a tile are NOT counted (whether a predicated store issues depends on the exec mask): at worst the wait also covers
    // the first one or two DMA instructions of the next stage.  Anything the compiler adds (a spill reload) only makes the wait
    // stricter, never too lax.
    long long t = (long long)blockIdx.x * nw + wave;
    long long rt = t;                                        // request cursor: tile
    int rp = 0, rslot = 0, cslot = 0, outst = 0;             // ... pair, slot; consumer slot; outstanding requests
    auto request_next = [&]() {
        if (rt < ntiles) {
            if (rp == 0) prefetch(rt);
            issue(rt, rp, ring + rslot * STG);
            rslot = rslot + 1 == NSTG ? 0 : rslot + 1;
            ++outst;
            if (++rp == NP) { rp = 0; rt += tstride; }
        }
    };
#pragma unroll
    for (int j = 0; j < NSTG; ++j) request_next();
    auto allowed_behind = [&](int p) {                       // VMEM operations younger than the request of pair p of the current tile
        const int younger = outst - 1;
        return younger * (2 * PP) + ((PF && p + younger >= NP) ? 5 : 0);
    };
    // Round 5 (two-stage form): the small inputs of a tile - its RPT rows of eta1 / eta2d and the three (N, K) values of every cell -
    // are loaded ONE TILE AHEAD into five registers per lane, right after the tile's last pair was re-requested: lane e holds element
    // e of the tile's RPT L eta floats (coalesced), a lane's own row is read back with 2 L ds_bpermute.  Loaded at the top of the
    // tile they cost an exposed memory round trip per tile behind the two stage requests already queued (vmcnt returns in order):
    // the stage stamps put 2.3 k (K = 16) to 8.3 k (K = 10) cycles on "eta, P_k table, Cholesky, mean", of which ~2 k is arithmetic.
    constexpr bool NX = VMP_RING_INPUTS_AHEAD && !PF;
    float nx_e1 = 0.f, nx_e2 = -0.5f, nx_gT = 0.f, nx_glz = 0.f, nx_lz = 0.f;
    auto load_inputs = [&](long long tt) {
        if constexpr (NX) {
            if (tt < ntiles) {
                const long long rows_left = a.N - tt * RPT;
                const int ne = (rows_left < RPT ? (int)rows_left : RPT) * L;
                const long long e0 = tt * (long long)RPT * L;
                const int le = lane < ne ? lane : ne - 1;                      // clamped: always a valid element
                nx_e1 = a.eta1[e0 + le]; nx_e2 = a.eta2d[e0 + le];
                const long long rw = tt * RPT + r;
                const long long cid = (lane_on && rw < a.N ? rw : tt * RPT) * K + kc;
                nx_gT = a.GT[cid]; nx_glz = a.Glz[cid]; nx_lz = a.lz[cid];
            }
        }
    };
    load_inputs(t);
    int tile_it = -1;
    for (; t < ntiles; t += tstride) {
        ++tile_it;
        RG_TS(0);
        const long long row = t * RPT + r;
        const bool on = lane_on && row < a.N;
        const long long rowc = on ? row : 0;
        const long long cellid = rowc * K + kc;

        float Lm[TRI], av[L], mu[L];
#pragma unroll
        for (int i = 0; i < TRI; ++i) Lm[i] = lane_on ? pk_lds[kc * PSTR + i] : 0.f;
        if constexpr (PF) wait_vmcnt(allowed_behind(0));     // pair 0 of this tile and everything older - its prefetch - has landed
        const int ro = (lane_on ? r : 0) * L;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            float e1v, e2v;
            if constexpr (PF) { e1v = pfb[ro + i]; e2v = pfb[WAVE + ro + i]; }
            else if constexpr (NX) {
                e1v = __uint_as_float(__builtin_amdgcn_ds_bpermute(4 * (ro + i), __float_as_uint(nx_e1)));
                e2v = __uint_as_float(__builtin_amdgcn_ds_bpermute(4 * (ro + i), __float_as_uint(nx_e2)));
            } else { e1v = a.eta1[rowc * L + i]; e2v = a.eta2d[rowc * L + i]; }
            const float e1 = on ? e1v : 0.f;
            const float e2 = on ? e2v : -0.5f;
            Lm[tri(i, i)] = fmaf(-2.f, e2, lane_on ? Lm[tri(i, i)] : 0.f);
            if constexpr (ST) av[i] = e1 + (lane_on ? th_lds[th_off + SvRingTab<L>::WTOT + SvRingTab<L>::LP + i] : 0.f);
            else av[i] = e1 + hkk[i];
        }
        float glzv, gTv, lzv;
        if constexpr (PF) { const int pl = lane_on ? r * K + k : 0; gTv = pfb[2 * WAVE + pl]; glzv = pfb[3 * WAVE + pl]; lzv = pfb[4 * WAVE + pl]; }
        else if constexpr (NX) { glzv = nx_glz; gTv = nx_gT; lzv = nx_lz; }
        else { glzv = a.Glz[cellid]; gTv = a.GT[cellid]; lzv = a.lz[cellid]; }
        float ld;
        cell_cholesky<L>(Lm, ld);
        solve_lower<L>(Lm, av);
#pragma unroll
        for (int i = 0; i < L; ++i) mu[i] = av[i];
        solve_lower_t<L>(Lm, mu);                           // mu~ = Pt^-1 ht
        RG_USE(mu[0]); RG_TS(1);

        const float glz = on ? glzv : 0.f;
        const float gT = on ? gTv : 0.f;
        const float rnk = on ? __expf(lzv) : 0.f;
        float gsum;
        if constexpr (K16) {
            gsum = row16_sum(glz);
        } else {
            gsum = row16_sum(mainl ? glz : 0.f);             // tile rows 0..3: the row's components are the DPP row's main lanes
            if constexpr (KR) {
                const float t1 = rows_total(row16_sum(mainl ? 0.f : glz));     // (glz is zero on lanes without a cell)
                gsum = mainl ? gsum : t1;
            } else if (RPT > 4) {
                const float t1 = row16_sum(pull(glz));         // DPP row dr: total of tile row 4 + dr
                const float tb = __uint_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_uint(t1)));
                gsum = mainl ? gsum : tb;
            }
        }
        const float Gc = glz - rnk * gsum;                  // through the log-sum-exp normalisation
        const float Gld = gT - Gc;                          // T' has +ld, c has -ld
        RG_USE(Gld); RG_TS(2);

        // ---- packed layouts of the sample loop (round 4).  Every product of the loop runs as v_pk_fma_f32 over PAIRS OF ADJACENT
        // COLUMNS (two floats that are neighbours in memory are neighbours in registers: no packing moves):
        //   rows of a lower triangle (W, M, Qy) padded to whole pairs: row i = RP(i) = (i+2)/2 pairs at RO(i) = (i+1)^2/4;
        //   the Cholesky factor by COLUMNS in row pairs (forward substitution updates the rows below the pivot):
        //   column j = pairs q0(j) = (j+1)/2 .. L/2-1 at CO(j) = j L/2 - j^2/4; the half that would hold the diagonal (j even)
        //   is a don't-care, that row is updated by a scalar fma; reciprocal diagonal in rd[].
        // A parameter that multiplies a pair is broadcast from one half of a register pair by the pk_*_b helpers (vmp_common.h:
        // the operand forms that are safe beside bf16 MFMAs).  The per-sample `on ? x : 0` selects are gone (64 VALU per pair):
        // an inactive lane reads the clamped copy of a valid cell (finite values), its upstream gradients gT, glz are zero and
        // everything it accumulates is masked when the tile's sums are formed.
        constexpr int LH = L / 2;
        auto RO = [](int i) { return ((i + 1) * (i + 1)) / 4; };
        auto RPn = [](int i) { return (i + 2) / 2; };
        constexpr int TPR = ((L + 1) * (L + 1)) / 4;             // pairs of a padded row-major triangle
        auto CO = [](int j) { return j * (L / 2) - (j * j) / 4; };
        auto Q0 = [](int j) { return (j + 1) / 2; };
        constexpr int TPC = L * (L / 2) - (L * L) / 4;           // pairs of the column-major strict lower triangle
        // Gaussian theta: packed storage.  Student-t: the round's first form is kept - scalar arithmetic, every row of W read
        // ONCE per sample pair from the LDS table and used for both samples: the packed form of it was slower in both variants
        // tried (two samples together: spills inside the pair loop, whose reloads drain the DMA ring - vmcnt is in order;
        // one sample at a time: twice the table reads, SQ_LDS_BANK_CONFLICT 6 M -> 21 M cycles; 2.89 -> 3.15 ms at C3)
        v2f LC[ST ? 1 : TPC];
        float rd[L];
#pragma unroll
        for (int j = 0; j < L; ++j) rd[j] = Lm[tri(j, j)];
        if constexpr (!ST) {
#pragma unroll
            for (int j = 0; j < L; ++j)
#pragma unroll
                for (int q = Q0(j); q < LH; ++q)
                    LC[CO(j) + q - Q0(j)] = v2f{2 * q > j ? Lm[tri(2 * q > j ? 2 * q : j + 1, j)] : 0.f, Lm[tri(2 * q + 1, j)]};
        }
        auto LM = [&](int i, int j) -> float {               // i >= j; the diagonal holds reciprocals
            if constexpr (ST) return Lm[tri(i, j)];
            else return i == j ? rd[i] : LC[CO(j) + i / 2 - Q0(j)][i & 1];
        };

        v2f Wsum2[ST ? 1 : LH], M2[ST ? 1 : TPR], mu2[ST ? 1 : LH];
        v2f cy2[SP ? LH : 1], Qy2[SP ? TPR : 1];             // Student-t, packed: theta-side sums of this cell
        if constexpr (SP) {
#pragma unroll
            for (int q = 0; q < LH; ++q) cy2[q] = v2f{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < TPR; ++i) Qy2[i] = v2f{0.f, 0.f};
        }
        float Wsum[ST ? L : 1], M[ST ? TRI : 1], cy[ST ? L : 1], Qy[ST ? TRI : 1], mm[ST ? L : 1];
        if constexpr (ST) {
#pragma unroll
            for (int i = 0; i < L; ++i) { Wsum[i] = 0.f; cy[i] = 0.f; mm[i] = th_lds[th_off + SvRingTab<L>::WTOT + i] - mu[i]; }
#pragma unroll
            for (int i = 0; i < TRI; ++i) { M[i] = 0.f; Qy[i] = 0.f; }
        } else {
#pragma unroll
            for (int q = 0; q < LH; ++q) { Wsum2[q] = v2f{0.f, 0.f}; mu2[q] = v2f{mu[2 * q], mu[2 * q + 1]}; }
#pragma unroll
            for (int i = 0; i < TPR; ++i) M2[i] = v2f{0.f, 0.f};
        }
        auto MU = [&](int i) -> float {                      // (one copy of mu~ lives across the sample loop)
            if constexpr (ST) return mu[i];
            else return mu2[i / 2][i & 1];
        };
        const float gts = gT * invS;
        const float nuL = nuk + (float)L;
        for (int p = 0; p < NP; ++p) {
            float* stage = ring + cslot * STG;
            // the stage about to be read holds the OLDEST outstanding request (see allowed_behind)
            wait_vmcnt(allowed_behind(p));
            RG_TS(3 + 4 * p);
            v2f xq[2][LH], gq[2][LH];                       // [sample of the pair][column pair]
#pragma unroll
            for (int q = 0; q < PP; ++q) {
                const f32x4 vx = *reinterpret_cast<const f32x4*>(stage + lane * (2 * L) + 4 * (q ^ sw));
                const f32x4 vg = *reinterpret_cast<const f32x4*>(stage + WAVE * 2 * L + lane * (2 * L) + 4 * (q ^ sw));
                // floats 4q .. 4q+3 of the cell's 2L-float pair row: sample (4q) / L, columns (4q) % L ..
                const int h0 = (4 * q) / L, c0 = ((4 * q) % L) / 2, h1 = (4 * q + 2) / L, c1 = ((4 * q + 2) % L) / 2;
                xq[h0][c0] = v2f{vx[0], vx[1]}; xq[h1][c1] = v2f{vx[2], vx[3]};
                gq[h0][c0] = v2f{vg[0], vg[1]}; gq[h1][c1] = v2f{vg[2], vg[3]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the stage is in registers: it may be overwritten
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < LH; ++q) asm volatile("" : "+v"(xq[h][q]), "+v"(gq[h][q]));
            RG_TS(4 + 4 * p);
            cslot = cslot + 1 == NSTG ? 0 : cslot + 1;
            --outst;
            request_next();                                 // the freed slot takes the next pair of the wave's sequence
            RG_TS(5 + 4 * p);
            if constexpr (ST) {
                // both samples of the pair against each row of W (one LDS read of the row serves two samples)
                using T = SvRingTab<L>;
                float d[2][L], y[2][L], gx[2][L], del2[2] = {0.f, 0.f};
                {
                    int mo = th_off + T::WTOT;
                    asm volatile("" : "+v"(mo));                       // not hoisted out of the sample loop
#pragma unroll
                    for (int i = 0; i < L; ++i) {
                        const float mv = th_lds[mo + i];
                        d[0][i] = xq[0][i / 2][i & 1] - mv; d[1][i] = xq[1][i / 2][i & 1] - mv;
                        gx[0][i] = gq[0][i / 2][i & 1]; gx[1][i] = gq[1][i / 2][i & 1];
                    }
                }
                int wo = th_off;
                asm volatile("" : "+v"(wo));
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    float w[8];
#pragma unroll
                    for (int q = 0; q <= i / 4; ++q) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(th_lds + wo + T::roff(i) + 4 * q);
#pragma unroll
                        for (int c = 0; c < 4; ++c) w[4 * q + c] = wv[c];
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float yy = 0.f;
#pragma unroll
                        for (int j = 0; j <= i; ++j) yy = fmaf(w[j], d[h][j], yy);
                        y[h][i] = yy;
                        del2[h] = fmaf(yy, yy, del2[h]);
                    }
                }
                // c_s = (nu+L)/(nu+delta^2) (student_t.py:31-37 differentiated)
                const float gc0 = gts * nuL * __builtin_amdgcn_rcpf(nuk + del2[0]), gc1 = gts * nuL * __builtin_amdgcn_rcpf(nuk + del2[1]);
                asm volatile("" : "+v"(wo));                           // second pass: the rows are read again, not kept
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    float w[8];
#pragma unroll
                    for (int q = 0; q <= i / 4; ++q) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(th_lds + wo + T::roff(i) + 4 * q);
#pragma unroll
                        for (int c = 0; c < 4; ++c) w[4 * q + c] = wv[c];
                    }
                    const float gy0 = gc0 * y[0][i], gy1 = gc1 * y[1][i];
                    cy[i] += gy0 + gy1;
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        gx[0][j] = fmaf(w[j], gy0, gx[0][j]);
                        gx[1][j] = fmaf(w[j], gy1, gx[1][j]);
                        Qy[tri(i, j)] = fmaf(gy1, d[1][j], fmaf(gy0, d[0][j], Qy[tri(i, j)]));
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    solve_lower<L>(Lm, gx[h]);                  // w_s = Lt^-1 gx_s
#pragma unroll
                    for (int i = 0; i < L; ++i) {
                        Wsum[i] += gx[h][i];
                        const float e = d[h][i] + mm[i];        // e_s = x_s - mu~ = Lt^-T eps_s
#pragma unroll
                        for (int j = 0; j <= i; ++j) M[tri(i, j)] = fmaf(e, gx[h][j], M[tri(i, j)]);
                    }
                }
            } else {
                // packed form (see the layouts above; Gaussian theta, and the Student-t theta of the four-stage kernel); the two samples of
                // the pair one after the other
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // d = x - m,  y = W d  (row i: pair products summed in two half-sums),  gy = y gT / S,  gx += W^T gy:
                    // d/dx of the theta term of T' (1/S) W^T W d
                    v2f d2[LH], y2[LH], gx2[LH];
#pragma unroll
                    for (int q = 0; q < LH; ++q) d2[q] = xq[h][q] - m2[q];
#pragma unroll
                    for (int i = 0; i < L; ++i) {
                        v2f acc = W2[RO(i)] * d2[0];
#pragma unroll
                        for (int q = 1; q < RPn(i); ++q) acc = __builtin_elementwise_fma(W2[RO(i) + q], d2[q], acc);
                        y2[i / 2][i & 1] = acc[0] + acc[1];
                    }
                    v2f gts2 = v2f{gts, gts};
                    if constexpr (SP) {                     // c_s = (nu+L)/(nu+delta^2)  (student_t.py:31-37 differentiated)
                        v2f dd = y2[0] * y2[0];
#pragma unroll
                        for (int q = 1; q < LH; ++q) dd = __builtin_elementwise_fma(y2[q], y2[q], dd);
                        const float gcs = gts * nuL * __builtin_amdgcn_rcpf(nuk + (dd[0] + dd[1]));
                        gts2 = v2f{gcs, gcs};
                    }
#pragma unroll
                    for (int q = 0; q < LH; ++q) { y2[q] = y2[q] * gts2; gx2[q] = gq[h][q]; }
                    if constexpr (SP) {
#pragma unroll
                        for (int q = 0; q < LH; ++q) cy2[q] = cy2[q] + y2[q];
                    }
#pragma unroll
                    for (int i = 0; i < L; ++i)
#pragma unroll
                        for (int q = 0; q < RPn(i); ++q) {
                            gx2[q] = pk_fma_b(W2[RO(i) + q], y2[i / 2], gx2[q], i & 1);                          // gx += W^T gy
                            if constexpr (SP) Qy2[RO(i) + q] = pk_fma_b(d2[q], y2[i / 2], Qy2[RO(i) + q], i & 1);  // Qy += gy d^T
                        }
                    // w_s = Lt^-1 gx_s: forward substitution by columns, the rows below the pivot in pairs
#pragma unroll
                    for (int j = 0; j < L; ++j) {
                        gx2[j / 2][j & 1] *= rd[j];
                        if ((j & 1) == 0) {
                            gx2[j / 2][1] = fmaf(-LC[CO(j)][1], gx2[j / 2][0], gx2[j / 2][1]);
#pragma unroll
                            for (int q = Q0(j) + 1; q < LH; ++q) gx2[q] = pk_fnma_b(LC[CO(j) + q - Q0(j)], gx2[j / 2], gx2[q], 0);
                        } else {
#pragma unroll
                            for (int q = Q0(j); q < LH; ++q) gx2[q] = pk_fnma_b(LC[CO(j) + q - Q0(j)], gx2[j / 2], gx2[q], 1);
                        }
                    }
                    v2f e2[LH];
#pragma unroll
                    for (int q = 0; q < LH; ++q) { Wsum2[q] = Wsum2[q] + gx2[q]; e2[q] = xq[h][q] - mu2[q]; }   // e_s = Lt^-T eps_s
#pragma unroll
                    for (int i = 0; i < L; ++i)
#pragma unroll
                        for (int q = 0; q < RPn(i); ++q) M2[RO(i) + q] = pk_fma_b(gx2[q], e2[i / 2], M2[RO(i) + q], i & 1);
                }
            }
            if constexpr (ST) { RG_USE(M[0]); } else { RG_USE(M2[0]); }
            RG_TS(6 + 4 * p);
        }
        load_inputs(t + tstride);                            // (every value of this tile's inputs has been consumed above)
        auto MM = [&](int i, int j) -> float {               // i >= j
            if constexpr (ST) return M[tri(i, j)];
            else return M2[RO(i) + j / 2][j & 1];
        };
        // ---- Student-t: this cell's theta-side gradients join the wave's accumulators (rows TH..PW-1) now, so that their
        //      registers are free during the assembly below:  dL/dm = -W^T cy,  dL/dW = Qy,  dL/dkappa = -gT
        if constexpr (STUDENT) {
            float tx[L];
#pragma unroll
            for (int j = 0; j < L; ++j) tx[j] = 0.f;
            if constexpr (SP) {
#pragma unroll
                for (int i = 0; i < L; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) tx[j] = fmaf(W2[RO(i) + j / 2][j & 1], cy2[i / 2][i & 1], tx[j]);
            } else {
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    float w[8];
#pragma unroll
                    for (int q = 0; q <= i / 4; ++q) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(th_lds + th_off + SvRingTab<L>::roff(i) + 4 * q);
#pragma unroll
                        for (int c = 0; c < 4; ++c) w[4 * q + c] = wv[c];
                    }
#pragma unroll
                    for (int j = 0; j <= i; ++j) tx[j] = fmaf(w[j], cy[i], tx[j]);
                }
            }
            float tvals[TH];
#pragma unroll
            for (int j = 0; j < L; ++j) tvals[j] = -tx[j];
#pragma unroll
            for (int i = 0; i < L; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) tvals[L + tri(i, j)] = SP ? Qy2[SP ? RO(i) + j / 2 : 0][j & 1] : Qy[ST ? tri(i, j) : 0];
            tvals[L + TRI] = -gT;
            acc_batches(tvals, TH);
        }
        RG_TS(23);
        // ---- assemble dLoss/dht and dLoss/dPt (symmetric, lower triangle): as in svae_estep_bwd_kernel
        float V[L];
#pragma unroll
        for (int i = 0; i < L; ++i) {
            if constexpr (ST) V[i] = Wsum[i];
            else V[i] = Wsum2[i / 2][i & 1];
        }
#pragma unroll
        for (int j = L - 1; j >= 0; --j) {                  // V = Pt^-1 sum_s gx_s  (back substitution, as solve_lower_t)
            V[j] *= rd[j];
#pragma unroll
            for (int i = 0; i < j; ++i) V[i] = fmaf(-LM(j, i), V[j], V[i]);
        }
        float gh[L];
#pragma unroll
        for (int i = 0; i < L; ++i) gh[i] = fmaf(Gc, MU(i), V[i]);
        float Cs[TRI], dg[L];
#pragma unroll
        for (int i = 0; i < L; ++i) dg[i] = 1.0f / rd[i];
#pragma unroll
        for (int i = 0; i < L; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                float s2 = 0.f;
#pragma unroll
                for (int pq = i; pq < L; ++pq) {
                    const float lpi = (pq == i) ? dg[i] : LM(pq, i);
                    s2 = fmaf(lpi, -MM(pq, j), s2);
                }
                Cs[tri(i, j)] = (i == j) ? (s2 + Gld) : s2;
            }
        float Y[TRI];
#pragma unroll
        for (int j = 0; j < L; ++j) {
            Y[tri(j, j)] = rd[j];
#pragma unroll
            for (int i = j + 1; i < L; ++i) {
                float s2 = 0.f;
#pragma unroll
                for (int pq = j; pq < i; ++pq) s2 = fmaf(LM(i, pq), Y[tri(pq, j)], s2);
                Y[tri(i, j)] = -s2 * rd[i];
            }
        }
        float gP[TRI];
#pragma unroll
        for (int i = 0; i < TRI; ++i) gP[i] = 0.f;
#pragma unroll
        for (int j = 0; j < L; ++j) {
            float Zc[L];
#pragma unroll
            for (int q = 0; q < L; ++q) {
                float s2 = 0.f;
#pragma unroll
                for (int pq = j; pq < L; ++pq) {
                    const float cqp = (q >= pq) ? Cs[tri(q, pq)] : Cs[tri(pq, q)];
                    s2 = fmaf(cqp, Y[tri(pq, j)], s2);
                }
                Zc[q] = s2;
            }
#pragma unroll
            for (int i = j; i < L; ++i) {
                float s2 = 0.f;
#pragma unroll
                for (int pq = i; pq < L; ++pq) s2 = fmaf(Y[tri(pq, i)], Zc[pq], s2);
                gP[tri(i, j)] = 0.5f * s2;
            }
        }
        // rank-one terms  - sym(V mu^T) - 1/2 Gc mu mu^T  =  -1/2 (tv mu^T + mu tv^T),  tv = V + 1/2 Gc mu
        float tv[L];
#pragma unroll
        for (int i = 0; i < L; ++i) tv[i] = -0.5f * fmaf(0.5f * Gc, MU(i), V[i]);
#pragma unroll
        for (int i = 0; i < L; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) gP[tri(i, j)] = fmaf(tv[i], MU(j), fmaf(MU(i), tv[j], gP[tri(i, j)]));

        RG_USE(gP[0]); RG_TS(24);
        // ---- per-row sums (over the components of a data row) -> d eta of the encoder.  The 2 L sums of a data row are formed by
        // a reduce-scatter over the row's 16 lanes (row16_reduce_scatter) and leave with ONE store instruction per tile: lane
        // (row, c) stores value c - d eta1[c] for c < L, d eta2d[c - L] for L <= c < 2 L - to two contiguous 4 L-byte rows (the
        // first form stored them with 2 L instructions from one lane per row; the four-stage kernel counts its memory operations)
        auto store_row_sums = [&](float (&v)[16], long long row_, bool row_ok, bool spare = false) {
            float tot = row16_reduce_scatter(v, col);
            if (spare) tot = rows_total(tot);
            const bool is1 = col < L;
            float* dst = (is1 ? a.g_eta1 : a.g_eta2d) + row_ * L + (is1 ? col : col - L);
            if (row_ok && col < 2 * L) *dst = is1 ? tot : -2.f * tot;      // p = -2 eta2d
        };
        if constexpr (K16) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = i < L ? (on ? gh[i < L ? i : 0] : 0.f) : i < 2 * L ? (on ? gP[tri(i < 2 * L && i >= L ? i - L : 0, i < 2 * L && i >= L ? i - L : 0)] : 0.f) : 0.f;
            store_row_sums(v, t * RPT + dr, t * RPT + dr < a.N);
        } else if constexpr (KR) {
            // main lanes: tile row dr; spare lanes: a chunk of tile row 4 + dr / CPR, summed over that row's CPR DPP rows
            float va[16], vb[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float x_ = i < L ? (on ? gh[i < L ? i : 0] : 0.f) : i < 2 * L ? (on ? gP[tri(i < 2 * L && i >= L ? i - L : 0, i < 2 * L && i >= L ? i - L : 0)] : 0.f) : 0.f;
                va[i] = mainl ? x_ : 0.f; vb[i] = mainl ? 0.f : x_;
            }
            const long long rowa = t * RPT + dr, rowb = t * RPT + 4 + dr / RT::CPR;
            store_row_sums(va, rowa, rowa < a.N);
            store_row_sums(vb, rowb, dr % RT::CPR == 0 && rowb < a.N, true);
        } else {
            // DPP row dr holds tile row dr in its main lanes; tile row 4 + dr is pulled into them from the spare lanes
            float va[16], vb[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {                  // all exchanges first, one wait, then the sums
                const float x_ = i < L ? (on ? gh[i < L ? i : 0] : 0.f) : i < 2 * L ? (on ? gP[tri(i < 2 * L && i >= L ? i - L : 0, i < 2 * L && i >= L ? i - L : 0)] : 0.f) : 0.f;
                va[i] = mainl ? x_ : 0.f; vb[i] = 0.f;
                if (i < 2 * L && RPT > 4) vb[i] = __uint_as_float(__builtin_amdgcn_ds_bpermute(pa, __float_as_uint(x_)));
            }
            const long long rowa = t * RPT + dr, rowb = rowa + 4;
            if (RPT > 4) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 2 * L; ++i) { asm volatile("" : "+v"(vb[i])); vb[i] = pv ? vb[i] : 0.f; }
            }
            store_row_sums(va, rowa, rowa < a.N);
            if (RPT > 4) store_row_sums(vb, rowb, dr < XR && rowb < a.N);
        }
        RG_TS(25);
        // ---- per-component sums (acc_batches above)
        {
            float pvals[TH];
#pragma unroll
            for (int i = 0; i < L; ++i) pvals[i] = on ? gh[i] : 0.f;
#pragma unroll
            for (int i = 0; i < TRI; ++i) pvals[L + i] = on ? gP[i] : 0.f;
            pvals[L + TRI] = on ? Gc : 0.f;
            acc_batches(pvals, 0);
        }
        RG_TS(26);
    }

    // ---- block reduction: waves in a fixed order, then one partial row per block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* out = a.partials + (long long)blockIdx.x * K * PW;
    for (int e = threadIdx.x; e < K * PW; e += blockDim.x) {
        const int kk = e / PW, f = e - kk * PW;
        float s2 = 0.f;
        if (f < PWa) {
            const float* base = smem + tab + nw * (NSTG * STG);
            for (int w = 0; w < nw; ++w) s2 += base[w * (PWa * 16) + f * 16 + kk];
        }
        out[e] = s2;
    }
    // rows of the partial buffer the ABI sized for more blocks than this kernel launches
    for (int b = blockIdx.x + gridDim.x; b < nblk_abi; b += gridDim.x) {
        float* z = a.partials + (long long)b * K * PW;
        for (int e = threadIdx.x; e < K * PW; e += blockDim.x) z[e] = 0.f;
    }
}

template <int L, int KS, bool STUDENT, int NSTG>
int launch_n(const EBwdArgs& a, int nblk_abi, void* stream) {
    constexpr int TRI = L * (L + 1) / 2, TH = L + TRI + 1, PWa = STUDENT ? 2 * TH : TH;
    const int K = a.K, RPT = WAVE / K;
    const int tab = ((K * (TRI | 1) + 3) & ~3) + ((STUDENT && NSTG == 2) ? K * SvRingTab<L>::TST : 0);
    const int per_wave = NSTG * svr_stage_floats<L>() + PWa * 16 + (NSTG == 4 ? 5 * WAVE : 0);
    const int maxw = NSTG == 2 ? SVR_NW : 4;
    const size_t budget = vmp::lds_budget() / sizeof(float);
    int nw = budget > (size_t)tab ? (int)((budget - tab) / per_wave) : 0;
    if (nw > maxw) nw = maxw;
    if (nw < 4) return -2;
    const long long ntiles = (a.N + RPT - 1) / RPT;
    long long bl = (ntiles + nw - 1) / nw;
    if (bl > 256) bl = 256;                                  // one block per CU
    const size_t lds = (size_t)(tab + nw * per_wave) * sizeof(float);
    auto kern = svae_estep_bwd_ring_kernel<L, KS, STUDENT, NSTG>;
    if (const int rc = vmp::set_dyn_lds(reinterpret_cast<const void*>(kern), lds, "svae_estep_bwd_ring_kernel")) return rc;
    hipLaunchKernelGGL(kern, dim3((int)bl), dim3(nw * WAVE), lds, static_cast<hipStream_t>(stream), a, nblk_abi);
    return check_launch("svae_estep_bwd_ring_kernel");
}

// (The kernel template still carries NSTG: the four-stage, one-wave-per-SIMD form measured SLOWER in round 4 - profiles/NOTES_r01-r04.md -
//  and is no longer instantiated or selectable; the shipped form is NSTG = 2: eight waves per CU, two stages each.)
template <int L, int KS, bool STUDENT>
int launch(const EBwdArgs& a, int nblk_abi, void* stream) {
    return launch_n<L, KS, STUDENT, 2>(a, nblk_abi, stream);
}

}  // namespace
