// T2: the SVAE E-step (reference models/svae.py:14-119) fused with the regulariser of the ELBO
// (models/svae.py:229-252), forward and backward, for gfx950.
//
// Cell arithmetic (validated against the reference graph): for datapoint n, component k
//   Pt = diag(p_n) + P_k,  ht = h_n + h_k,  Lt = chol(Pt),  a = Lt^-1 ht,  ld = sum_i log Lt_ii
//   c_nk = bias_k + 1/2 |a|^2 - ld,            log_z_nk = c_nk - logsumexp_k c_nk         (svae.py:50-92)
//   x_nks = Lt^-T (a + eps_nks)                                                            (svae.py:95-119)
//   T'_nk = mean_s[ log N(x_s; phi~_nk) - log N(x_s; theta_k) - E log pi_k ]
//         = -(L/2) log 2pi + ld - 1/(2S) sum_s |eps_s|^2 + 1/(2S) sum_s |W_k (x_s - m_k)|^2 - kappa_k
//   (Student-t theta, svae.py:265-322 / student_t.py:31-37:  ... + 1/(2S) sum_s (nu_k + L) log1p(|W_k (x_s - m_k)|^2 / nu_k) - kappa_k)
// so that the reference's regulariser is sum_nk r_nk (T'_nk + log_z_nk) (svae.py:245-252).  One 8x8 Cholesky
// per cell replaces the reference's 7 LU + 4 Cholesky factorisations.
//
// Work decomposition: one (n,k) cell per lane.  A wave tile is RPT = 64/K whole rows (RPT*K lanes active) so the
// softmax over k and the per-row gradient sums never leave the wave; the lane's component k = lane % K is the
// same for every tile, so the component's parameters stay resident in its VGPRs for the whole kernel.  The
// noise tile eps (layout (cell, L, S)) is brought in with coalesced vector loads and staged in LDS with an odd
// cell stride (conflict-free per-lane reads); the samples x are written back into the same LDS slots and leave
// with coalesced stores in the reference's (cell, S, L) layout.
#include "vmp_svae_cell.h"
#include <stdlib.h>

using namespace vmp;

namespace {

struct EFwdArgs {
    const float* eta1;      // (N,L)   encoder eta1
    const float* eta2d;     // (N,L)   encoder eta2 diagonal (negative)
    const float* hk;        // (K,L)   recognition eta1_k
    const float* Pk;        // (K,L,L) recognition precision (symmetric)
    const float* bias;      // (K)     B_k + log pi_k
    const float* noise;     // (N,K,L,S)
    const float* mk;        // (K,L)   E[mu_k] of theta
    const float* Wk;        // (K,L,L) lower-triangular W with W^T W = E[Sigma_k]^-1 (GMM) / Sigma_k^-1 (SMM)
    const float* kappa;     // (K)     the x-independent part of log p(x, z=k | theta)
    const float* nu;        // (K)     Student-t degrees of freedom, or NULL for the Gaussian theta
    float* x;               // (N,K,S,L)
    float* lz;              // (N,K)
    float* Tp;              // (N,K)
    long long N;
    int K, S, vec_ok;
    unsigned long long seed;   // in-kernel noise (noise == NULL): Philox4x32-7 key
    const unsigned long long* seed_dev;   // non-NULL: the key is read from this device word (graph-captured steps refresh it)
    // epilogue of the in-kernel-noise forms (all NULL: off) - what the step does next with the cell's values while they are in
    // registers: subsample_x with nb_out = 1 (svae.py:122-151: z_n ~ Cat(exp log_z_n), x_samples[n] = x[n, z_n, 0]; the draw's
    // uniform = the stand-alone sub-sampling kernel's, same key), r = exp(log_z) (svae.py:216), and - K = 16, L = 8, the
    // pair-staging forms - the raw M-step moments sum_n r_nk [1 | x_n | x_n x_n^T] of gmm.update_Nk/xk/Sk on x_samples
    // (svae.py:154-176, gmm.py:25-46) as per-block fp64 partials (K, 48)
    float* xs;              // (N,L)
    float* r;               // (N,K)
    double* mom;            // (blocks, 16, 48): features [x_0..x_7 | 1 | x_a x_b (a >= b, packed lower) | 0 0 0]
#ifdef VMP_DEBUG_TS
    long long* dbg_t;
#endif
};
__device__ __forceinline__ bool al16_dev(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
constexpr unsigned SUBSAMPLE_TAG = 0x5bb5a3c1u;  // 4th counter word of the categorical draw's Philox block (keeps it apart from the normals)
constexpr int MOMF = 48;                         // feature slots of the in-kernel moment partials (three 16-wide MFMA tiles)
constexpr int XSEL = 12;                         // per-row LDS record of the drawn sample: [x_0..x_7 | 1 | 0 | pad pad]


// ONE = every wave owns at most one tile (small batches: the reference's minibatches of 64-100 rows).  Nothing can then
// be overlapped ACROSS tiles, and the kernel as written for streaming pays its memory round trips one after the other
// (P_k table, parameters, eta, upstream gradients, first sample pair: 60 % of the wave's cycles were waits at N = 64).
// The ONE form requests every per-cell input - eta, the (N,K) gradients and the first TWO sample pairs - before the
// block's P_k table is staged, and keeps two pairs in flight in the sample loop.
template <int L, bool ONE>
__global__ __launch_bounds__(SV_NW * WAVE) void svae_estep_bwd_kernel(EBwdArgs a) {
    constexpr int TRI = SvGeo<L>::TRI;
    constexpr int PW = 2 * (L + TRI + 1);
    constexpr int TH = L + TRI + 1;                         // offset of the theta-side sums
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int K = a.K, S = a.S;
    const int LSn = L * S;
    const int RPT = WAVE / K, CT = RPT * K;
    const int PSTR = TRI | 1;
    float* pk_lds = smem;                                   // [K][PSTR]  lower triangle of P_k
    float* scr = smem + K * PSTR + wave * WAVE;
    const int PWa = (a.nu != nullptr) ? PW : TH;            // accumulator rows in use (theta-side sums only for Student-t)
    // row stride 65, not 64: the epilogue reads these arrays ACROSS rows (lane <-> row index), and with a stride of 64
    // words every lane of such a read hits the same LDS bank (measured: 8 us of a 28 us launch at N = 64)
    constexpr int AST = SV_AST;
    float* rows = smem + K * PSTR + nw * WAVE + wave * (2 * L * AST);   // this wave's row-sum scratch [2L][65]
    float* acc0 = smem + K * PSTR + nw * WAVE + nw * (2 * L * AST);
    float* accl = acc0 + wave * (PWa * AST);                // this wave's accumulators [PWa][65], lane-private columns
    const bool lane_on = lane < CT;
    const int r = lane / K, k = lane - r * K, rbase = lane_on ? r * K : 0;
    const int kc = lane_on ? k : 0;                         // unconditional loads from clamped indices + selects
    const long long ntiles = (a.N + RPT - 1) / RPT;
    SV_TS(0);
    const float invS = 1.0f / (float)S;
    float nxs[2 * L], ngx[2 * L];
    float nxs2[ONE ? 2 * L : 1], ngx2[ONE ? 2 * L : 1];     // ONE: the pair after next
    float pe1[ONE ? L : 1], pe2[ONE ? L : 1], pg[3] = {0.f, 0.f, 0.f};
    // Rows are fetched TWO samples at a time: 2*L floats = one 64-byte segment per array per lane, requested by
    // back-to-back loads.  Fetching a single 32-byte row per iteration made every row its own L2 request (the
    // other half of the segment is evicted from the 32 KiB L1 before the next sample needs it): 329 M requests
    // of ~31 B per launch at C3 (TCP_TCC_READ_REQ), i.e. the kernel was bound by L1<->L2 requests, not by HBM.
    // (one branch around the whole group of loads: skipping the tail prefetches matters - an unconditional
    //  clamped version measured 27% slower)
    auto load_pair = [&](const float* __restrict__ xc, const float* __restrict__ gc, bool on, int s2,
                         float (&xo)[2 * L], float (&go)[2 * L]) {
        const bool live = on && s2 < S;
        const bool both = s2 + 1 < S;
        if (live) {
            if ((L & 3) == 0 && a.vec_ok) {
#pragma unroll
                for (int q = 0; q < L / 4; ++q) {
                    const float4 v = reinterpret_cast<const float4*>(xc + s2 * L)[q];
                    const float4 w = reinterpret_cast<const float4*>(gc + s2 * L)[q];
                    xo[4 * q] = v.x; xo[4 * q + 1] = v.y; xo[4 * q + 2] = v.z; xo[4 * q + 3] = v.w;
                    go[4 * q] = w.x; go[4 * q + 1] = w.y; go[4 * q + 2] = w.z; go[4 * q + 3] = w.w;
                }
                if (both) {
#pragma unroll
                    for (int q = 0; q < L / 4; ++q) {
                        const float4 v = reinterpret_cast<const float4*>(xc + (s2 + 1) * L)[q];
                        const float4 w = reinterpret_cast<const float4*>(gc + (s2 + 1) * L)[q];
                        xo[L + 4 * q] = v.x; xo[L + 4 * q + 1] = v.y; xo[L + 4 * q + 2] = v.z; xo[L + 4 * q + 3] = v.w;
                        go[L + 4 * q] = w.x; go[L + 4 * q + 1] = w.y; go[L + 4 * q + 2] = w.z; go[L + 4 * q + 3] = w.w;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < L; ++i) { xo[i] = xc[s2 * L + i]; go[i] = gc[s2 * L + i]; }
                if (both) {
#pragma unroll
                    for (int i = 0; i < L; ++i) { xo[L + i] = xc[(s2 + 1) * L + i]; go[L + i] = gc[(s2 + 1) * L + i]; }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < L; ++i) {
            xo[i] = live ? xo[i] : 0.f; go[i] = live ? go[i] : 0.f;
            xo[L + i] = (live && both) ? xo[L + i] : 0.f; go[L + i] = (live && both) ? go[L + i] : 0.f;
        }
    };
    if constexpr (ONE) {
        const long long t = (long long)blockIdx.x * nw + wave;
        const long long row = t * RPT + r;
        const bool on = lane_on && t < ntiles && row < a.N;
        const long long rowc = on ? row : 0;
        const long long cellid = rowc * K + kc;
        load_pair(a.x + cellid * LSn, a.Gx + cellid * LSn, on, 0, nxs, ngx);
        load_pair(a.x + cellid * LSn, a.Gx + cellid * LSn, on, 2, nxs2, ngx2);
#pragma unroll
        for (int i = 0; i < L; ++i) { pe1[i] = a.eta1[rowc * L + i]; pe2[i] = a.eta2d[rowc * L + i]; }
        pg[0] = a.Glz[cellid]; pg[1] = a.GT[cellid]; pg[2] = a.lz[cellid];
    }
    SV_TS(1);

    // (requested before the table is staged: one memory round trip for both)
    float hkk[L], mkk[L], Wt[TRI];
    const bool student = a.nu != nullptr;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const float hv = a.hk[kc * L + i], mv = a.mk[kc * L + i];
        hkk[i] = lane_on ? hv : 0.f;
        mkk[i] = lane_on ? mv : 0.f;
#pragma unroll
        for (int j = 0; j <= i; ++j) { const float wv = a.Wk[(kc * L + i) * L + j]; Wt[tri(i, j)] = lane_on ? wv : 0.f; }
    }
    const float nuv = *(student ? a.nu + kc : a.bias);
    const float nuk = (student && lane_on) ? nuv : 1.f;

    for (int e = threadIdx.x; e < K * TRI; e += blockDim.x) {
        const int kk = e / TRI, idx = e - kk * TRI;
        int i = 0;
        while (tri(i + 1, 0) <= idx) ++i;
        const int j = idx - tri(i, 0);
        pk_lds[kk * PSTR + idx] = a.Pk[(kk * L + i) * L + j];
    }
    __syncthreads();
    SV_TS(2);
    SV_USE(hkk[0] + Wt[0] + nuk); SV_TS(3);
    for (int i = 0; i < PWa; ++i) accl[i * AST + lane] = 0.f;  // sums over this lane's cells (fixed k)
    SV_TS(4);

    bool first_tile = !ONE;
    for (long long t = (long long)blockIdx.x * nw + wave; t < ntiles; t += (long long)gridDim.x * nw) {
        const long long row = t * RPT + r;
        const bool on = lane_on && row < a.N;
        const long long rowc = on ? row : 0;
        const long long cellid = rowc * K + kc;             // clamped: always a valid cell

        float Lm[TRI], av[L], mu[L];
#pragma unroll
        for (int i = 0; i < TRI; ++i) Lm[i] = lane_on ? pk_lds[k * PSTR + i] : 0.f;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const float e1v = ONE ? pe1[i] : a.eta1[rowc * L + i], e2v = ONE ? pe2[i] : a.eta2d[rowc * L + i];
            const float e1 = on ? e1v : 0.f;
            const float e2 = on ? e2v : -0.5f;
            Lm[tri(i, i)] = fmaf(-2.f, e2, lane_on ? Lm[tri(i, i)] : 0.f);
            av[i] = e1 + hkk[i];
        }
        float ld;
        cell_cholesky<L>(Lm, ld);
        solve_lower<L>(Lm, av);
#pragma unroll
        for (int i = 0; i < L; ++i) mu[i] = av[i];
        solve_lower_t<L>(Lm, mu);                           // mu~ = Pt^-1 ht
        SV_USE(mu[0]); SV_TS(5);

        const float glzv = ONE ? pg[0] : a.Glz[cellid], gTv = ONE ? pg[1] : a.GT[cellid], lzv = ONE ? pg[2] : a.lz[cellid];
        const float glz = on ? glzv : 0.f;
        const float gT = on ? gTv : 0.f;
        const float rnk = on ? __expf(lzv) : 0.f;
        const float gsum = row_sum(glz, scr, lane, rbase, K);
        const float Gc = glz - rnk * gsum;                  // through the log-sum-exp normalisation
        const float Gld = gT - Gc;                          // T' has +ld, c has -ld
        SV_USE(Gld); SV_TS(6);

        float Wsum[L], M[TRI];
#pragma unroll
        for (int i = 0; i < L; ++i) Wsum[i] = 0.f;
#pragma unroll
        for (int i = 0; i < TRI; ++i) M[i] = 0.f;
        const float gts = gT * invS;
        const float* __restrict__ xc = a.x + cellid * LSn;
        const float* __restrict__ gc = a.Gx + cellid * LSn;
        // nxs / ngx: the next pair, in flight while this one is used.  The FIRST pair of a tile was requested before the
        // previous tile's assembly phase (below), so its latency is covered by ~4 k cycles of arithmetic instead of being
        // exposed behind the Cholesky of every tile.
        if (first_tile) { load_pair(xc, gc, on, 0, nxs, ngx); first_tile = false; }
        for (int s0 = 0; s0 < S; s0 += 2) {
            float xp[2 * L], gp[2 * L];
#pragma unroll
            for (int i = 0; i < 2 * L; ++i) { xp[i] = nxs[i]; gp[i] = ngx[i]; }
            if constexpr (ONE) {
#pragma unroll
                for (int i = 0; i < 2 * L; ++i) { nxs[i] = nxs2[i]; ngx[i] = ngx2[i]; }
                if (s0 + 4 < S) load_pair(xc, gc, on, s0 + 4, nxs2, ngx2);
            } else {
                if (s0 + 2 < S) load_pair(xc, gc, on, s0 + 2, nxs, ngx);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
            if (h == 1 && s0 + 1 >= S) break;
            float xs[L], gx[L];
#pragma unroll
            for (int i = 0; i < L; ++i) { xs[i] = xp[h * L + i]; gx[i] = gp[h * L + i]; }
            // d/dx of the theta term of T':  (1/S) c_s W^T W (x - m),  c_s = 1 (Gaussian) or (nu+L)/(nu+delta^2)
            float d[L], y[L];
#pragma unroll
            for (int i = 0; i < L; ++i) d[i] = xs[i] - mkk[i];
            float del2 = 0.f;
#pragma unroll
            for (int i = 0; i < L; ++i) {
                float yy = 0.f;
#pragma unroll
                for (int j = 0; j <= i; ++j) yy = fmaf(Wt[tri(i, j)], d[j], yy);
                y[i] = yy;
                del2 = fmaf(yy, yy, del2);
            }
            const float gc = student ? gts * (nuk + (float)L) / (nuk + del2) : gts;
#pragma unroll
            for (int i = 0; i < L; ++i) {
                const float gy = gc * y[i];
#pragma unroll
                for (int j = 0; j <= i; ++j) gx[j] = fmaf(Wt[tri(i, j)], gy, gx[j]);
            }
            if (student && on) {
                // theta is trainable in the SMM model: d/dm = -(the x-gradient of the theta term), d/dW = c y d^T
                float tx[L];
#pragma unroll
                for (int j = 0; j < L; ++j) tx[j] = 0.f;
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    const float gy = gc * y[i];
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        tx[j] = fmaf(Wt[tri(i, j)], gy, tx[j]);
                        accl[(TH + L + tri(i, j)) * AST + lane] += gy * d[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < L; ++j) accl[(TH + j) * AST + lane] -= tx[j];
            }
            solve_lower<L>(Lm, gx);                         // w_s = Lt^-1 gx_s
#pragma unroll
            for (int i = 0; i < L; ++i) {
                Wsum[i] += gx[i];
                const float e = xs[i] - mu[i];              // e_s = Lt^-T eps_s
#pragma unroll
                for (int j = 0; j <= i; ++j) M[tri(i, j)] = fmaf(e, gx[j], M[tri(i, j)]);
            }
            }
        }
        SV_USE(Wsum[0] + M[0]); SV_TS(7);
        if constexpr (!ONE) {   // first sample pair of this wave's NEXT tile
            const long long tn = t + (long long)gridDim.x * nw;
            const long long rown = tn * RPT + r;
            const bool onn = lane_on && tn < ntiles && rown < a.N;
            const long long celln = (onn ? rown : 0) * K + kc;
            load_pair(a.x + celln * LSn, a.Gx + celln * LSn, onn, 0, nxs, ngx);
        }
        // ---- assemble dLoss/dht and dLoss/dPt (symmetric, lower triangle)
        float V[L];
#pragma unroll
        for (int i = 0; i < L; ++i) V[i] = Wsum[i];
        solve_lower_t<L>(Lm, V);                            // V = Pt^-1 sum_s gx_s
        float gh[L];
#pragma unroll
        for (int i = 0; i < L; ++i) gh[i] = fmaf(Gc, mu[i], V[i]);

        // Cholesky adjoint with G_L = -tril(M):  A = Lt^T G_L (lower part), B = Phi(A), C = B + B^T + Gld I
        // real diagonal of Lt is 1/Lm[tri(j,j)]
        float Cs[TRI];                                      // symmetric C, lower triangle
        float dg[L];
#pragma unroll
        for (int i = 0; i < L; ++i) dg[i] = 1.0f / Lm[tri(i, i)];
#pragma unroll
        for (int i = 0; i < L; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                float s2 = 0.f;
#pragma unroll
                for (int p = i; p < L; ++p) {
                    const float lpi = (p == i) ? dg[i] : Lm[tri(p, i)];
                    s2 = fmaf(lpi, -M[tri(p, j)], s2);
                }
                Cs[tri(i, j)] = (i == j) ? (s2 + Gld) : s2;   // Phi halves the diagonal, B + B^T doubles it back
            }
        // Y = Lt^-1 (lower, explicit);  g = 1/2 Y^T C Y
        float Y[TRI];
#pragma unroll
        for (int j = 0; j < L; ++j) {
            Y[tri(j, j)] = Lm[tri(j, j)];
#pragma unroll
            for (int i = j + 1; i < L; ++i) {
                float s2 = 0.f;
#pragma unroll
                for (int p = j; p < i; ++p) s2 = fmaf(Lm[tri(i, p)], Y[tri(p, j)], s2);
                Y[tri(i, j)] = -s2 * Lm[tri(i, i)];
            }
        }
        float gP[TRI];
#pragma unroll
        for (int i = 0; i < TRI; ++i) gP[i] = 0.f;
        // Z = C Y column by column (Z[:,j] needs Y[p][j], p >= j), then gP[i][j] = 1/2 sum_{p>=i} Y[p][i] Z[p][j]
#pragma unroll
        for (int j = 0; j < L; ++j) {
            float Zc[L];
#pragma unroll
            for (int q = 0; q < L; ++q) {
                float s2 = 0.f;
#pragma unroll
                for (int p = j; p < L; ++p) {
                    const float cqp = (q >= p) ? Cs[tri(q, p)] : Cs[tri(p, q)];
                    s2 = fmaf(cqp, Y[tri(p, j)], s2);
                }
                Zc[q] = s2;
            }
#pragma unroll
            for (int i = j; i < L; ++i) {
                float s2 = 0.f;
#pragma unroll
                for (int p = i; p < L; ++p) s2 = fmaf(Y[tri(p, i)], Zc[p], s2);
                gP[tri(i, j)] = 0.5f * s2;
            }
        }
        // rank-one terms:  - sym(V mu^T) - 1/2 Gc mu mu^T
#pragma unroll
        for (int i = 0; i < L; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j)
                gP[tri(i, j)] += -0.5f * (V[i] * mu[j] + mu[i] * V[j]) - 0.5f * Gc * mu[i] * mu[j];
        SV_USE(gP[0]); SV_TS(8);

        // ---- per-component sums (registers) and per-row sums (through LDS)
        if (on) {
#pragma unroll
            for (int i = 0; i < L; ++i) accl[i * AST + lane] += gh[i];
#pragma unroll
            for (int i = 0; i < TRI; ++i) accl[(L + i) * AST + lane] += gP[i];
            accl[(L + TRI) * AST + lane] += Gc;
            if (student) accl[(TH + L + TRI) * AST + lane] -= gT;   // T' has -kappa_k
        }
        if (K == 16 || !ONE) {                              // K = 16: a row is one 16-lane DPP row.  Streaming sizes keep the
                                                            // call-per-value form: the batched form below measured 13 % SLOWER
                                                            // there (K = 10, N = 1e6: 2.07 -> 2.38 ms) - it pays at one tile per wave
                                                            // (so did a form with the row-leader lanes summing from the batched
                                                            // scratch: 2.12 -> 2.45 ms; the kernel sits at the 256-VGPR limit)
#pragma unroll
            for (int i = 0; i < L; ++i) {
                const float s1 = row_sum(on ? gh[i] : 0.f, scr, lane, rbase, K);
                const float s2 = row_sum(on ? gP[tri(i, i)] : 0.f, scr, lane, rbase, K);
                if (on && k == 0) {
                    a.g_eta1[row * L + i] = s1;
                    a.g_eta2d[row * L + i] = -2.f * s2;      // p = -2 eta2d
                }
            }
        } else {
            // all 2L values of every lane go to LDS at once; sum (row rr, quantity i) is formed by lane q = rr * 2L + i.
            // (2L row_sum calls one after the other were 2L dependent LDS round trips: 4.3 us of a 28 us launch at N = 64)
#pragma unroll
            for (int i = 0; i < L; ++i) {
                rows[i * AST + lane] = on ? gh[i] : 0.f;
                rows[(L + i) * AST + lane] = on ? gP[tri(i, i)] : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            for (int q0 = lane; q0 < 2 * L * RPT; q0 += 2 * WAVE) {     // two sums per lane: independent LDS reads in flight
                const int q1 = q0 + WAVE;
                const bool has1 = q1 < 2 * L * RPT;
                const int rr0 = q0 / (2 * L), i0 = q0 - rr0 * (2 * L);
                const int rr1 = has1 ? q1 / (2 * L) : rr0, i1 = has1 ? q1 - rr1 * (2 * L) : i0;
                const float* __restrict__ p0 = rows + i0 * AST + rr0 * K;
                const float* __restrict__ p1 = rows + i1 * AST + rr1 * K;
                float sq0 = 0.f, sq1 = 0.f;
#pragma unroll 4
                for (int j = 0; j < K; ++j) { sq0 += p0[j]; sq1 += p1[j]; }
                const long long row0 = t * RPT + rr0, row1 = t * RPT + rr1;
                if (row0 < a.N) {
                    if (i0 < L) a.g_eta1[row0 * L + i0] = sq0;
                    else a.g_eta2d[row0 * L + (i0 - L)] = -2.f * sq0;   // p = -2 eta2d
                }
                if (has1 && row1 < a.N) {
                    if (i1 < L) a.g_eta1[row1 * L + i1] = sq1;
                    else a.g_eta2d[row1 * L + (i1 - L)] = -2.f * sq1;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        SV_TS(9);
    }

    // ---- block reduction of the per-component sums: lanes with equal k of all waves, fixed order (row, then wave).
    // (The waves used to fold their accumulators into one array one after the other - nw rounds of PWa dependent LDS
    //  read-modify-writes between barriers: 8 us of a 28 us launch at N = 64.)
    __syncthreads();
    SV_TS(10);
    float* out = a.partials + (long long)blockIdx.x * K * PW;
    constexpr int EPT = 4;                                  // elements per thread and round: EPT * SV_NW independent LDS reads in flight
    for (int e0 = threadIdx.x; e0 < K * PW; e0 += SV_NW * WAVE * EPT) {
        float s2[EPT];
        int off[EPT];
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int e = e0 + u * SV_NW * WAVE;
            const int kk = e / PW, f = e - kk * PW;
            off[u] = (e < K * PW && f < PWa) ? f * AST + kk : -1;
            s2[u] = 0.f;
        }
        for (int rr = 0; rr < RPT; ++rr) {
            float v[SV_NW][EPT];
#pragma unroll
            for (int w = 0; w < SV_NW; ++w)
#pragma unroll
                for (int u = 0; u < EPT; ++u) v[w][u] = acc0[w * (PWa * AST) + (off[u] < 0 ? 0 : off[u]) + rr * K];
#pragma unroll
            for (int w = 0; w < SV_NW; ++w)
#pragma unroll
                for (int u = 0; u < EPT; ++u) s2[u] += v[w][u];
        }
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int e = e0 + u * SV_NW * WAVE;
            if (e < K * PW) out[e] = off[u] < 0 ? 0.f : s2[u];
        }
    }
    SV_TS(11);
}




// ---------------------------------------------------------------------------------------------------------
// subsample_x (reference svae.py:122-151): z_ns ~ Cat(exp log_z_n), x_samples[n,s,:] = x[n, z_ns, s, :].
// The categorical draw is the inverse CDF of a supplied uniform (or a supplied index, for parity tests).
// ---------------------------------------------------------------------------------------------------------
// Forward kernel, packed form: the sample loop handles TWO samples per iteration as the two halves of
// v_pk_*_f32 operands ({eps_i,s ; eps_i,s+1} are adjacent in the (cell, L, S) noise layout, so one ds_read2_b32
// delivers the pair), with the cell factor / theta parameters broadcast by op_sel (vmp_common.h).  P_k comes
// from an LDS table once per tile instead of living in 36 VGPRs.  ST is a compile-time S (0 = run-time): with
// it every LDS offset in the sample loop is an immediate.  Full tiles take check-free staging / copy-out loops
// whose (cell, offset) indices advance incrementally.
template <int L, int ST>
__global__ __launch_bounds__(SV_FWD_MAX_NW * WAVE) void svae_estep_fwd3_kernel(EFwdArgs a) {
    constexpr int TRI = SvGeo<L>::TRI;
    constexpr int TP = (TRI + 1) / 2, LP = (L + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int K = a.K;
    const int S = ST ? ST : a.S;
    const int LSn = L * S, CSTR = LSn | 1;                 // odd LDS stride per cell
    const int RPT = WAVE / K, CT = RPT * K;                // rows / cells per wave tile
    constexpr int PSTR = TRI | 1;
    float* pk_lds = smem;                                  // [K][PSTR] lower triangle of P_k
    float* et = smem + K * PSTR + wave * (WAVE * CSTR + WAVE);
    float* scr = et + WAVE * CSTR;
    const bool lane_on = lane < CT;
    const int r = lane / K, k = lane - r * K, rbase = lane_on ? r * K : 0;
    const bool k16 = (K == 16);                            // lane row of 16 = one data row: DPP reductions

    for (int e = threadIdx.x; e < K * TRI; e += blockDim.x) {
        const int kk = e / TRI, idx = e - kk * TRI;
        int i = 0;
        while (tri(i + 1, 0) <= idx) ++i;
        const int j = idx - tri(i, 0);
        pk_lds[kk * PSTR + idx] = a.Pk[(kk * L + i) * L + j];
    }
    __syncthreads();

    // resident component parameters (unconditional loads from clamped indices + value selects)
    float hkk[L];
    v2f mk2[LP], Wt2[TP];
    const bool student = a.nu != nullptr;
    const int kc = lane_on ? k : 0;
#pragma unroll
    for (int i = 0; i < 2 * LP; ++i) {
        const float mv = a.mk[kc * L + (i < L ? i : 0)];
        mk2[i >> 1][i & 1] = (lane_on && i < L) ? mv : 0.f;
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const float hv = a.hk[kc * L + i];
        hkk[i] = lane_on ? hv : 0.f;
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const float wv = a.Wk[(kc * L + i) * L + j];
            Wt2[tri(i, j) >> 1][tri(i, j) & 1] = lane_on ? wv : 0.f;
        }
    }
    if (TRI & 1) Wt2[TP - 1][1] = 0.f;
    float biask, kappak, nuk;
    {
        const float bv = a.bias[kc], kv = a.kappa[kc], nv = *(student ? a.nu + kc : a.bias);
        biask = lane_on ? bv : 0.f; kappak = lane_on ? kv : 0.f; nuk = (student && lane_on) ? nv : 1.f;
    }
    const float inv_nu = 1.0f / nuk;

    const long long ntiles = (a.N + RPT - 1) / RPT;
    const float invLS = 1.0f / (float)LSn, invS = 1.0f / (float)S;
    const int Q = LSn >> 2;                                // float4s per cell (fast paths need LSn % 4 == 0)
    const bool fastio = a.vec_ok && (LSn & 3) == 0 && (L & 3) == 0 && CT == WAVE;
    const int c_first = lane / (Q > 0 ? Q : 1), rem_first = lane - c_first * Q;
    const int dc = WAVE / (Q > 0 ? Q : 1), dr = WAVE - dc * Q;
    for (long long t = (long long)blockIdx.x * nw + wave; t < ntiles; t += (long long)gridDim.x * nw) {
        const long long row = t * RPT + r;
        const bool on = lane_on && row < a.N;
        const long long rows_here = (a.N - t * RPT) < RPT ? (a.N - t * RPT) : RPT;
        const int tot = (int)rows_here * K * LSn;          // floats of noise / samples in this tile
        const bool full = fastio && rows_here == RPT;
        // ---- stage the noise tile: coalesced global reads, padded per-cell layout in LDS
        {
            const float* __restrict__ g = a.noise + t * CT * LSn;
            if (full) {
                // every lane moves exactly Q float4s; float4 q = it*64 + lane belongs to cell q / Q at offset 4*(q % Q)
                const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g) + lane;
                int c = c_first, rem = rem_first;
                constexpr int UN = 10;
                for (int it0 = 0; it0 < Q; it0 += UN) {
                    float4 v[UN];
#pragma unroll
                    for (int u = 0; u < UN; ++u)
                        if (it0 + u < Q) v[u] = g4[(it0 + u) * WAVE];
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        if (it0 + u < Q) {
                            float* d = et + c * CSTR + 4 * rem;
                            d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                            c += dc; rem += dr;
                            if (rem >= Q) { rem -= Q; c += 1; }
                        }
                    }
                }
            } else if (a.vec_ok && (LSn & 3) == 0) {
                constexpr int UN = 10;
                for (int e0 = 4 * lane; e0 < tot; e0 += UN * 4 * WAVE) {
                    float4 v[UN];
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        const int e = e0 + u * 4 * WAVE;
                        v[u] = (e < tot) ? *reinterpret_cast<const float4*>(g + e) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        const int e = e0 + u * 4 * WAVE;
                        if (e < tot) {
                            const int c = (int)(((float)e + 0.5f) * invLS);
                            const int j = e - c * LSn;
                            float* d = et + c * CSTR + j;
                            d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                        }
                    }
                }
            } else {
                for (int e = lane; e < tot; e += WAVE) {
                    const int c = (int)(((float)e + 0.5f) * invLS);
                    et[c * CSTR + (e - c * LSn)] = g[e];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- cell factorisation
        float Lm[TRI], av[L];
        const long long rowc = on ? row : 0;
#pragma unroll
        for (int i = 0; i < TRI; ++i) { const float pvv = pk_lds[kc * PSTR + i]; Lm[i] = lane_on ? pvv : 0.f; }
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const float e1v = a.eta1[rowc * L + i], e2v = a.eta2d[rowc * L + i];
            const float e1 = on ? e1v : 0.f;
            const float e2 = on ? e2v : -0.5f;
            Lm[tri(i, i)] = fmaf(-2.f, e2, Lm[tri(i, i)]);
            av[i] = e1 + hkk[i];
        }
        float ld;
        cell_cholesky<L>(Lm, ld);
        solve_lower<L>(Lm, av);
        float aa = 0.f;
#pragma unroll
        for (int i = 0; i < L; ++i) aa = fmaf(av[i], av[i], aa);
        const float c = on ? (biask + 0.5f * aa - ld) : -INFINITY;
        float mx, se, ex;
        if (k16) {
            mx = row16_max(c);
            ex = on ? __expf(c - mx) : 0.f;
            se = row16_sum(ex);
        } else {
            mx = row_max(c, scr, lane, rbase, K);
            ex = on ? __expf(c - mx) : 0.f;
            se = row_sum(ex, scr, lane, rbase, K);
        }
        const float lz = c - mx - __logf(se);

        // factor and a = Lt^-1 ht as broadcastable pairs
        v2f Lm2[TP], av2[LP];
#pragma unroll
        for (int i = 0; i < 2 * TP; ++i) Lm2[i >> 1][i & 1] = (i < TRI) ? Lm[i < TRI ? i : 0] : 0.f;
#pragma unroll
        for (int i = 0; i < 2 * LP; ++i) av2[i >> 1][i & 1] = (i < L) ? av[i < L ? i : 0] : 0.f;

        // ---- samples (two at a time) and the per-cell regulariser term
        v2f eps2 = v2f{0.f, 0.f}, qth = v2f{0.f, 0.f};
        float* cell = et + lane * CSTR;
#pragma unroll 1
        for (int s = 0; s < S; s += 2) {
            const bool hv = s + 1 < S;                     // second half valid (uniform)
            v2f z[L];
#pragma unroll
            for (int i = 0; i < L; ++i) {
                v2f e = v2f{cell[i * S + s], cell[i * S + s + 1]};   // the element after the last one is padding / scratch
                if (!lane_on) e = v2f{0.f, 0.f};
                if (!hv) e.y = 0.f;
                eps2 = __builtin_elementwise_fma(e, e, eps2);
                z[i] = pk_add_b(e, av2[i >> 1], i & 1);
            }
            // z <- Lt^-T z (back substitution; diagonal of Lm holds reciprocals)
#pragma unroll
            for (int i = L - 1; i >= 0; --i) {
                v2f tt = z[i];
#pragma unroll
                for (int p2 = i + 1; p2 < L; ++p2) tt = pk_fnma_b(z[p2], Lm2[tri(p2, i) >> 1], tt, tri(p2, i) & 1);
                z[i] = pk_mul_b(tt, Lm2[tri(i, i) >> 1], tri(i, i) & 1);
            }
            v2f d[L];
#pragma unroll
            for (int i = 0; i < L; ++i) d[i] = pk_sub_b(z[i], mk2[i >> 1], i & 1);
            v2f del2 = v2f{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < L; ++i) {
                v2f y = pk_mul_b(d[0], Wt2[tri(i, 0) >> 1], tri(i, 0) & 1);
#pragma unroll
                for (int j = 1; j <= i; ++j) y = pk_fma_b(d[j], Wt2[tri(i, j) >> 1], y, tri(i, j) & 1);
                del2 = __builtin_elementwise_fma(y, y, del2);
            }
            if (!hv) del2.y = 0.f;
            if (student) {
                const float sc = nuk + (float)L;
                qth.x += sc * log1p_f(del2.x * inv_nu);
                qth.y += sc * log1p_f(del2.y * inv_nu);
            } else {
                qth += del2;
            }
            if (lane_on) {
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    cell[i * S + s] = z[i].x;
                    if (hv) cell[i * S + s + 1] = z[i].y;
                }
            }
        }
        if (on) {
            a.lz[row * K + k] = lz;
            a.Tp[row * K + k] = -0.5f * L * LOG_2PI + ld - 0.5f * invS * (eps2.x + eps2.y) + 0.5f * invS * (qth.x + qth.y) - kappak;
        }
        __builtin_amdgcn_wave_barrier();

        // ---- samples out: (cell, S, L) layout, coalesced
        {
            float* __restrict__ g = a.x + t * CT * LSn;
            if (full) {
                // float4 q = it*64 + lane of the tile = (cell q / Q, sample (q % Q) / (L/4), 4 dims from 4*((q % Q) % (L/4)))
                float4* __restrict__ g4 = reinterpret_cast<float4*>(g) + lane;
                int c2 = c_first, rem = rem_first;
                constexpr int L4 = (L / 4 > 0) ? L / 4 : 1;
                for (int it = 0; it < Q; ++it) {
                    const int s = rem / L4, l4 = rem - s * L4;
                    const float* src = et + c2 * CSTR + (4 * l4) * S + s;
                    float4 v;
                    v.x = src[0]; v.y = src[S]; v.z = src[2 * S]; v.w = src[3 * S];
                    g4[it * WAVE] = v;
                    c2 += dc; rem += dr;
                    if (rem >= Q) { rem -= Q; c2 += 1; }
                }
            } else if (a.vec_ok && (L & 3) == 0) {
                for (int o = 4 * lane; o < tot; o += 4 * WAVE) {
                    const int c2 = (int)(((float)o + 0.5f) * invLS);
                    const int rem = o - c2 * LSn;
                    const int s = rem / L, l = rem - s * L;
                    const float* src = et + c2 * CSTR + l * S + s;
                    float4 v;
                    v.x = src[0]; v.y = src[S]; v.z = src[2 * S]; v.w = src[3 * S];
                    *reinterpret_cast<float4*>(g + o) = v;
                }
            } else {
                for (int o = lane; o < tot; o += WAVE) {
                    const int c2 = (int)(((float)o + 0.5f) * invLS);
                    const int rem = o - c2 * LSn;
                    const int s = rem / L, l = rem - s * L;
                    g[o] = et[c2 * CSTR + l * S + s];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------
// In-kernel noise (reference models/svae.py:113-114 draws eps inside the step with TensorFlow's Philox stream).
// Generator: Philox4x32-7 - Random123's philox4x32 at R = 7 rounds, the smallest round count its authors report as
// Crush-resistant (Salmon et al., SC'11, table 2; R = 10 is their safety-margin default) - keyed by the seed, pinned by the
// Random123 known-answer vectors for 7 AND 10 rounds (tests/test_philox.py).  Counter = (cell id low, cell id high, block, 0) with
// cell = n K + k.  Round 6: one 128-bit block yields FOUR Box-Muller pairs, one per 32-bit word (round 5: three pairs of 21 + 21
// bits): the word's top 20 bits are the radius uniform u = (a + 1/2) 2^-20 in (0,1) (|eps| <= 5.4, as before), its low 12 bits
// the angle (4096 directions; each normal's marginal is a 4096-point periodic trapezoid rule over the angle of a smooth function
// of the radius - exact to rounding).  Block b = p ceil(L/4) + j of a cell holds, for the sample pair (2p, 2p+1), coordinates
// i = 4j .. 4j+3; word t of the block = (eps[4j+t, 2p], eps[4j+t, 2p+1]) = r (cos, sin): at L = 8 a sample pair is 2 blocks
// instead of 3 with one pair unused (28 instead of 42 v_mad_u64_u32 per 16 normals, no cross-word bit extraction).  Stateless: any
// kernel (or the host oracle, oracle/philox.py) can regenerate the same element from (seed, n, k, i, s).
// ---------------------------------------------------------------------------------------------------------
#ifndef VMP_PHILOX_ROUNDS
#define VMP_PHILOX_ROUNDS 7
#endif
template <int R>
__device__ __forceinline__ void philox4x32(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
        // a ^ b ^ c in ONE instruction: v_bitop3_b32 with truth table 0x96 (gfx950)
        const unsigned n0 = __builtin_amdgcn_bitop3_b32((unsigned)(p1 >> 32), c[1], k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32((unsigned)(p0 >> 32), c[3], k1, 0x96);
        c[1] = (unsigned)p1; c[3] = (unsigned)p0; c[0] = n0; c[2] = n2;
    }
}
// one 32-bit word -> one Box-Muller pair: radius from the top 20 bits, angle from the low 12.
// TAB: (cos, sin) of the 4096 directions come from an LDS table the block fills at start WITH THE SAME v_cos / v_sin instructions
// (bit-identical to the direct form) - one ds_read_b64 instead of two quarter-rate transcendentals per pair: the sample loop of the
// forward kernel is bound by VALU issue, and of its ~1.5 k cycles per sample pair 512 were v_log / v_sqrt / v_sin / v_cos.
#ifndef VMP_FWD_SINCOS_TAB
#define VMP_FWD_SINCOS_TAB 1
#endif
constexpr int SCT_WORDS = 2 * 4096;
__device__ __forceinline__ float bm_angle(unsigned b12) { return __uint_as_float((b12 << 11) | 0x3F800000u); }   // 1 + b 2^-12 revolutions
template <bool TAB>
__device__ __forceinline__ v2f box_muller_word(unsigned w, const float* __restrict__ sct) {
    const float u1 = fmaf((float)(w >> 12), 9.5367431640625e-07f, 4.76837158203125e-07f);   // (a + 1/2) 2^-20 in (0, 1)
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1), v_log_f32 = log2
    if constexpr (TAB) {
        const v2f cs = *reinterpret_cast<const v2f*>(sct + ((w << 1) & 0x1FFEu));
        return v2f{rad, rad} * cs;
    } else {
        // angle in revolutions: the 12 bits become the top mantissa bits of a float in [1, 2) - v_sin / v_cos take revolutions and
        // are periodic, so 1 + b 2^-12 is as good as b 2^-12 (shift + v_and_or instead of v_cvt + v_mul; exact either way)
        const float ang = __uint_as_float(((w << 11) & 0x007FF800u) | 0x3F800000u);
        return v2f{rad * __builtin_amdgcn_cosf(ang), rad * __builtin_amdgcn_sinf(ang)};
    }
}
// the four pairs of block `blk` of cell `cell`: coordinates 4j .. 4j+3 of sample pair p (blk = p ceil(L/4) + j)
template <bool TAB = false>
__device__ __forceinline__ void philox_normal8(unsigned long long cell, unsigned blk, unsigned long long seed, v2f (&p)[4],
                                               const float* __restrict__ sct = nullptr) {
    unsigned c[4] = {(unsigned)cell, (unsigned)(cell >> 32), blk, 0u};
    philox4x32<VMP_PHILOX_ROUNDS>(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
    for (int t = 0; t < 4; ++t) p[t] = box_muller_word<TAB>(c[t], sct);
}

struct NoiseArgs { float* out; long long cells; int L, S; unsigned long long seed; const unsigned long long* seed_dev; };
// Materialises the same stream as a (cells, L, S) tensor: for shapes the in-kernel path does not cover, and for tests.
__global__ __launch_bounds__(256) void philox_noise_kernel(NoiseArgs a) {
    const int SP = (a.S + 1) >> 1, L4 = (a.L + 3) / 4, NB = SP * L4;
    const long long total = a.cells * NB;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long cell = e / NB;
        const int b = (int)(e - cell * NB), sp = b / L4, j = b - sp * L4;
        v2f pr[4];
        philox_normal8((unsigned long long)cell, (unsigned)b, a.seed_dev ? *a.seed_dev : a.seed, pr);
        float* o = a.out + cell * a.L * a.S;
        const int s2 = 2 * sp;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = 4 * j + t;
            if (i < a.L) {
                o[i * a.S + s2] = pr[t].x;
                if (s2 + 1 < a.S) o[i * a.S + s2 + 1] = pr[t].y;
            }
        }
    }
}

// Forward kernel, LDS-DMA form.  Measured on the single-buffered kernels above (C3): loads+compute 1.3 ms,
// compute+stores 1.4 ms, everything 2.8 ms - a wave's load phase and store phase do not overlap, and 7 waves per CU
// with <=10 KB each in flight cannot hide it.  Here every wave owns TWO noise buffers: the next tile streams in with
// global_load_lds_dwordx4 (no VGPRs, a full 20 KB tile in flight per wave) while the current one is computed and
// written back.  The LDS image is lane-linear (dest = base + lane*16 B), so the padding goes into the SOURCE
// addresses: slot q = w*64 + lane of instruction w is float4 (q % QS) of cell (q / QS), QS = CS/4 with CS/4 odd,
// i.e. a cell stride of CS dwords whose 64-bit accesses are 2-way conflict-free at worst; pad slots re-fetch the
// cell's last float4.  One s_waitcnt vmcnt(0) per tile retires the tile's DMA (issued a whole tile earlier) and
// the previous tile's stores; the row loads of the tile are consumed BEFORE the next DMA is issued, so no
// compiler-generated wait ever covers a DMA in flight.
// RNG = true: no noise tensor at all - eps is generated in registers (Philox4x32-7, above) right where it is consumed;
// the LDS tile then only serves the (cell, S, L) output transposition.
// Noise-tensor form: the tile's noise DMA carries nt (bit 1) and its sample stores are streaming stores (bit 0) - a tile's 20 KB are
// requested once, whole lines at a time, and its 20 KB of samples are not read again by this kernel.  Same box: K = 16 2.00 -> 1.88
// and 2.07 -> 1.91 ms, K = 10 1.36 -> 1.28, Student-t 2.50 -> 2.34; either bit alone: 0-3 %.  (The in-kernel-noise form keeps plain
// stores: VALU-bound, and at minibatch sizes the decoder reads the samples microseconds later.  The ring backward is the opposite
// case - the two halves of a 128-byte line are asked for by different DMA instructions and the second must hit in L2: nt there
// cost 40 %, csrc/vmp_svae_ring.h VMP_RING_NT.)
#ifndef VMP_T2_NT
#define VMP_T2_NT 3
#endif
template <bool NT>
__device__ __forceinline__ void st_x4(float4* p, const float4& v) {
    if constexpr (NT) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p));
    else *p = v;
}
#ifndef VMP_FWD_BC_TILE
#define VMP_FWD_BC_TILE 1
#endif
#ifndef VMP_PST_NOSTORE
#define VMP_PST_NOSTORE 0            // exploration builds: 1 = the pair-staging form without its global stores
#endif
#ifndef VMP_FWD_PAIR_STAGE_BELOW
#define VMP_FWD_PAIR_STAGE_BELOW 7     // the pair-staging form where the tile buffer admits fewer waves than this (round 4: 8)
#endif
#ifndef VMP_FWD_PST2_ALWAYS
#define VMP_FWD_PST2_ALWAYS 0         // exploration builds: the two-pair staging form also where the tile buffer admits eight waves
#endif
#ifndef VMP_FWD_PAIR_STAGE2
#define VMP_FWD_PAIR_STAGE2 1         // 0: A/B builds without the two-pair staging form
#endif
#ifndef VMP_FWD_PAIR_STAGE
#define VMP_FWD_PAIR_STAGE 1          // 0: A/B builds without the per-pair staging form
#endif
template <int L, int ST, bool RNG, bool PS = false>
__global__ __launch_bounds__((RNG ? 8 : 4) * WAVE) void svae_estep_fwd4_kernel(EFwdArgs a, int CS_rt) {
    SV_TS(23);
    constexpr int TRI = SvGeo<L>::TRI;
    constexpr int TP = (TRI + 1) / 2, LP = (L + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int K = a.K;
    const int S = ST ? ST : a.S;
    const int LSn = L * S;
    constexpr int CS_ct = ST ? (L * ST + ((((L * ST) >> 2) & 1) ? 0 : 4)) : 0;   // same rule as the host (CS/4 odd)
    const int CS = ST ? CS_ct : CS_rt;
    const int Q = LSn >> 2, QS = CS >> 2;                  // data / total float4 slots per cell
    constexpr int QSc = ST ? (CS_ct >> 2) : 1, Qc = ST ? ((L * ST) >> 2) : 1;
    const int RPT = WAVE / K, CT = RPT * K;
    constexpr int PSTR = TRI | 1;
    constexpr bool PST_ = RNG && L == 8 && PS;
    // pair-staging forms: h_k, bias_k, kappa_k of the lane's component are used once per tile - they come from an LDS table
    // [K][HKS] instead of living in 10 VGPRs across the sample loop (the round-6 epilogue needs those registers)
    constexpr int HKS = 12;
    constexpr bool SCT = PST_ && VMP_FWD_SINCOS_TAB;     // (cos, sin) of the generator's 4096 directions in LDS (box_muller_word)
    const int tab0 = (K * PSTR + 3) & ~3;
    const int tab1 = tab0 + (PST_ ? K * HKS : 0);
    const int tab = tab1 + (SCT ? SCT_WORDS : 0);
    float* pk_lds = smem;
    float* hk_lds = smem + tab0;
    float* sct = smem + tab1;
    // two tile buffers per wave (the noise of the next tile arrives by DMA while this one is processed); with in-kernel noise
    // nothing is prefetched: ONE buffer, which lets 7 waves instead of 4 share the LDS of a CU
    constexpr int NBUF = RNG ? 1 : 2;
    // cells per tile buffer: the buffers are sized by the tile's CT = (64 / K) K cells, which at K = 10 (60 cells) lets a FOURTH
    // wave of the noise-tensor form (an EIGHTH of the in-kernel-noise form) share the CU's 160 KB (host side: fwd4_plan, run_fwd)
    const int BC = VMP_FWD_BC_TILE ? CT : WAVE;
    // PST (in-kernel noise, L = 8): no tile buffer at all.  The samples of a pair leave through a 4 KB per-wave staging area in OUTPUT
    // order - lane = cell writes its four 16-byte pieces [sample][coordinates 0-3 | 4-7] with ds_write_b128, four adjacent lanes read
    // back the 64 contiguous bytes of ONE cell and store them (16 cells x 64 B per store instruction): 4 + 4 LDS instructions per pair
    // instead of 8 writes + 16 gathered 4-byte reads, and eight waves per CU where the tile buffer allows seven (K = 16, K = 9).
    // (Where the tile buffer already admits eight waves - K = 10, 12: 60-cell tiles - this form measured 0-5 % SLOWER, same box;
    // the host picks it only when it adds a wave: fwd4_plan.)  Piece j of cell c lies at position
    // j ^ ((c >> 1) & 3) of the cell's 64 bytes: every 8-lane group of both the writes and the reads covers all 32 banks.
    constexpr bool PST = PST_;
    constexpr bool PST2 = PST && ST != 0 && (ST & 3) == 2 && VMP_FWD_PAIR_STAGE2;       // two pairs per flush (S / 2 odd), below
    // OIMG (in-kernel noise, L = 8, compile-time even S, tile-buffer form): the LDS image of the tile is written in OUTPUT order
    // [cell][s][l] - four ds_write_b128 per sample pair (the pair-staging form's pieces; conflict-free at the padded cell stride:
    // eight consecutive cells start at eight different 16-byte slots of the 128-byte bank window) - and leaves by ds_read_b128 +
    // coalesced float4 stores: 20 + 20 LDS instructions per cell instead of 40 ds_write_b64 + 80 gathered ds_read_b32.
    constexpr bool OIMG = RNG && L == 8 && ST != 0 && (ST & 1) == 0 && !PST;
    // in-kernel moments (PST forms, K = 16): a [4 rows][XSEL] record of the tile's drawn samples per wave, in front of the buffers
    const bool momon = RNG && PST && a.mom != nullptr;
    const int xsel_words = momon ? nw * 4 * XSEL : 0;
    float* xsel = smem + tab + wave * (4 * XSEL);
    float* buf0 = smem + tab + xsel_words + wave * (PST2 ? 2 * (WAVE * 16 + 16) : PST ? WAVE * 16 : NBUF * BC * CS);
    const bool lane_on = lane < CT;
    const int r = lane / K, k = lane - r * K, rbase = lane_on ? r * K : 0;
    const bool k16 = (K == 16);
    const unsigned long long rng_seed = (RNG && a.seed_dev) ? *a.seed_dev : a.seed;
    SV_TS(16);


    float hkk[L];
    v2f mk2[LP], Wt2[TP];
    const bool student = a.nu != nullptr;
    const int kc = lane_on ? k : 0;
#pragma unroll
    for (int i = 0; i < 2 * LP; ++i) {
        const float mv = a.mk[kc * L + (i < L ? i : 0)];
        mk2[i >> 1][i & 1] = (lane_on && i < L) ? mv : 0.f;
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
        if constexpr (!PST_) {
            const float hv = a.hk[kc * L + i];
            hkk[i] = lane_on ? hv : 0.f;
        } else {
            hkk[i] = 0.f;
        }
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const float wv = a.Wk[(kc * L + i) * L + j];
            Wt2[tri(i, j) >> 1][tri(i, j) & 1] = lane_on ? wv : 0.f;
        }
    }
    if (TRI & 1) Wt2[TP - 1][1] = 0.f;
    float biask = 0.f, kappak = 0.f, nuk;
    {
        const float nv = *(student ? a.nu + kc : a.bias);
        nuk = (student && lane_on) ? nv : 1.f;
        if constexpr (!PST_) {
            const float bv = a.bias[kc], kv = a.kappa[kc];
            biask = lane_on ? bv : 0.f; kappak = lane_on ? kv : 0.f;
        }
    }
    const float inv_nu = 1.0f / nuk;
    SV_USE(inv_nu + hkk[0] + Wt2[0][0] + biask); SV_TS(18);

    const long long ntiles = (a.N + RPT - 1) / RPT;
    const float invLS = 1.0f / (float)LSn, invS = 1.0f / (float)S;
    const long long tstride = (long long)gridDim.x * nw;
    // slot walk of this lane: slot q = w*64 + lane -> (cell q / QS, slot q % QS), advanced incrementally
    const int c_first = lane / QS, sl_first = lane - c_first * QS;
    const int dcs = WAVE / QS, drs = WAVE - dcs * QS;
    // same walk over the DATA float4s (copy-out of full tiles)
    const int o_first = lane / Q, orem_first = lane - o_first * Q;
    const int dco = WAVE / Q, dro = WAVE - dco * Q;

    // per-lane source offset of DMA instruction w within a tile (floats); fixed for the whole kernel, so with a
    // compile-time S it is computed once and kept in registers (one wave per SIMD: there are 512 of them)
    int dma_off[QSc];
    if constexpr (ST != 0) {
        int c = c_first, sl = sl_first;
#pragma unroll
        for (int w = 0; w < QSc; ++w) {
            dma_off[w] = c * LSn + 4 * (sl < Q ? sl : Q - 1);
            c += dcs; sl += drs;
            if (sl >= QS) { sl -= QS; c += 1; }
        }
    }
    int co_off[Qc];                                        // LDS offset of output float4 (it*64 + lane) of a full tile
    if constexpr (ST != 0) {
        constexpr int L4c = (L / 4 > 0) ? L / 4 : 1;
        int c2 = o_first, rem = orem_first;
#pragma unroll
        for (int it = 0; it < Qc; ++it) {
            const int s = rem / L4c, l4 = rem - s * L4c;
            co_off[it] = OIMG ? c2 * CS + 4 * rem : c2 * CS + (4 * l4) * S + s;
            c2 += dco; rem += dro;
            if (rem >= Q) { rem -= Q; c2 += 1; }
        }
    }
    auto issue_dma = [&](long long tt, float* buf) {
        const long long cells_left = (a.N - tt * RPT) * K;
        const int ncell = cells_left < CT ? (int)cells_left : CT;          // valid cells of the tile (>= 1)
        const float* __restrict__ g = a.noise + tt * CT * LSn;
        if constexpr (ST != 0) {
            if (ncell == WAVE) {
#pragma unroll
                for (int w = 0; w < QSc; ++w)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + dma_off[w]),
                                                     (__attribute__((address_space(3))) void*)(buf + w * (4 * WAVE)), 16, 0, (VMP_T2_NT & 2) ? 2 : 0);
                return;
            }
        }
        int c = c_first, sl = sl_first;
        for (int w = 0; w < QS; ++w) {
            const int cc = c < ncell ? c : ncell - 1;
            const int ss = sl < Q ? sl : Q - 1;
            const float* src = g + (long long)cc * LSn + 4 * ss;
            if (c < BC)                                     // slots past the buffer's last cell (BC < 64) are not written
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(buf + w * (4 * WAVE)), 16, 0, (VMP_T2_NT & 2) ? 2 : 0);
            c += dcs; sl += drs;
            if (sl >= QS) { sl -= QS; c += 1; }
        }
    };

    long long t = (long long)blockIdx.x * nw + wave;
    if constexpr (!RNG) { if (t < ntiles) issue_dma(t, buf0); }
    int cur = 0;
    // the encoder rows of a tile are fetched one tile ahead as well (plain loads, consumed before the next DMA is issued)
    float e1r[L], e2r[L];
    {
        const long long row0 = t * RPT + r;
        const long long rc0 = (t < ntiles && lane_on && row0 < a.N) ? row0 : 0;
#pragma unroll
        for (int i = 0; i < L; ++i) { e1r[i] = a.eta1[rc0 * L + i]; e2r[i] = a.eta2d[rc0 * L + i]; }
    }
    // (the P_k table is staged AFTER the first tile's requests are out: one memory round trip for all of them)
    for (int e = threadIdx.x; e < K * TRI; e += blockDim.x) {
        const int kk = e / TRI, idx = e - kk * TRI;
        int i = 0;
        while (tri(i + 1, 0) <= idx) ++i;
        const int j = idx - tri(i, 0);
        pk_lds[kk * PSTR + idx] = a.Pk[(kk * L + i) * L + j];
    }
    if constexpr (PST_) {
        for (int e = threadIdx.x; e < K * HKS; e += blockDim.x) {
            const int kk = e / HKS, i = e - kk * HKS;
            hk_lds[e] = i < L ? a.hk[kk * L + i] : i == L ? a.bias[kk] : i == L + 1 ? a.kappa[kk] : 0.f;
        }
    }
    if constexpr (SCT) {
        for (int e = threadIdx.x; e < 4096; e += blockDim.x) {
            const float ang = bm_angle((unsigned)e);
            *reinterpret_cast<v2f*>(sct + 2 * e) = v2f{__builtin_amdgcn_cosf(ang), __builtin_amdgcn_sinf(ang)};
        }
    }
    __syncthreads();
    SV_TS(17);
    // ---- epilogue state (in-kernel noise only)
    const bool epi = RNG && a.xs != nullptr;
    const unsigned long long rowmask = (K < 64 ? (1ull << K) : 0ull) - 1ull;
    f32x4 accm[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    unsigned fsel = 0;                                     // six 5-bit record offsets: features 16 g + (lane & 15), g = 0..2, as products rec[ia] * rec[ib]
    if (momon) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int f = 16 * g + (lane & 15);
            int ia = 9, ib = 9;                            // slot 9 = 0: the three unused feature slots
            if (f < 8) { ia = f; ib = 8; }                 // x_f * 1
            else if (f == 8) { ia = 8; ib = 8; }           // 1
            else if (f < 9 + 36) {
                const int tq = f - 9;
                int i2 = 0;
                while (tri(i2 + 1, 0) <= tq) ++i2;
                ia = i2; ib = tq - tri(i2, 0);
            }
            fsel |= (unsigned)(ia | (ib << 5)) << (10 * g);
        }
        if (lane < 4 * XSEL) { const int sl = lane % XSEL; xsel[lane] = (sl == 8) ? 1.f : 0.f; }
        __builtin_amdgcn_wave_barrier();
    }
#ifdef VMP_DEBUG_TS
    int fw_it = -1;
#define FW_TS(i) do { if (a.dbg_t && blockIdx.x == 0 && wave == 0 && fw_it == 8 && lane == 0) a.dbg_t[64 + (i)] = clock64(); } while (0)
#else
#define FW_TS(i) do { } while (0)
#endif
    for (; t < ntiles; t += tstride) {
#ifdef VMP_DEBUG_TS
        ++fw_it;
#endif
        FW_TS(0);
        float* et = buf0 + (RNG ? 0 : cur) * (BC * CS);
        const long long row = t * RPT + r;
        const bool on = lane_on && row < a.N;
        const long long rows_here = (a.N - t * RPT) < RPT ? (a.N - t * RPT) : RPT;
        const int tot = (int)rows_here * K * LSn;
        // full = every one of the tile's CT cells is valid: the samples leave through the LDS image as coalesced float4 stores of
        // the tile's CT * L * S contiguous floats.  (Round 4: also for K that does not divide 64 - CT = (64 / K) K < 64 cells - whose
        // lanes used to store their own rows straight from registers: 16 bytes per lane at a 4 L S byte stride, four partial
        // writes per 64-byte segment.)
        const bool full = a.vec_ok && (L & 3) == 0 && rows_here == RPT;
        const int nf4 = CT * Q;                             // float4s of a full tile
        // PST: wave-uniform base of the tile's samples, number of valid cells
        const long long tu = ((long long)__builtin_amdgcn_readfirstlane((int)(t >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        float* __restrict__ xtile = a.x + tu * CT * LSn;
        const int ncell_t = (int)rows_here * K;

        // ---- cell factorisation
        float Lm[TRI], av[L];
#pragma unroll
        for (int i = 0; i < TRI; ++i) { const float pvv = pk_lds[kc * PSTR + i]; Lm[i] = lane_on ? pvv : 0.f; }
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const float e1 = on ? e1r[i] : 0.f;
            const float e2 = on ? e2r[i] : -0.5f;
            Lm[tri(i, i)] = fmaf(-2.f, e2, Lm[tri(i, i)]);
            if constexpr (PST_) { const float hv = hk_lds[kc * HKS + i]; av[i] = e1 + (lane_on ? hv : 0.f); }
            else av[i] = e1 + hkk[i];
        }
        if constexpr (PST_) { const float bv = hk_lds[kc * HKS + L]; biask = lane_on ? bv : 0.f; }
        float ld;
        cell_cholesky<L>(Lm, ld);
        solve_lower<L>(Lm, av);
        float aa = 0.f;
#pragma unroll
        for (int i = 0; i < L; ++i) aa = fmaf(av[i], av[i], aa);
        const float c = on ? (biask + 0.5f * aa - ld) : -INFINITY;
        float mx, se, ex;
        if (k16) {
            mx = row16_max(c);
            ex = on ? __expf(c - mx) : 0.f;
            se = row16_sum(ex);
        } else {
            // the 64-float scratch of the K != 16 row reductions lies in LDS that is idle during the softmax: with two buffers the
            // OTHER one (between the previous tile's copy-out and the DMA issued after the softmax), with one buffer or the pair
            // staging area (in-kernel noise) that area itself (between the previous tile's last store and this tile's first sample)
            float* scr = buf0 + (RNG ? 0 : (cur ^ 1)) * (BC * CS);
            mx = row_max(c, scr, lane, rbase, K);
            ex = on ? __expf(c - mx) : 0.f;
            se = row_sum(ex, scr, lane, rbase, K);
        }
        const float lz = c - mx - __logf(se);
        SV_USE(lz); SV_TS(19); FW_TS(1);
        // ---- epilogue, part 1: the categorical draw of subsample_x (nb_out = 1) and r = exp(log_z).  Same arithmetic as
        // subsample_kernel (inclusive prefix sum of __expf(log_z) over the row's lanes, u from Philox block (n, 0, TAG)): the
        // two paths pick the same component for every row.
        bool sel = false;
        float rv = 0.f;
        if constexpr (RNG) {
            if (epi) {
                const unsigned long long rw = on ? (unsigned long long)row : 0ull;
                unsigned c4[4] = {(unsigned)rw, (unsigned)(rw >> 32), 0u, SUBSAMPLE_TAG};
                philox4x32<VMP_PHILOX_ROUNDS>(c4, (unsigned)rng_seed, (unsigned)(rng_seed >> 32));
                const float uu = (float)(c4[0] >> 8) * 5.9604644775390625e-08f;      // [0, 1)
                rv = on ? __expf(lz) : 0.f;                      // r = exp(log z); also the first term of the row's CDF
                float cum = rv;
                for (int o = 1; o < K; o <<= 1) {
                    const float up = __shfl_up(cum, o);
                    if (k >= o) cum += up;
                }
                const unsigned long long below = __ballot(on && k < K - 1 && cum <= uu);
                const int zk = __popcll((below >> rbase) & rowmask);
                sel = on && k == zk;
                if (on && a.r) a.r[row * K + k] = rv;
            }
        }

        // The factorisation above needed no noise: the previous tile's stores had that long to drain.  Now: next tile's
        // rows (plain loads), then ONE wait that retires this tile's DMA (issued a tile ago), the old stores and those
        // rows - consumed right here so that no compiler-generated wait ever covers the DMA issued next.
        {
            const long long rown = (t + tstride) * RPT + r;
            const long long rcn = (t + tstride < ntiles && lane_on && rown < a.N) ? rown : 0;
#pragma unroll
            for (int i = 0; i < L; ++i) { e1r[i] = a.eta1[rcn * L + i]; e2r[i] = a.eta2d[rcn * L + i]; }
        }
        if constexpr (!RNG) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < L; ++i) asm volatile("" : "+v"(e1r[i]), "+v"(e2r[i]));
        }
        // (in-kernel noise: there is no DMA to retire; the next tile's rows are simply requested here and waited for by the compiler
        //  where the next factorisation uses them - a whole sample loop later.  The explicit vmcnt(0) also waited, in order, for the
        //  PREVIOUS tile's sample stores: 3.3 k of a tile's 22.4 k cycles, profiles/r04_ring_stage_stamps.txt)
        SV_TS(20); FW_TS(2);
        if constexpr (!RNG) { if (t + tstride < ntiles) issue_dma(t + tstride, buf0 + (cur ^ 1) * (BC * CS)); }

        v2f Lm2[TP], av2[LP];
#pragma unroll
        for (int i = 0; i < 2 * TP; ++i) Lm2[i >> 1][i & 1] = (i < TRI) ? Lm[i < TRI ? i : 0] : 0.f;
#pragma unroll
        for (int i = 0; i < 2 * LP; ++i) av2[i >> 1][i & 1] = (i < L) ? av[i < L ? i : 0] : 0.f;

        // ---- samples (two at a time) and the per-cell regulariser term
        v2f eps2 = v2f{0.f, 0.f}, qth = v2f{0.f, 0.f};
        float* cell = et + (lane_on ? lane : 0) * CS;         // lanes past the tile's last cell read (and discard) cell 0: no LDS access outside the wave's buffer
        const unsigned long long cellid = (unsigned long long)(on ? row : 0) * (unsigned long long)K + (unsigned long long)kc;
        // the noise of sample pair `pr` (samples 2 pr, 2 pr + 1) of this lane's cell, generated in registers; pr may differ per lane
        auto gen_pair = [&](unsigned pr, v2f (&eo)[L]) {
            constexpr int L4 = (L + 3) / 4;
#pragma unroll
            for (int j = 0; j < L4; ++j) {
                v2f p4[4];
                philox_normal8<SCT>(cellid, pr * L4 + j, rng_seed, p4, sct);
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2)
                    if (4 * j + t2 < L) eo[4 * j + t2] = p4[t2];
            }
        };
        auto read_pair = [&](int s2, v2f (&eo)[L]) {
            if constexpr (RNG) {
                gen_pair((unsigned)s2 >> 1, eo);
                return;
            }
#pragma unroll
            for (int i = 0; i < L; ++i) {
                if constexpr (ST != 0 && (ST & 1) == 0) eo[i] = *reinterpret_cast<const v2f*>(cell + i * S + s2);   // 8-byte aligned: CS, S, s even
                else eo[i] = v2f{cell[i * S + s2], cell[i * S + s2 + 1]};
            }
        };
        // the two samples x = Ltilde^-T (a + eps) of a pair and their contributions to the cell's sums of eps^2 and of the theta term
        auto compute_pair = [&](const v2f (&ec)[L], bool hv, v2f (&z)[L]) {
#pragma unroll
            for (int i = 0; i < L; ++i) {
                v2f e = ec[i];
                // (in-kernel noise: lanes past the tile's last cell draw the noise of cell 0 - finite, and nothing of theirs is stored)
                if (!RNG && !lane_on) e = v2f{0.f, 0.f};
                if (!hv) e.y = 0.f;
                eps2 = __builtin_elementwise_fma(e, e, eps2);
                z[i] = pk_add_b(e, av2[i >> 1], i & 1);
            }
#pragma unroll
            for (int i = L - 1; i >= 0; --i) {
                v2f tt = z[i];
#pragma unroll
                for (int p2 = i + 1; p2 < L; ++p2) tt = pk_fnma_b(z[p2], Lm2[tri(p2, i) >> 1], tt, tri(p2, i) & 1);
                z[i] = pk_mul_b(tt, Lm2[tri(i, i) >> 1], tri(i, i) & 1);
            }
            v2f d[L];
#pragma unroll
            for (int i = 0; i < L; ++i) d[i] = pk_sub_b(z[i], mk2[i >> 1], i & 1);
            v2f del2 = v2f{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < L; ++i) {
                v2f y = pk_mul_b(d[0], Wt2[tri(i, 0) >> 1], tri(i, 0) & 1);
#pragma unroll
                for (int j = 1; j <= i; ++j) y = pk_fma_b(d[j], Wt2[tri(i, j) >> 1], y, tri(i, j) & 1);
                del2 = __builtin_elementwise_fma(y, y, del2);
            }
            if (!hv) del2.y = 0.f;
            if (student) {
                const float sc = nuk + (float)L;
                qth.x += sc * log1p_f(del2.x * inv_nu);
                qth.y += sc * log1p_f(del2.y * inv_nu);
            } else {
                qth += del2;
            }
        };
        // epilogue, part 2: sample 0 of the drawn component IS x_samples[n] (its lane holds it in registers during pair 0)
        auto emit_sel = [&](const v2f (&z)[L]) {
            if (sel) {
                float* __restrict__ xo = a.xs + row * L;
                if ((L & 3) == 0 && al16_dev(a.xs)) {
#pragma unroll
                    for (int q = 0; q < L / 4; ++q)
                        reinterpret_cast<float4*>(xo)[q] = float4{z[4 * q].x, z[4 * q + 1].x, z[4 * q + 2].x, z[4 * q + 3].x};
                } else {
#pragma unroll
                    for (int i = 0; i < L; ++i) xo[i] = z[i].x;
                }
                if constexpr (PST) {
                    if (momon) {
                        float* xr = xsel + r * XSEL;
#pragma unroll
                        for (int q = 0; q < L / 4; ++q)
                            *reinterpret_cast<f32x4*>(xr + 4 * q) = f32x4{z[4 * q].x, z[4 * q + 1].x, z[4 * q + 2].x, z[4 * q + 3].x};
                    }
                }
            }
        };
        if constexpr (PST2) {
            // Pair staging, TWO pairs per flush, every global store a whole 128-byte line.  A cell's S L floats are P = S / 2 chunks of
            // 64 bytes (one sample pair each), P odd: cells with an EVEN absolute index start on a line, odd ones in mid-line, and the
            // five lines of such a couple are [e0 e1][e2 e3][e4 o0][o1 o2][o3 o4].  So odd cells take their pairs one step ahead
            // (step t: pair (t + 1) mod P): steps (0,1), (2,3), .. then complete one line per cell, written by eight adjacent lanes, and
            // the last step's two 64-byte chunks of a couple are neighbours in memory and in the lanes.  (The one-pair form's
            // 64-byte segment stores did not overlap with the arithmetic: 1.30 ms without them, 1.78 ms with, K = 16, N = 1e6.)
            constexpr int P = ST / 2;
            constexpr int SLOT = WAVE * 16 + 16;             // second slot 64 bytes further: a cell's two chunks in different bank halves
            const unsigned abs0 = (unsigned)tu * (unsigned)CT;                   // parity of the tile's first cell
            const unsigned par = (abs0 + (unsigned)lane) & 1u;
            auto stage = [&](const v2f (&z)[L], int slot) {
                float* wp = buf0 + slot * SLOT + lane * 16;
                const int sw = (lane >> 1) & 3;
#pragma unroll
                for (int j = 0; j < 4; ++j) {               // piece j = [sample j >> 1][coordinates 4 (j & 1) ..]
                    const int i0 = 4 * (j & 1), h2 = j >> 1;
                    *reinterpret_cast<f32x4*>(wp + 4 * (j ^ sw)) = f32x4{z[i0][h2], z[i0 + 1][h2], z[i0 + 2][h2], z[i0 + 3][h2]};
                }
            };
            auto store16 = [&](unsigned off_floats, const f32x4& v, bool ok) {
                if (ok) {
                    if (a.vec_ok) {
                        asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(off_floats * 4u), "v"(v), "s"(xtile) : "memory");
                    } else {
                        float* dst = xtile + off_floats;
                        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
                    }
                }
            };
#pragma unroll 1
            for (int t2 = 0; t2 < P; t2 += 2) {
                const bool dbl = t2 + 1 < P;
                {
                    v2f z[L], ec[L];
                    unsigned q = (unsigned)t2 + par;
                    if (q >= (unsigned)P) q -= P;
                    gen_pair(q, ec);
                    compute_pair(ec, true, z);
                    if constexpr (RNG) { if (epi && q == 0u) emit_sel(z); }
                    stage(z, 0);
                }
                if (dbl) {
                    v2f z[L], ec[L];
                    unsigned q = (unsigned)t2 + 1u + par;
                    if (q >= (unsigned)P) q -= P;
                    gen_pair(q, ec);
                    compute_pair(ec, true, z);
                    if constexpr (RNG) { if (epi && q == 0u) emit_sel(z); }
                    stage(z, 1);
                }
                __builtin_amdgcn_wave_barrier();
                if (dbl) {
                    // lane -> (cell 8 it + (lane >> 3), slot (lane >> 2) & 1, piece lane & 3); swizzle ((cell >> 1) & 3) = (lane >> 4) & 3
                    const int sl = (lane >> 2) & 1, pc = lane & 3;
                    const float* rp = buf0 + sl * SLOT + (lane >> 3) * 16 + 4 * (pc ^ ((lane >> 4) & 3));
                    f32x4 ov[8];
#pragma unroll
                    for (int it = 0; it < 8; ++it) ov[it] = *reinterpret_cast<const f32x4*>(rp + it * 128);
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const unsigned cc = 8u * it + ((unsigned)lane >> 3);
                        const unsigned parc = (abs0 + cc) & 1u;
                        store16(cc * (unsigned)LSn + 16u * ((unsigned)t2 + parc + (unsigned)sl) + 4u * pc, ov[it], (int)cc < ncell_t);
                    }
                } else {
                    const int pc = lane & 3;
                    const float* rp = buf0 + (lane >> 2) * 16 + 4 * (pc ^ ((lane >> 3) & 3));
                    f32x4 ov[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) ov[it] = *reinterpret_cast<const f32x4*>(rp + it * 256);
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const unsigned cc = 16u * it + ((unsigned)lane >> 2);
                        const unsigned parc = (abs0 + cc) & 1u;
                        store16(cc * (unsigned)LSn + (parc ? 0u : 16u * (P - 1)) + 4u * pc, ov[it], (int)cc < ncell_t);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
        v2f en[L];                                         // noise tensor: next pair's noise, read from LDS while this pair is computed
        if constexpr (!RNG) read_pair(0, en);
#pragma unroll 1
        for (int s = 0; s < S; s += 2) {
            const bool hv = (ST != 0 && (ST & 1) == 0) ? true : (s + 1 < S);
            v2f z[L], ec[L];
            if constexpr (RNG) {
                read_pair(s, ec);                           // generated where it is consumed: nothing to prefetch, no copies
            } else {
#pragma unroll
                for (int i = 0; i < L; ++i) ec[i] = en[i];
                read_pair((s + 2 < S) ? s + 2 : s, en);
            }
            compute_pair(ec, hv, z);
            if constexpr (RNG) { if (epi && s == 0) emit_sel(z); }
            if constexpr (PST) {
                float* wp = buf0 + lane * 16;
                const int sw = (lane >> 1) & 3;
#pragma unroll
                for (int j = 0; j < 4; ++j) {               // piece j = [sample j >> 1][coordinates 4 (j & 1) ..]
                    const int i0 = 4 * (j & 1), h2 = j >> 1;
                    *reinterpret_cast<f32x4*>(wp + 4 * (j ^ sw)) = f32x4{z[i0][h2], z[i0 + 1][h2], z[i0 + 2][h2], z[i0 + 3][h2]};
                }
                __builtin_amdgcn_wave_barrier();
                // lane -> (cell 16 it + (lane >> 2), piece lane & 3); the cell's swizzle ((cell >> 1) & 3) does not depend on `it`
                const float* rp = buf0 + (lane >> 2) * 16 + 4 * ((lane & 3) ^ ((lane >> 3) & 3));
                f32x4 ov[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) ov[it] = *reinterpret_cast<const f32x4*>(rp + it * 256);
                const int pc = lane & 3;
                const bool pok = hv || pc < 2;              // an odd S: the last pair has one sample
                float* __restrict__ gs = xtile + (unsigned)(s * L);      // wave-uniform: scalar base + 32-bit lane offset
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int cc = 16 * it + (lane >> 2);
                    if (cc < ncell_t && pok && !(VMP_PST_NOSTORE)) {
                        const unsigned ob = (unsigned)(cc * LSn + 4 * pc) * 4u;
                        if (a.vec_ok) {
                            asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(ob), "v"(ov[it]), "s"(gs) : "memory");
                        } else {
                            float* dst = gs + (ob >> 2);
                            dst[0] = ov[it][0]; dst[1] = ov[it][1]; dst[2] = ov[it][2]; dst[3] = ov[it][3];
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            } else if (!full) {
                // tiles that are not 64 whole cells (K does not divide 64, or the last rows): every lane stores its own
                // cell's two sample rows - 2 * 4L contiguous bytes - straight from registers.  (Staging them in LDS and
                // copying out element by element cost ~20 index instructions per float: 3.9 us of a 12 us launch at
                // N = 64, K = 10, and 40 % on top of the arithmetic of every K = 10 tile at any N.)
                if (on) {
                    float* __restrict__ xo = a.x + cellid * (unsigned long long)LSn + (unsigned)(s * L);
                    if ((L & 3) == 0 && a.vec_ok) {
#pragma unroll
                        for (int q = 0; q < L / 4; ++q)
                            reinterpret_cast<float4*>(xo)[q] = float4{z[4 * q].x, z[4 * q + 1].x, z[4 * q + 2].x, z[4 * q + 3].x};
                        if (hv) {
#pragma unroll
                            for (int q = 0; q < L / 4; ++q)
                                reinterpret_cast<float4*>(xo + L)[q] = float4{z[4 * q].y, z[4 * q + 1].y, z[4 * q + 2].y, z[4 * q + 3].y};
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < L; ++i) { xo[i] = z[i].x; if (hv) xo[L + i] = z[i].y; }
                    }
                }
            } else if (OIMG) {
                if (lane_on) {
                    float* wp = cell + s * L;                 // 64 bytes: [sample s: l 0-3 | 4-7][sample s + 1: l 0-3 | 4-7]
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i0 = 4 * (j & 1), h2 = j >> 1;
                        *reinterpret_cast<f32x4*>(wp + 4 * j) = f32x4{z[i0][h2], z[i0 + 1][h2], z[i0 + 2][h2], z[i0 + 3][h2]};
                    }
                }
            } else if (lane_on) {
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    if constexpr (ST != 0 && (ST & 1) == 0) {
                        *reinterpret_cast<v2f*>(cell + i * S + s) = z[i];
                    } else {
                        cell[i * S + s] = z[i].x;
                        if (hv) cell[i * S + s + 1] = z[i].y;
                    }
                }
            }
        }
        }   // !PST2
        if constexpr (PST) {
            if (momon) {
                // epilogue, part 3 (K = 16): the tile's contribution to sum_n r_nk phi(x_n), phi = [x | 1 | x x^T lower], as three
                // v_mfma_f32_16x16x4_f32: A = r (lane = (row, component) is the operand layout: M = component, k-slot = row of the
                // tile), B = this lane's feature of its row's drawn sample, built from the row's LDS record
                __builtin_amdgcn_wave_barrier();
                const float* xr = xsel + (lane >> 4) * XSEL;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float fa = xr[(fsel >> (10 * g)) & 31u], fb = xr[(fsel >> (10 * g + 5)) & 31u];
                    accm[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(rv, fa * fb, accm[g], 0, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if constexpr (PST_) { const float kv = hk_lds[kc * HKS + L + 1]; kappak = lane_on ? kv : 0.f; }
        if (on) {
            a.lz[row * K + k] = lz;
            a.Tp[row * K + k] = -0.5f * L * LOG_2PI + ld - 0.5f * invS * (eps2.x + eps2.y) + 0.5f * invS * (qth.x + qth.y) - kappak;
        }
        __builtin_amdgcn_wave_barrier();
        SV_TS(21); FW_TS(3);

        // ---- samples out: (cell, S, L) layout, coalesced
        {
            float* __restrict__ g = a.x + t * CT * LSn;
            if (!PST && full) {
                float4* __restrict__ g4 = reinterpret_cast<float4*>(g) + lane;
                constexpr int L4 = (L / 4 > 0) ? L / 4 : 1;
                // batches of CB float4s: all their LDS reads in flight before the first store (one wave per SIMD: nobody
                // else hides the LDS latency)
                constexpr int CB = 5;
                if constexpr (ST != 0) {
#pragma unroll
                    for (int it0 = 0; it0 < Qc; it0 += CB) {
                        float4 v[CB];
#pragma unroll
                        for (int u = 0; u < CB; ++u) {
                            if (it0 + u < Qc) {
                                const float* src = et + ((it0 + u) * WAVE + lane < nf4 ? co_off[it0 + u] : 0);
                                if constexpr (OIMG) {
                                    const f32x4 q4 = *reinterpret_cast<const f32x4*>(src);
                                    v[u] = float4{q4[0], q4[1], q4[2], q4[3]};
                                } else {
                                    v[u].x = src[0]; v[u].y = src[S]; v[u].z = src[2 * S]; v[u].w = src[3 * S];
                                }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < CB; ++u)
                            if (it0 + u < Qc && (it0 + u) * WAVE + lane < nf4) st_x4<!RNG && (VMP_T2_NT & 1) != 0>(g4 + (it0 + u) * WAVE, v[u]);
                    }
                } else {
                    int c2 = o_first, rem = orem_first;
                    for (int it0 = 0; it0 < Q; it0 += CB) {
                        float4 v[CB];
#pragma unroll
                        for (int u = 0; u < CB; ++u) {
                            if (it0 + u < Q) {
                                const int s = rem / L4, l4 = rem - s * L4;
                                const float* src = et + (c2 < CT ? c2 : 0) * CS + (4 * l4) * S + s;
                                v[u].x = src[0]; v[u].y = src[S]; v[u].z = src[2 * S]; v[u].w = src[3 * S];
                                c2 += dco; rem += dro;
                                if (rem >= Q) { rem -= Q; c2 += 1; }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < CB; ++u)
                            if (it0 + u < Q && (it0 + u) * WAVE + lane < nf4) st_x4<!RNG && (VMP_T2_NT & 1) != 0>(g4 + (it0 + u) * WAVE, v[u]);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        SV_TS(22); FW_TS(4);
        cur ^= 1;
    }
    if constexpr (PST) {
        if (momon) {
            // the waves' fp32 accumulators (a few hundred rows each) -> one fp64 partial per block, waves in a fixed order;
            // accumulator register v of lane l = (component 4 (l >> 4) + v, feature 16 g + (l & 15))
            __syncthreads();
            float* red = smem + tab;                        // [nw][3 * 4 * 64]: the staging areas are idle now
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int v = 0; v < 4; ++v) red[wave * 768 + (g * 4 + v) * WAVE + lane] = accm[g][v];
            __syncthreads();
            for (int e = threadIdx.x; e < 768; e += blockDim.x) {
                double sacc = 0.0;
                for (int w = 0; w < nw; ++w) sacc += (double)red[w * 768 + e];
                const int gv = e >> 6, ln = e & 63, g = gv >> 2, v = gv & 3;
                a.mom[((size_t)blockIdx.x * 16 + 4 * (ln >> 4) + v) * MOMF + 16 * g + (ln & 15)] = sacc;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Minibatch form of the in-kernel-noise forward (round 6): lane = (cell, sample PAIR).  At the reference's operating point
// (experiments.py:26: minibatches of 64-100 rows) a wave of the streaming kernels owns ONE tile and walks its S / 2 sample pairs one
// after the other - a chain of five generator + solve + theta-term bodies behind the factorisation, 15.6 us for N = 64.  Here a
// BLOCK owns the tile and wave p of it takes pair p: every wave repeats the cell factorisation (it is the shorter part) and does one
// pair; the per-cell sums over samples (|eps|^2, the theta term) meet in LDS, wave 0 adds them in pair order and writes log z, T'
// and the epilogue outputs (its pair holds sample 0, the one subsample_x keeps).  Same stream, same per-sample arithmetic as the
// streaming forms (x bit-identical); T' differs from theirs in the last bits (partial sums per pair instead of one running sum).
// ---------------------------------------------------------------------------------------------------------
#ifndef VMP_FWD1
#define VMP_FWD1 1                  // 0: A/B builds without the minibatch form
#endif
constexpr int FWD1_MAX_PAIRS = 8;        // block = S / 2 waves (S <= 16; 256 registers per lane)
constexpr int FWD1_MAX_TILES = 256;      // beyond that the streaming forms take over (one block per CU is then no longer latency-bound)
template <int L>
__global__ __launch_bounds__(FWD1_MAX_PAIRS * WAVE) void svae_estep_fwd1_kernel(EFwdArgs a) {
    constexpr int TRI = SvGeo<L>::TRI;
    constexpr int TP = (TRI + 1) / 2, LP = (L + 1) / 2, L4 = (L + 3) / 4;
    __shared__ float scr_all[FWD1_MAX_PAIRS][WAVE];
    __shared__ v2f red[FWD1_MAX_PAIRS][2][WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int K = a.K, S = a.S, LSn = L * S;
    const int RPT = WAVE / K, CT = RPT * K;
    const bool lane_on = lane < CT;
    const int r = lane / K, k = lane - r * K, rbase = lane_on ? r * K : 0, kc = lane_on ? k : 0;
    const long long t = blockIdx.x, row = t * RPT + r;
    const bool on = lane_on && row < a.N;
    const long long rowc = on ? row : 0;
    const unsigned long long rng_seed = a.seed_dev ? *a.seed_dev : a.seed;
    const bool student = a.nu != nullptr;
    float* scr = scr_all[wave];
    // ---- cell factorisation (every wave: all inputs requested at once)
    float Lm[TRI], av[L];
    v2f mk2[LP], Wt2[TP];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const float e1 = a.eta1[rowc * L + i], e2 = a.eta2d[rowc * L + i], hv = a.hk[kc * L + i], mv = a.mk[kc * L + i];
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const float pv = a.Pk[(kc * L + i) * L + j], wv = a.Wk[(kc * L + i) * L + j];
            Lm[tri(i, j)] = lane_on ? pv : 0.f;
            Wt2[tri(i, j) >> 1][tri(i, j) & 1] = lane_on ? wv : 0.f;
        }
        Lm[tri(i, i)] = fmaf(-2.f, on ? e2 : -0.5f, Lm[tri(i, i)]);
        av[i] = (on ? e1 : 0.f) + (lane_on ? hv : 0.f);
        mk2[i >> 1][i & 1] = lane_on ? mv : 0.f;
    }
    if (L & 1) mk2[LP - 1][1] = 0.f;
    if (TRI & 1) Wt2[TP - 1][1] = 0.f;
    const float bv = a.bias[kc], kv = a.kappa[kc], nv = *(student ? a.nu + kc : a.bias);
    const float biask = lane_on ? bv : 0.f, kappak = lane_on ? kv : 0.f, nuk = (student && lane_on) ? nv : 1.f;
    const float inv_nu = 1.0f / nuk;
    float ld;
    cell_cholesky<L>(Lm, ld);
    solve_lower<L>(Lm, av);
    float aa = 0.f;
#pragma unroll
    for (int i = 0; i < L; ++i) aa = fmaf(av[i], av[i], aa);
    const float c = on ? (biask + 0.5f * aa - ld) : -INFINITY;
    const float mx = row_max(c, scr, lane, rbase, K);
    const float ex = on ? __expf(c - mx) : 0.f;
    const float se = row_sum(ex, scr, lane, rbase, K);
    const float lz = c - mx - __logf(se);
    // ---- this wave's sample pair
    v2f Lm2[TP], av2[LP];
#pragma unroll
    for (int i = 0; i < 2 * TP; ++i) Lm2[i >> 1][i & 1] = (i < TRI) ? Lm[i < TRI ? i : 0] : 0.f;
#pragma unroll
    for (int i = 0; i < 2 * LP; ++i) av2[i >> 1][i & 1] = (i < L) ? av[i < L ? i : 0] : 0.f;
    const unsigned long long cellid = (unsigned long long)rowc * (unsigned long long)K + (unsigned long long)kc;
    const int s0 = 2 * wave;
    const bool hv = s0 + 1 < S;
    v2f eps2 = v2f{0.f, 0.f}, qth = v2f{0.f, 0.f}, z[L];
    {
        v2f ec[L];
#pragma unroll
        for (int j = 0; j < L4; ++j) {
            v2f p4[4];
            philox_normal8<false>((unsigned long long)cellid, (unsigned)(wave * L4 + j), rng_seed, p4);
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2)
                if (4 * j + t2 < L) ec[4 * j + t2] = p4[t2];
        }
#pragma unroll
        for (int i = 0; i < L; ++i) {
            v2f e = ec[i];
            if (!hv) e.y = 0.f;
            eps2 = __builtin_elementwise_fma(e, e, eps2);
            z[i] = pk_add_b(e, av2[i >> 1], i & 1);
        }
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
            v2f tt = z[i];
#pragma unroll
            for (int p2 = i + 1; p2 < L; ++p2) tt = pk_fnma_b(z[p2], Lm2[tri(p2, i) >> 1], tt, tri(p2, i) & 1);
            z[i] = pk_mul_b(tt, Lm2[tri(i, i) >> 1], tri(i, i) & 1);
        }
        v2f d[L];
#pragma unroll
        for (int i = 0; i < L; ++i) d[i] = pk_sub_b(z[i], mk2[i >> 1], i & 1);
        v2f del2 = v2f{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < L; ++i) {
            v2f y = pk_mul_b(d[0], Wt2[tri(i, 0) >> 1], tri(i, 0) & 1);
#pragma unroll
            for (int j = 1; j <= i; ++j) y = pk_fma_b(d[j], Wt2[tri(i, j) >> 1], y, tri(i, j) & 1);
            del2 = __builtin_elementwise_fma(y, y, del2);
        }
        if (!hv) del2.y = 0.f;
        if (student) {
            const float sc = nuk + (float)L;
            qth.x = sc * log1p_f(del2.x * inv_nu);
            qth.y = sc * log1p_f(del2.y * inv_nu);
        } else {
            qth = del2;
        }
    }
    if (on) {
        float* __restrict__ xo = a.x + cellid * (unsigned long long)LSn + (unsigned)(s0 * L);
        if ((L & 3) == 0 && a.vec_ok) {
#pragma unroll
            for (int q = 0; q < L / 4; ++q) reinterpret_cast<float4*>(xo)[q] = float4{z[4 * q].x, z[4 * q + 1].x, z[4 * q + 2].x, z[4 * q + 3].x};
            if (hv) {
#pragma unroll
                for (int q = 0; q < L / 4; ++q) reinterpret_cast<float4*>(xo + L)[q] = float4{z[4 * q].y, z[4 * q + 1].y, z[4 * q + 2].y, z[4 * q + 3].y};
            }
        } else {
#pragma unroll
            for (int i = 0; i < L; ++i) { xo[i] = z[i].x; if (hv) xo[L + i] = z[i].y; }
        }
    }
    red[wave][0][lane] = eps2;
    red[wave][1][lane] = qth;
    __syncthreads();
    if (wave != 0) return;
    // ---- wave 0: sums over the pairs in pair order, log z, T', epilogue
    v2f e2s = red[0][0][lane], qts = red[0][1][lane];
    for (int w = 1; w < nw; ++w) { e2s += red[w][0][lane]; qts += red[w][1][lane]; }
    const float invS = 1.0f / (float)S;
    if (on) {
        a.lz[row * K + k] = lz;
        a.Tp[row * K + k] = -0.5f * L * LOG_2PI + ld - 0.5f * invS * (e2s.x + e2s.y) + 0.5f * invS * (qts.x + qts.y) - kappak;
    }
    if (a.xs) {
        // the categorical draw of subsample_x (one draw per row): same arithmetic as subsample_kernel / the streaming forms' epilogue
        const unsigned long long rowmask = (K < 64 ? (1ull << K) : 0ull) - 1ull;
        unsigned c4[4] = {(unsigned)rowc, (unsigned)((unsigned long long)rowc >> 32), 0u, SUBSAMPLE_TAG};
        philox4x32<VMP_PHILOX_ROUNDS>(c4, (unsigned)rng_seed, (unsigned)(rng_seed >> 32));
        const float uu = (float)(c4[0] >> 8) * 5.9604644775390625e-08f;
        const float rv = on ? __expf(lz) : 0.f;
        float cum = rv;
        for (int o = 1; o < K; o <<= 1) {
            const float up = __shfl_up(cum, o);
            if (k >= o) cum += up;
        }
        const unsigned long long below = __ballot(on && k < K - 1 && cum <= uu);
        const int zk = __popcll((below >> rbase) & rowmask);
        if (on && a.r) a.r[row * K + k] = rv;
        if (on && k == zk) {
            float* __restrict__ xo = a.xs + row * L;
#pragma unroll
            for (int i = 0; i < L; ++i) xo[i] = z[i].x;
        }
    }
}

// Large-S form of the forward kernel (evaluation runs use S=100, experiments.py:283): the cell's L*S noise block no
// longer fits the per-wave LDS tile, so the samples are processed SC at a time.  Same lane mapping and arithmetic
// order per sample as svae_estep_fwd_kernel; eps^2 / q_theta accumulate across chunks in sample order.
template <int L>
__global__ __launch_bounds__(SV_FWD_MAX_NW * WAVE) void svae_estep_fwd_chunked_kernel(EFwdArgs a, int SC) {
    constexpr int TRI = SvGeo<L>::TRI;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int K = a.K, S = a.S;
    const int LSn = L * S, CSTR = (L * SC) | 1;
    const int RPT = WAVE / K, CT = RPT * K;
    float* et = smem + wave * (WAVE * CSTR + WAVE);
    float* scr = et + WAVE * CSTR;
    const bool lane_on = lane < CT;
    const int r = lane / K, k = lane - r * K, rbase = lane_on ? r * K : 0;
    float Pl[TRI], hkk[L], mkk[L], Wt[TRI];
    const bool student = a.nu != nullptr;
    const int kc = lane_on ? k : 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const float hv = a.hk[kc * L + i], mv = a.mk[kc * L + i];
        hkk[i] = lane_on ? hv : 0.f;
        mkk[i] = lane_on ? mv : 0.f;
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const float pvv = a.Pk[(kc * L + i) * L + j], wv = a.Wk[(kc * L + i) * L + j];
            Pl[tri(i, j)] = lane_on ? pvv : (i == j ? 1.f : 0.f);
            Wt[tri(i, j)] = lane_on ? wv : 0.f;
        }
    }
    const float bv = a.bias[kc], kv = a.kappa[kc], nv = *(student ? a.nu + kc : a.bias);
    const float biask = lane_on ? bv : 0.f, kappak = lane_on ? kv : 0.f, nuk = (student && lane_on) ? nv : 1.f;
    const float inv_nu = 1.0f / nuk, invS = 1.0f / (float)S;

    const long long ntiles = (a.N + RPT - 1) / RPT;
    for (long long t = (long long)blockIdx.x * nw + wave; t < ntiles; t += (long long)gridDim.x * nw) {
        const long long row = t * RPT + r;
        const bool on = lane_on && row < a.N;
        const long long rows_here = (a.N - t * RPT) < RPT ? (a.N - t * RPT) : RPT;
        const int cells = (int)rows_here * K;
        float Lm[TRI], av[L];
#pragma unroll
        for (int i = 0; i < TRI; ++i) Lm[i] = Pl[i];
        const long long rowc = on ? row : 0;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const float e1v = a.eta1[rowc * L + i], e2v = a.eta2d[rowc * L + i];
            const float e1 = on ? e1v : 0.f;
            const float e2 = on ? e2v : -0.5f;
            Lm[tri(i, i)] = fmaf(-2.f, e2, Lm[tri(i, i)]);
            av[i] = e1 + hkk[i];
        }
        float ld;
        cell_cholesky<L>(Lm, ld);
        solve_lower<L>(Lm, av);
        float aa = 0.f;
#pragma unroll
        for (int i = 0; i < L; ++i) aa = fmaf(av[i], av[i], aa);
        const float c = on ? (biask + 0.5f * aa - ld) : -INFINITY;
        const float mx = row_max(c, scr, lane, rbase, K);
        const float ex = on ? __expf(c - mx) : 0.f;
        const float se = row_sum(ex, scr, lane, rbase, K);
        const float lz = c - mx - __logf(se);

        float eps2 = 0.f, qth = 0.f;
        float* cell = et + lane * CSTR;
        const float* __restrict__ gin = a.noise + t * CT * LSn;
        float* __restrict__ gout = a.x + t * CT * LSn;
        for (int c0 = 0; c0 < S; c0 += SC) {
            const int sc = (S - c0) < SC ? (S - c0) : SC;
            const int per = L * sc, tot = cells * per;
            for (int e = lane; e < tot; e += WAVE) {       // noise (cell, L, S) -> LDS (cell, L, sc)
                const int c2 = e / per, rem = e - c2 * per;
                const int i = rem / sc, s = rem - i * sc;
                et[c2 * CSTR + i * sc + s] = gin[c2 * LSn + i * S + c0 + s];
            }
            __builtin_amdgcn_wave_barrier();
            for (int s = 0; s < sc; ++s) {
                float z[L];
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    const float e = lane_on ? cell[i * sc + s] : 0.f;
                    eps2 = fmaf(e, e, eps2);
                    z[i] = av[i] + e;
                }
                solve_lower_t<L>(Lm, z);
                float d[L];
#pragma unroll
                for (int i = 0; i < L; ++i) d[i] = z[i] - mkk[i];
                float del2 = 0.f;
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    float y = 0.f;
#pragma unroll
                    for (int j = 0; j <= i; ++j) y = fmaf(Wt[tri(i, j)], d[j], y);
                    del2 = fmaf(y, y, del2);
                }
                qth += student ? (nuk + (float)L) * log1p_f(del2 * inv_nu) : del2;
                if (lane_on) {
#pragma unroll
                    for (int i = 0; i < L; ++i) cell[i * sc + s] = z[i];
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int o = lane; o < tot; o += WAVE) {       // LDS (cell, L, sc) -> x (cell, S, L)
                const int c2 = o / per, rem = o - c2 * per;
                const int s = rem / L, l = rem - s * L;
                gout[c2 * LSn + (c0 + s) * L + l] = et[c2 * CSTR + l * sc + s];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (on) {
            a.lz[row * K + k] = lz;
            a.Tp[row * K + k] = -0.5f * L * LOG_2PI + ld - 0.5f * invS * eps2 + 0.5f * invS * qth - kappak;
        }
    }
}


struct SubArgs {
    const float* x;         // (N,K,S,L)
    const float* lz;        // (N,K)
    const float* u;         // (N,S_out) uniforms in [0,1) or NULL
    const long long* z;     // (N,S_out) indices or NULL
    float* out;             // (N,S_out,L)
    long long* z_out;       // (N,S_out) chosen component (may be NULL)
    long long N;
    int K, S, L, S_out;
    int rng;                // u == z == NULL: uniforms from Philox4x32-7, key = seed, counter = (n, s, tag)
    unsigned long long seed;
    const unsigned long long* seed_dev;   // non-NULL: key read from this device word
};

// One (n,k) cell per lane, RPT = 64 / K whole rows per wave: log_z is read coalesced (a thread-per-row loop over k touched 64
// cache lines per load instruction - the access pattern that bound the round-2 backward kernel), the inverse CDF is an
// inclusive prefix sum over the K lanes of the row (shuffle-up steps, guarded so that they never cross a row) followed by a
// count of the lanes whose cumulative probability does not exceed u; the chosen sample row leaves through the row's first
// lanes, 16 bytes each where the layout allows it.
__global__ __launch_bounds__(256) void subsample_kernel(SubArgs a) {
    constexpr int UN = 4;                                    // tiles per wave and turn: their three dependent loads (log_z -> u -> sample row) overlap
    const int lane = threadIdx.x & 63;
    const int K = a.K, RPT = WAVE / K, CT = RPT * K;
    const bool lane_on = lane < CT;
    const int r = lane / K, k = lane - r * K, rbase = r * K;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long ntiles = (a.N + RPT - 1) / RPT;
    const bool vec = (a.L & 3) == 0 && al16_dev(a.x) && al16_dev(a.out);
    const unsigned long long rowmask = (K < 64 ? (1ull << K) : 0ull) - 1ull;
    const unsigned long long sub_seed = a.seed_dev ? *a.seed_dev : a.seed;
    for (long long t0 = wave * UN; t0 < ntiles; t0 += nwaves * UN) {
        long long n[UN];
        bool on[UN];
        float cum[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            n[q] = (t0 + q) * RPT + r;
            on[q] = lane_on && n[q] < a.N;
            cum[q] = on[q] ? a.lz[n[q] * K + k] : -INFINITY;
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) cum[q] = __expf(cum[q]);
        if (!a.z) {
            for (int o = 1; o < K; o <<= 1) {
#pragma unroll
                for (int q = 0; q < UN; ++q) {
                    const float up = __shfl_up(cum[q], o);
                    if (k >= o) cum[q] += up;
                }
            }
        }
        for (int s = 0; s < a.S_out; ++s) {
            int zk[UN];
            if (a.z) {
#pragma unroll
                for (int q = 0; q < UN; ++q) zk[q] = on[q] ? (int)a.z[n[q] * a.S_out + s] : 0;
            } else {
                float uu[UN];
#pragma unroll
                for (int q = 0; q < UN; ++q) {
                    if (a.rng) {
                        // one Philox block per (row, draw); the tag word keeps this stream apart from the E-step's normals
                        unsigned c[4] = {(unsigned)n[q], (unsigned)((unsigned long long)n[q] >> 32), (unsigned)s, SUBSAMPLE_TAG};
                        philox4x32<VMP_PHILOX_ROUNDS>(c, (unsigned)sub_seed, (unsigned)(sub_seed >> 32));
                        uu[q] = (float)(c[0] >> 8) * 5.9604644775390625e-08f;      // [0, 1)
                    } else {
                        uu[q] = on[q] ? a.u[n[q] * a.S_out + s] : 0.f;
                    }
                }
#pragma unroll
                for (int q = 0; q < UN; ++q) {
                    const unsigned long long below = __ballot(on[q] && k < K - 1 && cum[q] <= uu[q]);   // lanes k with cdf_k <= u
                    zk[q] = __popcll((below >> rbase) & rowmask);                                     // first k with u < cdf_k (K-1 if none)
                }
            }
            if (vec) {
                float4 v[UN];
#pragma unroll
                for (int q = 0; q < UN; ++q) {
                    const bool act = on[q] && k < (a.L >> 2);
                    const float* __restrict__ src = a.x + (((act ? n[q] : 0) * K + (act ? zk[q] : 0)) * a.S + s) * a.L;
                    v[q] = reinterpret_cast<const float4*>(src)[act ? k : 0];
                }
#pragma unroll
                for (int q = 0; q < UN; ++q)
                    if (on[q] && k < (a.L >> 2)) reinterpret_cast<float4*>(a.out + (n[q] * a.S_out + s) * a.L)[k] = v[q];
            } else {
#pragma unroll
                for (int q = 0; q < UN; ++q)
                    if (on[q]) {
                        const float* __restrict__ src = a.x + ((n[q] * K + zk[q]) * a.S + s) * a.L;
                        float* __restrict__ dst = a.out + (n[q] * a.S_out + s) * a.L;
                        for (int l = k; l < a.L; l += K) dst[l] = src[l];
                    }
            }
#pragma unroll
            for (int q = 0; q < UN; ++q)
                if (on[q] && k == 0 && a.z_out) a.z_out[n[q] * a.S_out + s] = zk[q];
        }
    }
}

int check_sv(long long N, int K, int L, int S) {
    if (N <= 0 || S <= 0) { set_error("N and S must be positive"); return VMP_E_BADARG; }
    if (L < 1 || L > VMP_MAX_D) { set_error("L=%d outside compiled range 1..%d", L, VMP_MAX_D); return VMP_E_DIM; }
    if (K < 1 || K > VMP_MAX_K) { set_error("K=%d outside compiled range 1..%d", K, VMP_MAX_K); return VMP_E_DIM; }
    if (S > (1 << 16)) { set_error("S=%d too large", S); return VMP_E_DIM; }
    return 0;
}

#ifdef VMP_DEBUG_TS
static long long* g_dbg_svae = nullptr;
#endif
// Small latent dimensions (C1 / C2: L = 2) leave most of a SIMD's registers and of the CU's LDS idle at one block per CU, and their
// tiles are short dependent chains: several blocks per CU (round 6; the big-L geometries stay as tuned).  Per-lane registers by L
// (tools/kreg.py): forward 77 / 95 (L = 2 / 3), generic backward 71 / 95.
inline int fwd4_blocks_per_cu(int L, size_t lds_block) {
    const int by_regs = L <= 2 ? 3 : L == 3 ? 2 : 1;                 // 8-wave blocks: 6 / 4 / 2 waves per SIMD
    const int by_lds = lds_block ? (int)(lds_budget() / lds_block) : 1;
    const int b = by_regs < by_lds ? by_regs : by_lds;
    return b < 1 ? 1 : b;
}
inline int sv_blocks_l(long long N, int K, int L) {
    const int RPT = WAVE / K;
    const long long ntiles = (N + RPT - 1) / RPT;
    long long b = (ntiles + SV_NW - 1) / SV_NW;
    const long long cap = L <= 2 ? 2048 : L == 3 ? 1280 : 512;     // 4-wave blocks: 8 / 5 / 2 per CU
    if (b > cap) b = cap;
    if (b > SV_MAX_BLOCKS) b = SV_MAX_BLOCKS;
    return (int)b;
}

int sv_blocks(long long N, int K) {
    const int RPT = WAVE / K;
    const long long ntiles = (N + RPT - 1) / RPT;
    long long b = (ntiles + SV_NW - 1) / SV_NW;
    // 512 = two 4-wave blocks per CU, all resident at once; with 1024 the second round of blocks starts unevenly
    // (measured at C3: backward 3.65 -> 3.44 ms)
    if (b > 512) b = 512;
    if (b > SV_MAX_BLOCKS) b = SV_MAX_BLOCKS;
    return (int)b;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#define VMP_DISPATCH_L(Lv, CALL)           \
    switch (Lv) {                           \
        case 1: { constexpr int LL = 1; CALL; } break; \
        case 2: { constexpr int LL = 2; CALL; } break; \
        case 3: { constexpr int LL = 3; CALL; } break; \
        case 4: { constexpr int LL = 4; CALL; } break; \
        case 5: { constexpr int LL = 5; CALL; } break; \
        case 6: { constexpr int LL = 6; CALL; } break; \
        case 7: { constexpr int LL = 7; CALL; } break; \
        case 8: { constexpr int LL = 8; CALL; } break; \
        default: break;                     \
    }

#ifndef VMP_BWD1
#define VMP_BWD1 1                  // 0: A/B builds without the minibatch form of the backward kernel
#endif
constexpr int BWD1_MAX_PAIRS = 8, BWD1_MAX_TILES = 256;         // (kernel: vmp_svae_mini.hip)
bool bwd1_applies(int64_t N, int K, int L, int S, bool student) {
    const long long nt = (N + WAVE / K - 1) / (WAVE / K);
    return VMP_BWD1 && !student && nt <= BWD1_MAX_TILES && (S + 1) / 2 <= BWD1_MAX_PAIRS;
}

}  // namespace

extern "C" {

#ifdef VMP_DEBUG_TS
void vmp_debug_set_svae_timestamps(long long* p) { g_dbg_svae = p; }      // exploration builds only (tools/svae_ts.py)
#endif
int vmp_svae_bwd_partial_words(int L) { return 2 * (L + L * (L + 1) / 2 + 1); }

size_t vmp_svae_workspace_bytes(int64_t N, int K, int L) {
    (void)N;
    return (size_t)SV_MAX_BLOCKS * K * vmp_svae_bwd_partial_words(L) * sizeof(float);
}

int vmp_svae_bwd_blocks(int64_t N, int K) { return sv_blocks(N, K); }

// LDS-DMA / in-kernel-noise forward kernel: launch geometry, or 0 waves when the shape is not covered
static int fwd4_plan(int K, int L, int S, int& CS, size_t& lds4, bool rng = false, bool* pair_stage = nullptr) {
    if (pair_stage) *pair_stage = false;
    CS = L * S;
    if (((CS >> 2) & 1) == 0) CS += 4;
    size_t table = (size_t)((K * ((L * (L + 1) / 2) | 1) + 3) & ~3) * sizeof(float);
    const int BC = VMP_FWD_BC_TILE ? (WAVE / K) * K : WAVE;      // cells per tile buffer (the kernel's BC)
    size_t pw = (size_t)((rng ? 1 : 2) * BC * CS) * sizeof(float);
    // the tile-buffer forms: whole 16-byte pieces per cell, and a cell's noise block inside the per-wave LDS tile
    const bool fits = (L * S) % 4 == 0 && (size_t)(L * S | 1) * WAVE * sizeof(float) <= 36 * 1024;
    const size_t budget = lds_budget();
    int nw4 = (fits && budget > table) ? (int)((budget - table) / pw) : 0;
    if (nw4 > (rng ? 8 : 4)) nw4 = rng ? 8 : 4;
    // S = 10 (the compiled-in sample count): the two-pair staging form, eight waves, whole-line stores - where the tile buffer
    // admits fewer than eight waves (K = 16, 7, 8, 9)
    const bool pst2 = VMP_FWD_PAIR_STAGE2 && S == 10 && (nw4 < 8 || VMP_FWD_PST2_ALWAYS);
    if (rng && L == 8 && VMP_FWD_PAIR_STAGE && (nw4 < VMP_FWD_PAIR_STAGE_BELOW || pst2) && pair_stage) {
        // in-kernel noise, and the tile buffer admits few waves or does not fit at all (large S: evaluation runs use S = 100,
        // experiments.py:283): the per-pair staging form (PST), which has no S-sized buffer.  (Round 4 also chose it where the tile
        // buffer admits seven waves - K = 16, 7, 8, 9 at S = 10.  With round 5's cheaper generator the ONE-pair form is bound by its
        // 64-byte segment stores: same box, K = 16 1.78 -> 1.52 ms, K = 8 1.04 -> 0.90, K = 7 0.91 -> 0.84 with the seven-wave tile buffer.)
        *pair_stage = true;
        table += (size_t)K * 12 * sizeof(float);             // the kernel's [K][HKS] table of h_k | bias_k | kappa_k
        if (VMP_FWD_SINCOS_TAB) table += (size_t)SCT_WORDS * sizeof(float);   // and its (cos, sin) table of the generator's directions
        pw = (size_t)(pst2 ? 2 * (WAVE * 16 + 16) : WAVE * 16) * sizeof(float);
        nw4 = budget > table ? (int)((budget - table) / pw) : 0;
        if (nw4 > 8) nw4 = 8;
    }
    if (nw4 < 1) return 0;
    lds4 = table + pw * nw4;
    return nw4;
}

static int run_fwd(EFwdArgs a, int L, void* stream, bool rng) {
#ifdef VMP_DEBUG_TS
    a.dbg_t = g_dbg_svae;
#endif
    const long long N = a.N;
    const int K = a.K, S = a.S;
    const float* noise = a.noise;
    int rc;
    if (rng) {
        int CS = 0;
        size_t lds4 = 0;
        bool ps = false;
        {   // minibatch sizes: one block per tile, one wave per sample pair (svae_estep_fwd1_kernel)
            const long long nt1 = (N + WAVE / K - 1) / (WAVE / K);
            const int P = (S + 1) / 2;
            if (VMP_FWD1 && !a.mom && nt1 <= FWD1_MAX_TILES && P <= FWD1_MAX_PAIRS) {
                rc = -1;
                VMP_DISPATCH_L(L, {
                    hipLaunchKernelGGL((svae_estep_fwd1_kernel<LL>), dim3((int)nt1), dim3(P * WAVE), 0, static_cast<hipStream_t>(stream), a);
                    rc = check_launch("svae_estep_fwd1_kernel");
                });
                return rc;
            }
        }
        const int nw4 = fwd4_plan(K, L, S, CS, lds4, true, &ps);
        if (nw4 < 1) { set_error("in-kernel noise covers L = 8, and L < 8 with L*S %% 4 == 0 tiles that fit the LDS (L=%d, S=%d)", L, S); return VMP_E_DIM; }
        const int RPT4 = WAVE / K;
        long long bl = ((N + RPT4 - 1) / RPT4 + nw4 - 1) / nw4;
        const long long blcap = 256ll * fwd4_blocks_per_cu(L, lds4);
        if (bl > blcap) bl = blcap;
        if (a.mom) {
            if (!(ps && K == 16 && L == 8)) { set_error("in-kernel moments cover K = 16, L = 8 (vmp_svae_fwd_mom_blocks)"); return VMP_E_DIM; }
            lds4 += (size_t)nw4 * 4 * XSEL * sizeof(float);          // the waves' drawn-sample records
            if (lds4 < (size_t)(((K * ((L * (L + 1) / 2) | 1) + 3) & ~3) + K * 12 + (VMP_FWD_SINCOS_TAB ? SCT_WORDS : 0)) * sizeof(float) + (size_t)nw4 * 768 * sizeof(float)) {
                set_error("in-kernel moments: staging area smaller than the block reduction");
                return VMP_E_WS;
            }
        }
        rc = -1;
        VMP_DISPATCH_L(L, {
            if (LL == 8 && ps) {
                if (S == 10) {
                    if ((rc = set_dyn_lds(reinterpret_cast<const void*>(svae_estep_fwd4_kernel<8, 10, true, true>), lds4, "svae_estep_fwd")) != 0) return rc;
                    hipLaunchKernelGGL((svae_estep_fwd4_kernel<8, 10, true, true>), dim3((int)bl), dim3(nw4 * WAVE), lds4, static_cast<hipStream_t>(stream), a, CS);
                } else {
                    if ((rc = set_dyn_lds(reinterpret_cast<const void*>(svae_estep_fwd4_kernel<8, 0, true, true>), lds4, "svae_estep_fwd")) != 0) return rc;
                    hipLaunchKernelGGL((svae_estep_fwd4_kernel<8, 0, true, true>), dim3((int)bl), dim3(nw4 * WAVE), lds4, static_cast<hipStream_t>(stream), a, CS);
                }
            } else if (S == 10) {
                if ((rc = set_dyn_lds(reinterpret_cast<const void*>(svae_estep_fwd4_kernel<LL, 10, true>), lds4, "svae_estep_fwd")) != 0) return rc;
                hipLaunchKernelGGL((svae_estep_fwd4_kernel<LL, 10, true>), dim3((int)bl), dim3(nw4 * WAVE), lds4, static_cast<hipStream_t>(stream), a, CS);
            } else {
                if ((rc = set_dyn_lds(reinterpret_cast<const void*>(svae_estep_fwd4_kernel<LL, 0, true>), lds4, "svae_estep_fwd")) != 0) return rc;
                hipLaunchKernelGGL((svae_estep_fwd4_kernel<LL, 0, true>), dim3((int)bl), dim3(nw4 * WAVE), lds4, static_cast<hipStream_t>(stream), a, CS);
            }
            rc = check_launch("svae_estep_fwd4_kernel<rng>");
        });
        return rc;
    }
    if ((size_t)(L * S | 1) * WAVE * sizeof(float) > 36 * 1024) {
        // the cell's noise block does not fit the LDS tile: process the samples SC at a time
        int SC = (32 * 1024 / (int)(WAVE * sizeof(float))) / L;   // L*SC*64*4 B <= 32 KiB per wave
        const size_t pw = (size_t)(WAVE * ((L * SC) | 1) + WAVE) * sizeof(float);
        const int nwc = 4;
        const int RPTc = WAVE / K;
        long long bl = ((N + RPTc - 1) / RPTc + nwc - 1) / nwc;
        if (bl > 1024) bl = 1024;
        rc = -1;
        VMP_DISPATCH_L(L, {
            if ((rc = set_dyn_lds(reinterpret_cast<const void*>(svae_estep_fwd_chunked_kernel<LL>), (pw * nwc), "svae_estep_fwd")) != 0) return rc;
            hipLaunchKernelGGL((svae_estep_fwd_chunked_kernel<LL>), dim3((int)bl), dim3(nwc * WAVE), pw * nwc, static_cast<hipStream_t>(stream), a, SC);
            rc = check_launch("svae_estep_fwd_chunked_kernel");
        });
        return rc;
    }
    if ((L * S) % 4 == 0 && al16(noise)) {
        // LDS-DMA kernel: cell stride CS = L*S rounded so that CS/4 is odd
        int CS = L * S;
        if (((CS >> 2) & 1) == 0) CS += 4;
        const size_t table = (size_t)((K * ((L * (L + 1) / 2) | 1) + 3) & ~3) * sizeof(float);
        const int BC = VMP_FWD_BC_TILE ? (WAVE / K) * K : WAVE;   // cells per tile buffer (the kernel's BC)
        const size_t pw = (size_t)(2 * BC * CS) * sizeof(float);      // (the row-reduction scratch lies in the idle buffer)
        const size_t budget = lds_budget();
        int nw4 = budget > table ? (int)((budget - table) / pw) : 0;
        if (nw4 > 4) nw4 = 4;
        if (nw4 >= 1) {
            const size_t lds4 = table + pw * nw4;
            const int RPT4 = WAVE / K;
            long long bl = ((N + RPT4 - 1) / RPT4 + nw4 - 1) / nw4;
            const long long blcap = 256ll * fwd4_blocks_per_cu(L, lds4);
            if (bl > blcap) bl = blcap;
            rc = -1;
            VMP_DISPATCH_L(L, {
                if (S == 10) {
                    if ((rc = set_dyn_lds(reinterpret_cast<const void*>(svae_estep_fwd4_kernel<LL, 10, false>), lds4, "svae_estep_fwd")) != 0) return rc;
                    hipLaunchKernelGGL((svae_estep_fwd4_kernel<LL, 10, false>), dim3((int)bl), dim3(nw4 * WAVE), lds4, static_cast<hipStream_t>(stream), a, CS);
                } else {
                    if ((rc = set_dyn_lds(reinterpret_cast<const void*>(svae_estep_fwd4_kernel<LL, 0, false>), lds4, "svae_estep_fwd")) != 0) return rc;
                    hipLaunchKernelGGL((svae_estep_fwd4_kernel<LL, 0, false>), dim3((int)bl), dim3(nw4 * WAVE), lds4, static_cast<hipStream_t>(stream), a, CS);
                }
                rc = check_launch("svae_estep_fwd4_kernel");
            });
            return rc;
        }
    }
    {   // register-staged kernel: L*S not a multiple of 4, or a misaligned noise tensor
        const size_t table = (size_t)K * ((L * (L + 1) / 2) | 1) * sizeof(float);
        const size_t pw = (size_t)(WAVE * (L * S | 1) + WAVE) * sizeof(float);
        const size_t budget3 = lds_budget() - 10 * 1024;
        int nw3 = budget3 > table ? (int)((budget3 - table) / pw) : 0;        // one block per CU, as many waves as 160 KiB of LDS hold
        if (nw3 > SV_FWD_MAX_NW) nw3 = SV_FWD_MAX_NW;
        if (nw3 < 1) nw3 = 1;
        const size_t lds3 = table + pw * nw3;
        const int RPT3 = WAVE / K;
        long long bl = ((N + RPT3 - 1) / RPT3 + nw3 - 1) / nw3;
        if (bl > 256) bl = 256;
        rc = -1;
        VMP_DISPATCH_L(L, {
            if (S == 10) {
                if (lds3 > 64 * 1024)
                    if ((rc = set_dyn_lds(reinterpret_cast<const void*>(svae_estep_fwd3_kernel<LL, 10>), lds3, "svae_estep_fwd")) != 0) return rc;
                hipLaunchKernelGGL((svae_estep_fwd3_kernel<LL, 10>), dim3((int)bl), dim3(nw3 * WAVE), lds3, static_cast<hipStream_t>(stream), a);
            } else {
                if (lds3 > 64 * 1024)
                    if ((rc = set_dyn_lds(reinterpret_cast<const void*>(svae_estep_fwd3_kernel<LL, 0>), lds3, "svae_estep_fwd")) != 0) return rc;
                hipLaunchKernelGGL((svae_estep_fwd3_kernel<LL, 0>), dim3((int)bl), dim3(nw3 * WAVE), lds3, static_cast<hipStream_t>(stream), a);
            }
            rc = check_launch("svae_estep_fwd3_kernel");
        });
        return rc;
    }
}

int vmp_svae_estep_fwd(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                       const float* noise, const float* mk, const float* Wk, const float* kappa, const float* nu,
                       int64_t N, int K, int L, int S, float* x, float* lz, float* Tp, void* stream) {
    int rc = check_sv(N, K, L, S);
    if (rc) return rc;
    if (!eta1 || !eta2d || !hk || !Pk || !bias || !noise || !mk || !Wk || !kappa || !x || !lz || !Tp) {
        set_error("vmp_svae_estep_fwd: null pointer");
        return VMP_E_BADARG;
    }
    EFwdArgs a{eta1, eta2d, hk, Pk, bias, noise, mk, Wk, kappa, nu, x, lz, Tp, N, K, S, 0, 0ull};
    a.vec_ok = al16(noise) && al16(x);
    return run_fwd(a, L, stream, false);
}

int vmp_svae_rng_in_kernel(int K, int L, int S) {
    int CS = 0;
    size_t lds4 = 0;
    bool ps = false;
    return (K >= 1 && K <= 64 && L >= 1 && L <= 8 && fwd4_plan(K, L, S, CS, lds4, true, &ps) >= 1) ? 1 : 0;
}

static int philox_noise_impl(uint64_t seed, const uint64_t* seed_dev, int64_t N, int K, int L, int S, float* noise, void* stream) {
    int rc = check_sv(N, K, L, S);
    if (rc) return rc;
    if (!noise) { set_error("vmp_svae_philox_noise: null pointer"); return VMP_E_BADARG; }
    NoiseArgs na{noise, (long long)N * K, L, S, (unsigned long long)seed, reinterpret_cast<const unsigned long long*>(seed_dev)};
    const long long total = na.cells * ((S + 1) / 2) * ((L + 3) / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(philox_noise_kernel, dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), na);
    return check_launch("philox_noise_kernel");
}

int vmp_svae_philox_noise(uint64_t seed, int64_t N, int K, int L, int S, float* noise, void* stream) {
    return philox_noise_impl(seed, nullptr, N, K, L, S, noise, stream);
}

int vmp_svae_philox_noise_dev(const uint64_t* seed_dev, int64_t N, int K, int L, int S, float* noise, void* stream) {
    if (!seed_dev) { set_error("vmp_svae_philox_noise_dev: null pointer"); return VMP_E_BADARG; }
    return philox_noise_impl(0, seed_dev, N, K, L, S, noise, stream);
}

int vmp_svae_estep_fwd_rng(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                           uint64_t seed, const float* mk, const float* Wk, const float* kappa, const float* nu,
                           int64_t N, int K, int L, int S, float* x, float* lz, float* Tp, float* noise_ws, void* stream) {
    int rc = check_sv(N, K, L, S);
    if (rc) return rc;
    if (!eta1 || !eta2d || !hk || !Pk || !bias || !mk || !Wk || !kappa || !x || !lz || !Tp) {
        set_error("vmp_svae_estep_fwd_rng: null pointer");
        return VMP_E_BADARG;
    }
    EFwdArgs a{eta1, eta2d, hk, Pk, bias, nullptr, mk, Wk, kappa, nu, x, lz, Tp, N, K, S, 0, (unsigned long long)seed};
    if (vmp_svae_rng_in_kernel(K, L, S)) {
        a.vec_ok = al16(x);
        return run_fwd(a, L, stream, true);
    }
    if (!noise_ws) { set_error("vmp_svae_estep_fwd_rng: this shape needs the (N,K,L,S) noise workspace"); return VMP_E_WS; }
    rc = vmp_svae_philox_noise(seed, N, K, L, S, noise_ws, stream);
    if (rc) return rc;
    a.noise = noise_ws;
    a.vec_ok = al16(noise_ws) && al16(x);
    return run_fwd(a, L, stream, false);
}

int vmp_svae_estep_fwd_rng_dev(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                               const uint64_t* seed_dev, const float* mk, const float* Wk, const float* kappa, const float* nu,
                               int64_t N, int K, int L, int S, float* x, float* lz, float* Tp, void* stream) {
    int rc = check_sv(N, K, L, S);
    if (rc) return rc;
    if (!eta1 || !eta2d || !hk || !Pk || !bias || !seed_dev || !mk || !Wk || !kappa || !x || !lz || !Tp) {
        set_error("vmp_svae_estep_fwd_rng_dev: null pointer");
        return VMP_E_BADARG;
    }
    if (!vmp_svae_rng_in_kernel(K, L, S)) {
        set_error("vmp_svae_estep_fwd_rng_dev: K=%d L=%d S=%d is outside the in-kernel generator's shapes (vmp_svae_rng_in_kernel)", K, L, S);
        return VMP_E_DIM;
    }
    EFwdArgs a{eta1, eta2d, hk, Pk, bias, nullptr, mk, Wk, kappa, nu, x, lz, Tp, N, K, S, 0, 0ull,
               reinterpret_cast<const unsigned long long*>(seed_dev)};
    a.vec_ok = al16(x);
    return run_fwd(a, L, stream, true);
}

int vmp_svae_fwd_mom_blocks(int64_t N, int K, int L, int S) {
    int CS = 0;
    size_t lds4 = 0;
    bool ps = false;
    if (N <= 0 || K != 16 || L != 8) return 0;
    if (VMP_FWD1 && (N + 3) / 4 <= FWD1_MAX_TILES && (S + 1) / 2 <= FWD1_MAX_PAIRS) return 0;   // minibatch form (svae_estep_fwd1_kernel): no in-kernel moments
    const int nw4 = fwd4_plan(K, L, S, CS, lds4, true, &ps);
    if (nw4 < 1 || !ps) return 0;
    if (lds4 + (size_t)nw4 * 4 * XSEL * sizeof(float) > lds_budget()) return 0;
    const int RPT4 = WAVE / K;
    long long bl = ((N + RPT4 - 1) / RPT4 + nw4 - 1) / nw4;
    if (bl > 256) bl = 256;
    return (int)bl;
}

int vmp_svae_estep_fwd_rng_epi(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                               uint64_t seed, const uint64_t* seed_dev, const float* mk, const float* Wk, const float* kappa,
                               const float* nu, int64_t N, int K, int L, int S, float* x, float* lz, float* Tp,
                               float* x_samples, float* r, double* mom, size_t mom_bytes, void* stream) {
    int rc = check_sv(N, K, L, S);
    if (rc) return rc;
    if (!eta1 || !eta2d || !hk || !Pk || !bias || !mk || !Wk || !kappa || !x || !lz || !Tp || !x_samples) {
        set_error("vmp_svae_estep_fwd_rng_epi: null pointer");
        return VMP_E_BADARG;
    }
    if (!vmp_svae_rng_in_kernel(K, L, S)) {
        set_error("vmp_svae_estep_fwd_rng_epi: K=%d L=%d S=%d is outside the in-kernel generator's shapes (vmp_svae_rng_in_kernel)", K, L, S);
        return VMP_E_DIM;
    }
    if (mom) {
        const int nb = vmp_svae_fwd_mom_blocks(N, K, L, S);
        if (nb < 1) { set_error("vmp_svae_estep_fwd_rng_epi: no in-kernel moments for K=%d L=%d S=%d (vmp_svae_fwd_mom_blocks)", K, L, S); return VMP_E_DIM; }
        if (mom_bytes < (size_t)nb * 16 * MOMF * sizeof(double)) { set_error("vmp_svae_estep_fwd_rng_epi: moment buffer too small"); return VMP_E_WS; }
    }
    EFwdArgs a{eta1, eta2d, hk, Pk, bias, nullptr, mk, Wk, kappa, nu, x, lz, Tp, N, K, S, 0, (unsigned long long)seed,
               reinterpret_cast<const unsigned long long*>(seed_dev), x_samples, r, mom};
    a.vec_ok = al16(x);
    return run_fwd(a, L, stream, true);
}

// ---- round 6, the minibatch training step: E-step backward with the ELBO's scalar tail inside (svae_estep_bwd1_kernel<L, true>).
// Replaces, for batches the minibatch form covers (vmp_svae_bwd_tail_applies), the tail launch (vmp_svae_elbo_tail / the tail blocks
// of vmp_decoder_elbo) + vmp_svae_estep_bwd_n: dLoss/dlog_z, dLoss/dT' never reach memory.  Outputs as vmp_svae_estep_bwd_n
// (partials: one row per tile = vmp_svae_bwd_blocks_for) plus r = exp(log_z) (N,K) and tail_part (tiles, 2) fp64 - the per-tile
// terms of [sum w A, sum r (T' + log z)], summed by vmp_svae_step_final.
int vmp_svae_bwd_tail_applies(int64_t N, int K, int L, int S) {
    return N > 0 && K >= 1 && K <= WAVE && L >= 1 && L <= 8 && S >= 1 && bwd1_applies(N, K, L, S, false) ? 1 : 0;
}

int vmp_svae_estep_bwd_tail(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                            const float* mk, const float* Wk, const float* x, const float* lz, const float* T_prime, const float* ll,
                            float sigma, const float* Gx, int64_t N, int K, int L, int S, float* g_eta1, float* g_eta2d,
                            float* partials, size_t partial_bytes, float* r, double* tail_part, size_t tail_bytes, void* stream) {
    int rc = check_sv(N, K, L, S);
    if (rc) return rc;
    if (!eta1 || !eta2d || !hk || !Pk || !bias || !mk || !Wk || !x || !lz || !T_prime || !ll || !Gx || !g_eta1 || !g_eta2d || !partials ||
        !r || !tail_part || sigma == 0.f) {
        set_error("vmp_svae_estep_bwd_tail: null pointer or sigma == 0");
        return VMP_E_BADARG;
    }
    if (!vmp_svae_bwd_tail_applies(N, K, L, S)) {
        set_error("vmp_svae_estep_bwd_tail: N=%lld K=%d L=%d S=%d outside the minibatch form (<= %d tiles, S <= %d)", (long long)N, K, L, S,
                  BWD1_MAX_TILES, 2 * BWD1_MAX_PAIRS);
        return VMP_E_DIM;
    }
    const int nt = (int)((N + WAVE / K - 1) / (WAVE / K));
    const int PW = vmp_svae_bwd_partial_words(L);
    if (partial_bytes < (size_t)nt * K * PW * sizeof(float) || tail_bytes < (size_t)nt * 2 * sizeof(double)) {
        set_error("vmp_svae_estep_bwd_tail: partials / tail_part buffer too small for %d tiles", nt);
        return VMP_E_WS;
    }
    EBwdArgs a{eta1, eta2d, hk, Pk, bias, mk, Wk, nullptr, x, lz, Gx, nullptr, nullptr, g_eta1, g_eta2d, partials, N, K, S, 0};
    a.Tp = T_prime; a.ll = ll; a.r_out = r; a.tail_part = tail_part; a.sigma = sigma;
    return svae_bwd1_launch(a, L, nt, (S + 1) / 2, true, stream);
}

int vmp_svae_bwd_blocks_for(int64_t N, int K, int L, int S, int student) {
    if (N <= 0 || K < 1 || K > WAVE) return 0;
    if (bwd1_applies(N, K, L, S, student != 0)) return (int)((N + WAVE / K - 1) / (WAVE / K));
    return sv_blocks_l(N, K, L);                              // L <= 3 (generic kernel): more blocks per CU; otherwise vmp_svae_bwd_blocks
}

int vmp_svae_estep_bwd_n(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                         const float* mk, const float* Wk, const float* nu, const float* x, const float* lz, const float* Gx,
                         const float* Glz, const float* GT, int64_t N, int K, int L, int S, float* g_eta1, float* g_eta2d,
                         float* partials, size_t partial_bytes, int nblk, void* stream) {
    int rc = check_sv(N, K, L, S);
    if (rc) return rc;
    if (!eta1 || !eta2d || !hk || !Pk || !bias || !mk || !Wk || !x || !lz || !Gx || !Glz || !GT || !g_eta1 || !g_eta2d || !partials) {
        set_error("vmp_svae_estep_bwd: null pointer");
        return VMP_E_BADARG;
    }
    const int blocks = sv_blocks(N, K);
    const int PW = vmp_svae_bwd_partial_words(L);
    const bool use1 = bwd1_applies(N, K, L, S, nu != nullptr) && nblk == vmp_svae_bwd_blocks_for(N, K, L, S, nu != nullptr);
    const bool wide = !use1 && L <= 3 && nblk == sv_blocks_l(N, K, L);        // small L: the generic kernel on more blocks per CU
    if (!use1 && !wide && nblk != blocks) {
        set_error("vmp_svae_estep_bwd_n: nblk = %d is neither vmp_svae_bwd_blocks_for (%d) nor vmp_svae_bwd_blocks (%d)", nblk,
                  vmp_svae_bwd_blocks_for(N, K, L, S, nu != nullptr), blocks);
        return VMP_E_BADARG;
    }
    if (partial_bytes < (size_t)nblk * K * PW * sizeof(float)) { set_error("vmp_svae_estep_bwd: partials buffer too small"); return VMP_E_WS; }
    EBwdArgs a{eta1, eta2d, hk, Pk, bias, mk, Wk, nu, x, lz, Gx, Glz, GT, g_eta1, g_eta2d, partials, N, K, S, 0};
    a.vec_ok = al16(x) && al16(Gx);
#ifdef VMP_DEBUG_TS
    a.dbg_t = g_dbg_svae;
#endif
#ifndef VMP_T2_RING
#define VMP_T2_RING 1         // 0: build without the LDS-ring backward kernel (A/B measurements: tools/build_variant.sh)
#endif
    const long long ntiles_g = (N + WAVE / K - 1) / (WAVE / K);
    if (use1) {
        // minibatch sizes, Gaussian theta: one block per tile, one wave per sample pair (svae_estep_bwd1_kernel)
        const int P = (S + 1) / 2;
        rc = -1;
        rc = svae_bwd1_launch(a, L, (int)ntiles_g, P, false, stream);
        return rc;
    }
    const int gblocks = wide ? nblk : blocks;
    const bool one = ntiles_g <= (long long)gblocks * SV_NW;      // every wave has at most one tile: latency form
    if (VMP_T2_RING && (K == 16 || !one)) {
        // LDS-ring kernels (vmp_svae_ring.hip: quad-coalesced LDS-DMA of sample pairs, two pairs in flight per wave; 8 <= K <= 16,
        // even L >= 4, even S >= 4, Gaussian or Student-t theta).  Batches of one tile per wave with K != 16 keep the generic
        // kernel's latency form (tuned at the reference's minibatch size).
        rc = svae_bwd_ring_launch(a, L, blocks, stream);
        if (rc != -2) return rc;
    }
    const int PWa = nu ? PW : PW / 2;
    const size_t lds = (size_t)(K * ((L * (L + 1) / 2) | 1) + SV_NW * WAVE + SV_NW * 2 * L * SV_AST + SV_NW * PWa * SV_AST) * sizeof(float);
    rc = -1;
    VMP_DISPATCH_L(L, {
        if (one) hipLaunchKernelGGL((svae_estep_bwd_kernel<LL, true>), dim3(gblocks), dim3(SV_NW * WAVE), lds, static_cast<hipStream_t>(stream), a);
        else hipLaunchKernelGGL((svae_estep_bwd_kernel<LL, false>), dim3(gblocks), dim3(SV_NW * WAVE), lds, static_cast<hipStream_t>(stream), a);
        rc = check_launch("svae_estep_bwd_kernel");
    });
    return rc;
}

int vmp_svae_estep_bwd(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                       const float* mk, const float* Wk, const float* nu, const float* x, const float* lz, const float* Gx,
                       const float* Glz, const float* GT, int64_t N, int K, int L, int S, float* g_eta1, float* g_eta2d,
                       float* partials, size_t partial_bytes, void* stream) {
    return vmp_svae_estep_bwd_n(eta1, eta2d, hk, Pk, bias, mk, Wk, nu, x, lz, Gx, Glz, GT, N, K, L, S, g_eta1, g_eta2d, partials,
                                partial_bytes, N > 0 && K >= 1 && K <= WAVE ? sv_blocks(N, K) : 0, stream);
}

static int subsample_impl(const char* what, const float* x, const float* lz, const float* u, const int64_t* z, int rng, uint64_t seed,
                          const uint64_t* seed_dev, int64_t N, int K, int S, int L, int S_out, float* out, int64_t* z_out,
                          void* stream) {
    if (!x || !lz || (!u && !z && !rng) || !out || N <= 0 || K <= 0 || S <= 0 || L <= 0 || S_out <= 0 || S_out > S) {
        set_error("%s: bad argument", what);
        return VMP_E_BADARG;
    }
    SubArgs a{x, lz, u, reinterpret_cast<const long long*>(z), out, reinterpret_cast<long long*>(z_out), N, K, S, L, S_out,
              rng, (unsigned long long)seed, reinterpret_cast<const unsigned long long*>(seed_dev)};
    if (K > WAVE) { set_error("%s: K=%d > 64", what, K); return VMP_E_DIM; }
    const int RPTs = WAVE / K;
    long long blocks = ((N + RPTs - 1) / RPTs + 15) / 16;               // 4 waves per block, 4 tiles of RPT rows per wave and turn
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(subsample_kernel, dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch(what);
}

int vmp_svae_subsample(const float* x, const float* lz, const float* u, const int64_t* z, int64_t N, int K, int S, int L,
                       int S_out, float* out, int64_t* z_out, void* stream) {
    return subsample_impl("vmp_svae_subsample", x, lz, u, z, 0, 0, nullptr, N, K, S, L, S_out, out, z_out, stream);
}

int vmp_svae_subsample_rng(const float* x, const float* lz, uint64_t seed, const uint64_t* seed_dev, int64_t N, int K, int S,
                           int L, int S_out, float* out, int64_t* z_out, void* stream) {
    return subsample_impl("vmp_svae_subsample_rng", x, lz, nullptr, nullptr, 1, seed, seed_dev, N, K, S, L, S_out, out, z_out, stream);
}

}  // extern "C"
