/*
 * vmp_hip.h - C ABI of libvmp_hip.so: the MI355X (gfx950) implementation of the VMP hot path of
 * emtiyaz/vmp-for-svae (GMM / SMM structured VAE).
 *
 * The reference has no FFI: its boundary is the Python function surface of distributions/{gaussian,niw,dirichlet,student_t}.py and
 * models/{gmm,smm,svae,vae}.py.  This library sits UNDER the package's mirror of that
 * surface (vmp-for-svae_amd/{distributions,models}); each entry point names the reference code it replaces.
 *
 * Conventions (all entry points)
 *   - plain C symbols; every pointer is a DEVICE pointer unless the name ends in _host;
 *   - row-major, batch-leading fp32 tensors exactly as the reference lays them out: x (N,D), r (N,K),
 *     (K,D,D), (N,K,S,L) ...; integer sizes: N int64, everything else int;
 *   - caller allocates inputs, outputs and workspace; the library never allocates device memory (one exception: vmp_exch_alloc, the uncached IPC exchange buffer of the peer form);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return 0 = ok, <0 = invalid argument (VMP_E_*), >0 = hipError_t of a failed launch;
 *     vmp_last_error() returns a thread-local message for the last non-zero return;
 *   - no global mutable state: callable concurrently from any host thread.
 */
#ifndef VMP_HIP_H
#define VMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMP_ABI_VERSION 1

#define VMP_E_BADARG   (-1)   /* null pointer / non-positive size            */
#define VMP_E_DIM      (-2)   /* D or K outside the compiled range           */
#define VMP_E_WS       (-3)   /* workspace too small                         */

#define VMP_MAX_D 8           /* latent / data dimension of the mixture (compiled range 1..8)  */
#define VMP_MAX_K 64          /* mixture components                                             */

/* mixture flavour */
#define VMP_GMM 0             /* models/gmm.py  (Bishop 10.2)                */
#define VMP_SMM 1             /* models/smm.py  (Archambeau & Verleysen)     */

int         vmp_abi_version(void);
const char* vmp_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * T1: pure mixture VMP (models/gmm.py:25-269, models/smm.py:25-245)
 * ------------------------------------------------------------------------------------------------ */

/* Number of fp32 words per component of the E-step parameter pack written by vmp_mix_finalize and
 * consumed by vmp_mix_estep: [ m_k (D) | W_k packed lower-triangular, row-major (D(D+1)/2) | c | h | ua | ub ]
 * with  q_nk = || W_k (x_n - m_k) ||^2,  log rho_nk = c - h*q_nk,  u_nk = ua / (q_nk + ub).           */
int    vmp_mix_pack_words(int D);

/* Raw sufficient statistics layout (fp64), per component k:  [ Nk | Wk | sx (D) | sxx (D*D, symmetric) ].
 * Nk = sum_n r_nk, Wk = sum_n w_nk, sx = sum_n w_nk x_n, sxx = sum_n w_nk x_n x_n^T, w = r (GMM) or r*u (SMM). */
int    vmp_mix_stats_words(int D);

/* Bytes of workspace needed by vmp_mix_stats / vmp_mix_estep(with stats) for N rows. */
size_t vmp_mix_workspace_bytes(int64_t N, int D, int K);

/* M-pass over the data: weighted raw moments.
 * Replaces the N-sized part of gmm.update_Nk/xk/Sk (models/gmm.py:25-46) and smm.update_Nk/Wk/xk/Sk
 * (models/smm.py:25-50); the centring (x - x_k) the reference does in a second pass is done in fp64 in
 * vmp_mix_finalize / on the host side from the raw moments.
 *   x (N,D), r (N,K), u (N,K) or NULL (then w = r).   stats: (K, vmp_mix_stats_words(D)) fp64, overwritten.  */
int    vmp_mix_stats(const float* x, const float* r, const float* u, const float* pivot, int64_t N, int D, int K,
                     double* stats, void* ws, size_t ws_bytes, void* stream);

/* Pivot for the moment accumulation (D floats): the mean of up to 4096 evenly strided rows of x.  The moment
 * kernels accumulate products of (x - pivot) in fp32 and the finalize kernels un-shift exactly in fp64, which
 * makes the one-pass raw-moment form as accurate as the reference's two-pass centred update_Sk (gmm.py:39-46)
 * for data far from the origin.  `pivot` arguments below may be NULL (no shift).                           */
int    vmp_mix_pivot(const float* x, int64_t N, int D, float* pivot_out, void* stream);

/* K-sized posterior update + E-step parameter pack, all in fp64 on the device, rounded to fp32 on output.
 * Replaces gmm.update_alphak/betak/mk/Ck/vk (models/gmm.py:49-81), P_k = matrix_inverse(C_k) (gmm.py:260),
 * compute_expct_log_det_prec (gmm.py:117-131, including its det<=1e-20 guard), compute_log_pi (gmm.py:134-138);
 * SMM flavour: models/smm.py:53-85, 99-116 and the constants of compute_rnk/compute_expct_unk (smm.py:119-137).
 *   prior (standard form): alpha0 (K), beta0 (K), m0 (K,D), C0 (K,D,D), v0 (K); kappa (K) or NULL (GMM).
 *   outputs (any may be NULL): alpha,beta,v (K); m, xbar (K,D); C, S (K,D,D); pi = exp(E log pi) (K);
 *   pack (K, vmp_mix_pack_words(D)).                                                                     */
int    vmp_mix_finalize(const double* stats, int D, int K, int flavour,
                        const float* alpha0, const float* beta0, const float* m0, const float* C0, const float* v0,
                        const float* kappa,
                        float* alpha, float* beta, float* m, float* C, float* v, float* xbar, float* S, float* pi,
                        float* pack, void* stream);

/* E-step parameter pack from explicit posterior parameters (for the stand-alone e_step API):
 * gmm.e_step(x, alpha_k, beta_k, m_k, P_k, v_k) (models/gmm.py:154-174) / smm.e_step (models/smm.py:140-164).
 *   P (K,D,D) is the precision the reference passes in.                                                   */
int    vmp_mix_pack_from_params(int D, int K, int flavour, const float* alpha, const float* beta, const float* m,
                                const float* P, const float* v, const float* kappa, float* pack, float* pi,
                                void* stream);

/* E-pass over the data: responsibilities (and SMM scales), optionally FUSED with the raw moments of the
 * NEW responsibilities (= the next M-pass), so that one VMP iteration streams x once and writes r once.
 * Replaces compute_expct_mahalanobis_dist + compute_rnk (models/gmm.py:84-94,141-151), the missing-data variant
 * (gmm.py:97-114) when miss_mask != NULL, and smm.expct_mahalanobis_dist/compute_rnk/compute_expct_unk
 * (models/smm.py:88-96,119-137).
 *   x (N,D); pack from vmp_mix_finalize / vmp_mix_pack_from_params; miss_mask (N,D) uint8 or NULL;
 *   r_out (N,K); u_out (N,K) (SMM, else NULL); logr_out (N,K) or NULL (= log r_out as gmm.py:267);
 *   stats_out (K, vmp_mix_stats_words(D)) fp64 or NULL; ws needed only when stats_out != NULL.            */
int    vmp_mix_estep(const float* x, int64_t N, int D, int K, int flavour, const float* pack,
                     const uint8_t* miss_mask, float* r_out, float* u_out, float* logr_out,
                     const float* pivot, double* stats_out, void* ws, size_t ws_bytes, void* stream);

/* Fast path of the VMP iteration - exactly two launches per iteration, no intermediate stats buffer:
 *   vmp_mix_estep_fused : the E-pass kernel with fused raw moments; leaves per-block fp64 partials in `ws`
 *   vmp_mix_finalize_ws : reduces those partials in a fixed order (deterministic), then does what
 *                         vmp_mix_finalize does; stats_out (K, stats_words) is optional.
 * Both must be given the same (N, D, K, flavour) and the same ws.  vmp_mix_stats_ws is the stand-alone
 * M-pass leaving partials in ws (first iteration).  Reference: the loop body of gmm.inference
 * (models/gmm.py:258-263) / smm.inference (models/smm.py:232-238).                                          */
int    vmp_mix_estep_fused(const float* x, int64_t N, int D, int K, int flavour, const float* pack,
                           float* r_out, float* u_out, float* logr_out, const float* pivot,
                           void* ws, size_t ws_bytes, void* stream);
int    vmp_mix_stats_ws(const float* x, const float* r, const float* u, const float* pivot, int64_t N, int D, int K,
                        void* ws, size_t ws_bytes, void* stream);
int    vmp_mix_finalize_ws(const void* ws, const float* pivot, int64_t N, int D, int K, int flavour,
                           const float* alpha0, const float* beta0, const float* m0, const float* C0, const float* v0,
                           const float* kappa,
                           float* alpha, float* beta, float* m, float* C, float* v, float* xbar, float* S, float* pi,
                           float* pack, double* stats_out, void* stream);

/* `iterations` VMP iterations enqueued back-to-back from one call (2 launches each, no host round trip):
 * the loop `for i in range(nb_iters): sess.run(update)` of models/gmm.py:377-379.  ws must hold the partial moments
 * of the current (r, u) (vmp_mix_stats_ws or a previous iteration).                                             */
/* Opt-in ACCURATE E-part (round 6).  The SMM's log rho_nk = c_k - (D + kappa)/2 q_nk (models/smm.py:119-128) is LINEAR in the
 * expected Mahalanobis distance with a factor 6.5 at D = 8, kappa = 5: rows without a close component have |log rho| ~ 1e2..1e3,
 * where an fp32 q (absolute error 1e-7 q) and an fp32 log rho cost 1..4e-5 on r_nk - above the 1e-5 the north-star states.
 * vmp_mix_finalize_ws64 = vmp_mix_finalize_ws that also writes the E-step pack in fp64 (pack64: K x vmp_mix_pack_words(D) doubles);
 * vmp_mix_estep_accurate evaluates compute_expct_mahalanobis_dist / compute_rnk / compute_expct_unk (models/gmm.py:84-94,141-151,
 * models/smm.py:88-96,119-137) from it entirely in fp64 and rounds r (u, log r) once on the way out.  It does NOT accumulate
 * moments: follow it with vmp_mix_stats_ws_accurate - gmm.update_Nk/xk/Sk (gmm.py:25-46) / smm.update_* (smm.py:25-50) with every
 * product and sum in fp64, into the same workspace partials vmp_mix_stats_ws leaves (a 5e-8 relative error of P_k, the level of the
 * default moments, is 1e-4 on the SMM's log rho of rows without a close component).  Three streaming launches per iteration
 * instead of one - the default stays the fused pass.                                                                       */
int    vmp_mix_finalize_ws64(const void* ws, const float* pivot, int64_t N, int D, int K, int flavour,
                             const float* alpha0, const float* beta0, const float* m0, const float* C0, const float* v0,
                             const float* kappa, float* alpha, float* beta, float* m, float* C, float* v, float* xbar,
                             float* S, float* pi, float* pack, double* pack64, double* stats_out, void* stream);
int    vmp_mix_estep_accurate(const float* x, int64_t N, int D, int K, int flavour, const double* pack64, float* r_out,
                              float* u_out, float* logr_out, void* stream);
int    vmp_mix_stats_ws_accurate(const float* x, const float* r, const float* u, const float* pivot, int64_t N, int D, int K,
                                 void* ws, size_t ws_bytes, void* stream);
int    vmp_mix_iterate(const float* x, int64_t N, int D, int K, int flavour,
                       const float* alpha0, const float* beta0, const float* m0, const float* C0, const float* v0,
                       const float* kappa, const float* pivot, float* r, float* u,
                       float* alpha, float* beta, float* m, float* C, float* v, float* xbar, float* S, float* pi,
                       float* pack, void* ws, size_t ws_bytes, int iterations, void* stream);

/* ------------------------------------------------------------------------------------------------
 * T2: SVAE E-step fused with the ELBO regulariser (models/svae.py:14-119 and :229-252)
 * ------------------------------------------------------------------------------------------------
 * Per (n,k) cell:  Pt = diag(-2 eta2d_n) + P_k,  ht = eta1_n + h_k,  Lt = chol(Pt),
 *   log_z_nk  = normalise_k( bias_k + 1/2 |Lt^-1 ht|^2 - sum_i log Lt_ii )        compute_log_z_given_y, svae.py:50-92
 *   x_nks     = Lt^-T (Lt^-1 ht + eps_nks)                                         sample_x_per_comp,     svae.py:95-119
 *   T'_nk     = mean_s[ log N(x_s; phi~_nk) - log N(x_s; theta_k) - E log pi_k ]   the two per-sample densities of
 *               compute_elbo (svae.py:236-243; gaussian.py:74-105) in closed form, so that
 *               regulariser = sum_nk exp(log_z_nk) (T'_nk + log_z_nk)               (svae.py:245-252)
 * Inputs: eta1, eta2d (N,L) encoder outputs; hk (K,L), Pk (K,L,L) symmetric, bias (K) = B_k + log pi_k from
 * unpack_recognition_gmm (svae.py:342-358); noise (N,K,L,S) replaces tf.random_normal (svae.py:114);
 * theta side: log p(x, z=k | theta) = kappa_k - 1/2 |W_k (x - m_k)|^2 with W_k (K,L,L) LOWER triangular,
 * W^T W = E[Sigma_k]^-1, kappa_k = sum log W_ii - L/2 log 2pi + E log pi_k (GMM theta, svae.py:205-214, no gradient);
 * with nu != NULL the Student-t of compute_elbo_smm (svae.py:265-322, student_t.py:31-37):
 * kappa_k - 1/2 (nu_k + L) log1p(|W_k (x - m_k)|^2 / nu_k), W = chol(Sigma_k)^-1, kappa_k = lgamma terms - sum log L_ii
 * + E log pi_k (mu_k, L_k trainable: experiments.py:160-161).  Outputs: x (N,K,S,L), lz (N,K), Tp (N,K).     */
int    vmp_svae_estep_fwd(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                          const float* noise, const float* mk, const float* Wk, const float* kappa, const float* nu,
                          int64_t N, int K, int L, int S, float* x, float* lz, float* Tp, void* stream);
/* In-kernel noise (models/svae.py:113-114 draws eps inside the step: tf.random_normal, i.e. TensorFlow's Philox stream).
 * Same operation with eps generated where it is consumed: Philox4x32-7 (Random123's philox4x32 at 7 rounds) keyed by `seed`,
 * counter = (cell low, cell high, block, 0), cell = n K + k, block b = (s >> 1) ceil(L/4) + j -> four Box-Muller pairs, one per
 * 32-bit word (radius uniform from the word's top 20 bits, angle from its low 12): (eps[i,s], eps[i,s+1]) for i = 4j .. 4j+3 of
 * the cell's (L,S) noise block (csrc/vmp_svae.hip, oracle/philox.py).  No (N,K,L,S) tensor is read or written; the backward pass needs
 * none (it works from the saved samples x).  Shapes outside the in-kernel path (vmp_svae_rng_in_kernel == 0: L < 8 and L*S
 * not a multiple of 4 or a cell tile larger than the LDS; L = 8 is covered for every S) materialise the same stream in `noise_ws` (N,K,L,S) first.
 * vmp_svae_philox_noise writes that stream as a tensor (tests; callers that want to keep the draw).               */
int    vmp_svae_rng_in_kernel(int K, int L, int S);
int    vmp_svae_philox_noise(uint64_t seed, int64_t N, int K, int L, int S, float* noise, void* stream);
int    vmp_svae_philox_noise_dev(const uint64_t* seed_dev, int64_t N, int K, int L, int S, float* noise, void* stream);   /* key from a device word */
int    vmp_svae_estep_fwd_rng(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                              uint64_t seed, const float* mk, const float* Wk, const float* kappa, const float* nu,
                              int64_t N, int K, int L, int S, float* x, float* lz, float* Tp, float* noise_ws, void* stream);
/* The same with the key read from a DEVICE word at execution time (in-kernel shapes only, vmp_svae_rng_in_kernel != 0): a
 * launch captured in a HIP graph draws fresh noise on every replay once the caller refreshes *seed_dev.            */
int    vmp_svae_estep_fwd_rng_dev(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                                  const uint64_t* seed_dev, const float* mk, const float* Wk, const float* kappa,
                                  const float* nu, int64_t N, int K, int L, int S, float* x, float* lz, float* Tp,
                                  void* stream);
/* The in-kernel-noise E-step with the step's next three operations done in its epilogue, while a cell's values are in
 * registers (in-kernel shapes only; key = `seed`, or *seed_dev when seed_dev != NULL):
 *   x_samples (N,L) = subsample_x(x, lz) with ONE draw per row (models/svae.py:122-151 as its caller uses it, svae.py:514:
 *                     z_n ~ Cat(exp lz_n), x_samples[n] = x[n, z_n, 0, :]) - bit-identical to vmp_svae_subsample_rng with the same
 *                     key and S_out = 1 (same uniforms, same inverse-CDF arithmetic);
 *   r (N,K)         = exp(lz) (svae.py:216; may be NULL);
 *   mom             = per-block fp64 partials of the M-step's raw moments sum_n r_nk [x_n | 1 | x_n x_n^T lower-packed | 0 0 0]
 *                     with x_n = x_samples[n] (svae.m_step -> gmm.update_Nk/xk/Sk, svae.py:154-176, gmm.py:25-46), laid out
 *                     (vmp_svae_fwd_mom_blocks(N,K,L,S), 16, 48); NULL = not wanted.  Exists for K = 16, L = 8
 *                     (vmp_svae_fwd_mom_blocks returns 0 otherwise: use vmp_mix_stats on x_samples and r).
 * vmp_svae_mom_cvi adds the partials in a fixed order into stats_out (K, 2+L+L*L) fp64 (vmp_mix_stats layout; may be NULL when
 * theta is given) and, when t_alpha != NULL, applies vmp_svae_cvi_update from them in the same launch.                    */
int    vmp_svae_fwd_mom_blocks(int64_t N, int K, int L, int S);
int    vmp_svae_estep_fwd_rng_epi(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                                  uint64_t seed, const uint64_t* seed_dev, const float* mk, const float* Wk, const float* kappa,
                                  const float* nu, int64_t N, int K, int L, int S, float* x, float* lz, float* Tp,
                                  float* x_samples, float* r, double* mom, size_t mom_bytes, void* stream);
int    vmp_svae_mom_cvi(const double* mom, int nblk, const float* p_alpha, const float* p_A, const float* p_b,
                        const float* p_beta, const float* p_vhat, float* t_alpha, float* t_A, float* t_b, float* t_beta,
                        float* t_vhat, float* s_alpha, float* s_A, float* s_b, float* s_beta, float* s_vhat,
                        const float* rho_dev, float rho, int K, int L, double* stats_out, void* stream);

/* Backward of the above: given dLoss/dx (N,K,S,L) (from the decoder), dLoss/dlog_z (N,K), dLoss/dT' (N,K), writes
 * dLoss/deta1, dLoss/deta2d (N,L) and per-block partial sums over n of dLoss/d{hk, Pk, bias}:
 * partials (vmp_svae_bwd_blocks(N,K), K, vmp_svae_bwd_partial_words(L)) = [ g_hk (L) | g_Pk lower triangle of the
 * symmetric gradient, row-major packed (L(L+1)/2) | g_bias | g_mk (L) | g_Wk lower packed | g_kappa ] (the last three
 * are zero except g_kappa unless nu != NULL); the caller sums them over the first axis.
 * Replaces TF autodiff through svae.py:50-119 and gaussian.py:74-105 (opt.compute_gradients, experiments.py:232). */
int    vmp_svae_bwd_partial_words(int L);
int    vmp_svae_bwd_blocks(int64_t N, int K);
size_t vmp_svae_workspace_bytes(int64_t N, int K, int L);
int    vmp_svae_estep_bwd(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                          const float* mk, const float* Wk, const float* nu, const float* x, const float* lz,
                          const float* Gx, const float* Glz, const float* GT, int64_t N, int K, int L, int S,
                          float* g_eta1, float* g_eta2d, float* partials, size_t partial_bytes, void* stream);
/* The same with the number of partial rows stated by the caller (round 6).  nblk = vmp_svae_bwd_blocks_for(N, K, L, S, nu != NULL)
 * selects the launch geometry of that query: at minibatch sizes (the reference's operating point, experiments.py:26) and a
 * Gaussian theta that is one block per tile of 64 / K rows with one wave per sample pair - one partial row per TILE instead of
 * per 4-wave block; nblk = vmp_svae_bwd_blocks(N, K) is vmp_svae_estep_bwd.  The caller reduces exactly nblk rows.     */
int    vmp_svae_bwd_blocks_for(int64_t N, int K, int L, int S, int student);
int    vmp_svae_estep_bwd_n(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                            const float* mk, const float* Wk, const float* nu, const float* x, const float* lz,
                            const float* Gx, const float* Glz, const float* GT, int64_t N, int K, int L, int S,
                            float* g_eta1, float* g_eta2d, float* partials, size_t partial_bytes, int nblk, void* stream);

/* subsample_x (models/svae.py:122-151): z_ns ~ Cat(exp lz_n) and x_samples[n,s,:] = x[n, z_ns, s, :] for s < S_out
 * (the reference draws all S and keeps s = 0, svae.py:514).  The draw is the inverse CDF of the supplied uniform
 * u (N,S_out) - replacing tf.multinomial - or the supplied index z (N,S_out) when z != NULL.  out (N,S_out,L).  */
int    vmp_svae_subsample(const float* x, const float* lz, const float* u, const int64_t* z, int64_t N, int K, int S,
                          int L, int S_out, float* out, int64_t* z_out, void* stream);
/* The same with the uniforms drawn in the kernel: u_ns = top 24 bits of word 0 of Philox4x32-7(key = seed, counter =
 * (n low, n high, s, 0x5bb5a3c1)) * 2^-24 - a stream apart from the E-step's normals (whose counter word 3 is 0).  The key
 * is `seed`, or *seed_dev when seed_dev != NULL (graph-captured steps).                                              */
int    vmp_svae_subsample_rng(const float* x, const float* lz, uint64_t seed, const uint64_t* seed_dev, int64_t N, int K,
                              int S, int L, int S_out, float* out, int64_t* z_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K-sized parameter maps of the training step, one launch each (one thread per component, fp64 inside)
 * ------------------------------------------------------------------------------------------------
 * vmp_svae_phi_prep_fwd : svae.unpack_recognition_gmm (svae.py:342-358) + the k-only part of compute_log_z_given_y
 *   (svae.py:70-92):  L_k = tril(L_raw), softplus on the diagonal;  P_k = L_k L_k^T (= -2 eta2);
 *   bias_k = -1/2 |L_k^-1 mu_k|^2 + sum_i log (L_k)_ii + log softmax(pi_raw)_k  - the inputs of vmp_svae_estep_fwd.
 * vmp_svae_phi_prep_bwd : its adjoint: (g_hk, g_P (w.r.t. the full matrix), g_bias) -> gradients w.r.t. the three
 *   'phi_gmm' variables (what TF's autodiff does through tril / softplus / matmul / matrix_solve / softmax).
 * vmp_svae_theta_pack   : theta side of compute_elbo (svae.py:205-214): niw.natural_to_standard + expected_values
 *   (niw.py:8-43), dirichlet.expected_log_pi (dirichlet.py:8-22) -> m_k, W_k = chol(E[Sigma_k])^-1 (lower),
 *   kappa_k = sum_i log W_ii - L/2 log 2pi + E log pi_k   (no gradient: stop_gradient in the reference).
 * vmp_svae_cvi_update   : svae.m_step in natural parameters (svae.py:154-176 = prior + raw moments, +1 on v_hat,
 *   SURVEY appendix A.6) and update_gmm_params (svae.py:376-403): theta <- (1-rho) theta + rho theta*, in place;
 *   theta* is also written when the s_* pointers are given.  stats = vmp_mix_stats layout (fp64).
 *   rho_dev != NULL reads the step size from device memory (graph-captured steps).                            */
int    vmp_svae_phi_prep_fwd(const float* mu_k, const float* L_raw, const float* pi_raw, int K, int L, float* Lk,
                             float* P, float* bias, void* stream);
int    vmp_svae_phi_prep_bwd(const float* mu_k, const float* L_raw, const float* pi_raw, const float* g_hk,
                             const float* g_P, const float* g_bias, int K, int L, float* g_mu, float* g_Lraw,
                             float* g_piraw, void* stream);
/* Fixed-order fp64 reduction of vmp_svae_estep_bwd's per-block partials into the K-sized gradients: g_hk (K,L),
 * g_P (K,L,L, symmetric), g_bias (K) and - Student-t theta only, else NULL - g_mk (K,L), g_W (K,L,L, lower), g_kappa (K). */
int    vmp_svae_bwd_reduce(const float* partials, int nblk, int K, int L, float* g_hk, float* g_P, float* g_bias,
                           float* g_mk, float* g_W, float* g_kappa, void* stream);
int    vmp_svae_theta_pack(const float* alpha, const float* A, const float* b, const float* beta, const float* v_hat,
                           int K, int L, float* m, float* W, float* kappa, void* stream);
/* vmp_svae_phi_prep_fwd and vmp_svae_theta_pack (independent K-sized maps, both run once per training step) in ONE launch. */
int    vmp_svae_prep_fwd(const float* mu_k, const float* L_raw, const float* pi_raw, const float* alpha, const float* A,
                         const float* b, const float* beta, const float* v_hat, int K, int L, float* Lk, float* P,
                         float* bias, float* m, float* W, float* kappa, void* stream);
/* Small batches (N <= 512 rows: the reference's minibatches), one process: the M-step moments of (x_samples (N,L), r (N,K))
 * - exactly vmp_mix_stats' small-batch sums, written to stats_out (K, 2+L+L*L) fp64 - and vmp_svae_cvi_update from them in
 * ONE launch.  Arguments as vmp_svae_cvi_update.                                                                      */
int    vmp_svae_stats_cvi(const float* x_samples, const float* r, int64_t N, const float* p_alpha, const float* p_A,
                          const float* p_b, const float* p_beta, const float* p_vhat, float* t_alpha, float* t_A,
                          float* t_b, float* t_beta, float* t_vhat, float* s_alpha, float* s_A, float* s_b, float* s_beta,
                          float* s_vhat, const float* rho_dev, float rho, int K, int L, double* stats_out, void* stream);
int    vmp_svae_cvi_update(const double* stats, const float* p_alpha, const float* p_A, const float* p_b,
                           const float* p_beta, const float* p_vhat, float* t_alpha, float* t_A, float* t_b,
                           float* t_beta, float* t_vhat, float* s_alpha, float* s_A, float* s_b, float* s_beta,
                           float* s_vhat, const float* rho_dev, float rho, int K, int L, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Reconstruction term (models/vae.py:201-250, weights branch :233-248)
 * ------------------------------------------------------------------------------------------------
 * A_nk = sum_{s,d} [ (y_nd - mean_nksd)^2 / var_nksd + log(var_nksd + 1e-8) ]   -- the tensor the reference
 * contracts with the responsibilities in einsum('nksd,nk->') (vae.py:240).  y (N,Dy); mean, var (N,K,S,Dy).
 * Backward: gmean = gA_nk * d/dmean, gvar = gA_nk * d/dvar (same shapes as mean / var).                        */
/* eps: the reference adds 1e-8 inside the log in the weights branch (vae.py:240) and nothing in the plain-VAE branch
 * (weights=None, vae.py:225; call with K = 1, means (M,1,S,L)).                                                    */
int    vmp_diag_gauss_loglike_fwd(const float* y, const float* mean, const float* var, int64_t N, int K, int S,
                                  int Dy, float eps, float* A, void* stream);
int    vmp_diag_gauss_loglike_bwd(const float* y, const float* mean, const float* var, const float* gA,
                                  int64_t N, int K, int S, int Dy, float eps, float* gmean, float* gvar, void* stream);

/* Bernoulli decoder: rows_nks = sum_d m_nd * ( -log(1 + exp(-logit_nksd * y_nd)) ), y in {-1,+1},
 * m = 1 or the missing-data mask (N,D) - the per-sample-row part of vae.expected_bernoulli_loglike
 * (models/vae.py:175-198) and losses.bernoulli_logprob (losses.py:41-80).  logits (N,K,S,D); rows, g_rows (N,K,S).   */
int    vmp_bernoulli_rows_fwd(const float* y, const float* logits, const uint8_t* mask, int64_t N, int K, int S, int D,
                              float* rows, void* stream);
int    vmp_bernoulli_rows_bwd(const float* y, const float* logits, const uint8_t* mask, const float* g_rows, int64_t N,
                              int K, int S, int D, float* g_logits, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused decoder MLP + reconstruction term (models/vae.py:75-128 make_nnet, :138-151 make_decoder, :233-248)
 * ------------------------------------------------------------------------------------------------
 * The decoder the SVAE driver builds (experiments.py:140: [(U,tanh),(U,tanh),(Dy,'standard')]) applied to the
 * N*K*S sample rows x (N,K,S,L), fused with the per-row part of vae.expected_diagonal_gaussian_loglike:
 *   h0 = tanh(x W0 + b0), h1 = tanh(h0 W1 + b1), [raw1|raw2] = h1 W2 + b2            (vae.py:17-25, 86-93)
 *   mean = raw1 + x Ws + bs1,  var = softplus(raw2) + log1p(exp(bs2))                 (vae.py:28-49, 97-116)
 *   ll_nks = sum_d [ (y_nd - mean_d)^2 / var_d + log(var_d + 1e-8) ]                  (vae.py:236-240)
 * so that A_nk of vmp_diag_gauss_loglike_fwd = sum_s ll_nks, without the (rows x U) activations or the
 * (N,K,S,Dy) decoder outputs ever reaching HBM.  fp32 MFMA (v_mfma_f32_16x16x4_f32), L, Dy <= 8, U <= 64.
 * Parameters in the reference's variable layout: W0 (L,U) 'layer_0/kernel', b0 (U), W1 (U,U), b1 (U),
 * W2 (U,2Dy) 'gaussian_output/kernel', b2 (2Dy), Ws (L,Dy) 'shortcut/W', bs1 (Dy) 'shortcut/b1', bs2 (Dy).
 *   fwd: ll (N,K,S) and/or (mean, var) (N,K,S,Dy) - either may be NULL (K = S = 1 gives the plain decoder);
 *        y may be NULL when ll is.
 *   bwd: given gA (N,K) = dLoss/dA_nk writes dx (N,K,S,L) and the flat parameter gradient
 *        dparams [W0|b0|W1|b1|W2|b2|Ws|bs1|bs2] (vmp_decoder_param_words floats); deterministic.  With ll != NULL
 *        the same pass also writes ll (N,K,S) - value and gradient from one launch, for callers that know gA
 *        beforehand (the ELBO's dLoss/dA_nk = r_nk / 2S does not depend on the decoder).                     */
int    vmp_decoder_param_words(int L, int U, int Dy);
size_t vmp_decoder_workspace_bytes(int64_t N, int K, int S, int L, int U, int Dy);
int    vmp_decoder_loglike_fwd(const float* x, const float* y, const float* W0, const float* b0, const float* W1,
                               const float* b1, const float* W2, const float* b2, const float* Ws, const float* bs1,
                               const float* bs2, int64_t N, int K, int S, int L, int Dy, int U, float* ll, float* mean,
                               float* var, void* stream);
/* Backward pass of the same network as a stand-alone Gaussian-head MLP (the encoder, vae.make_encoder
 * models/vae.py:131-135; forward = vmp_decoder_loglike_fwd with ll == NULL, K = S = 1): the upstream gradients of the two
 * head outputs (mean, var) are inputs.  x (R,L); gmean, gvar (R,Dy); dx (R,L) may be NULL (x is data).        */
int    vmp_mlp_gauss_bwd(const float* x, const float* gmean, const float* gvar, const float* W0, const float* b0,
                         const float* W1, const float* b1, const float* W2, const float* b2, const float* Ws,
                         const float* bs1, const float* bs2, int64_t R, int L, int Dy, int U, float* dx, float* dparams,
                         void* ws, size_t ws_bytes, void* stream);
/* The same MLP with either Gaussian head of models/vae.py:28-50 (make_gaussian_layer): outputs (out1, out2) =
 * (mean, var_scale * var) - var_scale = 1: 'standard' (mean, softplus), var_scale = -1/2: the encoder's 'natparam' head
 * (eta1, -1/2 softplus) of experiments.py:139 - so that the head's scaling costs no launch of its own, forward or
 * backward (g_out2 is the upstream gradient of out2).  x (R,L); out1, out2, g_out1, g_out2 (R,Dy).              */
int    vmp_mlp_gauss_head_fwd(const float* x, const float* W0, const float* b0, const float* W1, const float* b1,
                              const float* W2, const float* b2, const float* Ws, const float* bs1, const float* bs2,
                              int64_t R, int L, int Dy, int U, float var_scale, float* out1, float* out2, void* stream);
int    vmp_mlp_gauss_head_bwd(const float* x, const float* g_out1, const float* g_out2, float var_scale, const float* W0,
                              const float* b0, const float* W1, const float* b1, const float* W2, const float* b2,
                              const float* Ws, const float* bs1, const float* bs2, int64_t R, int L, int Dy, int U,
                              float* dx, float* dparams, void* ws, size_t ws_bytes, void* stream);
int    vmp_decoder_loglike_bwd(const float* x, const float* y, const float* gA, const float* W0, const float* b0,
                               const float* W1, const float* b1, const float* W2, const float* b2, const float* Ws,
                               const float* bs1, const float* bs2, int64_t N, int K, int S, int L, int Dy, int U,
                               float* dx, float* dparams, float* ll, void* ws, size_t ws_bytes, void* stream);
/* As vmp_decoder_loglike_bwd with the upstream gradient given through LOG weights: gA_nk = w_scale * exp(log_w_nk).  The
 * ELBO's weights are r_nk = exp(log z_nk) (models/svae.py:216-219, vae.py:240): with w_scale = -sigma / 2S the launch
 * returns sigma * d rec / d(x, parameters) without a separate exp / scaling pass over (N,K).  w_scale != 0.          */
int    vmp_decoder_loglike_bwd_logw(const float* x, const float* y, const float* log_w, float w_scale, const float* W0,
                                    const float* b0, const float* W1, const float* b1, const float* W2, const float* b2,
                                    const float* Ws, const float* bs1, const float* bs2, int64_t N, int K, int S, int L,
                                    int Dy, int U, float* dx, float* dparams, float* ll, void* ws, size_t ws_bytes,
                                    void* stream);

/* compute_elbo for the fused decoder in THREE launches: vmp_decoder_loglike_bwd_logw with w_scale = -sigma / 2S, then ONE
 * launch in which some blocks reduce the decoder's parameter partials and the others run the (N,K) pass of
 * vmp_svae_elbo_tail (below) - both wait for the decoder kernel only - then the one-wave sum of the tail's partials.  Arguments as those two calls; tail_ws as vmp_svae_elbo_tail's ws.  N >= 1.   */
int    vmp_decoder_elbo(const float* x, const float* y, const float* log_z, const float* T_prime, float sigma,
                        const float* W0, const float* b0, const float* W1, const float* b1, const float* W2,
                        const float* b2, const float* Ws, const float* bs1, const float* bs2, int64_t N, int K, int S,
                        int L, int Dy, int U, float* dx, float* dparams, float* ll, float* scalars, float* g_log_z,
                        float* g_T_prime, float* r, void* ws, size_t ws_bytes, void* tail_ws, size_t tail_ws_bytes,
                        void* stream);

/* ------------------------------------------------------------------------------------------------
 * Scalar tail of the SVAE ELBO (models/svae.py:216-254 compute_elbo; vae.py:232-250) - two launches
 * ------------------------------------------------------------------------------------------------
 * From log_z (N,K), T_prime (N,K) (the theta side of the regulariser, as the fused E-step returns it) and the
 * per-sample reconstruction sums ll (N,K,S) of the decoder kernel:
 *   r = exp(log_z);  rec = -1/(2S) sum_nk r_nk sum_s ll_nks - N Dy/2 log(2 pi);  reg = sum_nk r_nk (T'_nk + log_z_nk)
 *   scalars = [elbo = rec - reg, rec, reg]   (fp64 sums, fixed order: deterministic)
 *   g_log_z = sigma * d elbo / d log_z,  g_T_prime = sigma * d elbo / d T'   (sigma = -1 for loss = -elbo)
 * Two launches: the (N,K) pass with per-block partial sums, then a one-wave sum of the partials in a fixed order.
 * ws: vmp_svae_elbo_tail_workspace_bytes() bytes of scratch (no initialisation needed); one per concurrently running stream. */
size_t vmp_svae_elbo_tail_workspace_bytes(void);
int    vmp_svae_elbo_tail(const float* log_z, const float* T_prime, const float* ll, int64_t N, int K, int S, int Dy,
                          float sigma, float* scalars, float* g_log_z, float* g_T_prime, float* r, void* ws,
                          size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The minibatch training step in 6 launches (round 6; experiments.py:196-267 at its own operating point, minibatches
 * of 64-100 rows, where a step is bound by the NUMBER of launches - 13 + the input copy + an eager scalar launch before)
 * ------------------------------------------------------------------------------------------------
 *   vmp_mlp_gauss_head_fwd_prep  encoder (vae.make_encoder, vae.py:131-135) + recognition unpacking + theta packing
 *                                (svae.py:342-358, 205-214) + the replayed step's scalars from a table
 *   vmp_svae_estep_fwd_rng_epi   E-step, sub-sample, r                    (svae.py:14-151)
 *   vmp_decoder_elbo_lazy        decoder value + gradients; parameter partials stay in ws
 *   vmp_svae_estep_bwd_tail      ELBO tail + E-step backward              (svae.py:216-254 and the autodiff of :14-119)
 *   vmp_mlp_gauss_head_bwd_lazy  encoder backward; parameter partials stay in ws
 *   vmp_svae_step_final          partial rows -> gradients of phi_gmm (autodiff of svae.py:342-358), both MLP partial
 *                                reductions, Adam on all 21 tensors, M-step moments + CVI update, ELBO scalars
 * Every value equals what the stand-alone launches produce (same device functions, same summation orders); the ELBO
 * scalars are summed per tile instead of per tail block (fp64: equal to fp32 rounding).                              */

/* The step's first launch: vmp_mlp_gauss_head_fwd (the encoder: x (R,L) -> out1, out2 (R,Dy)) and vmp_svae_prep_fwd2 with latent
 * size Dy (recognition unpacking + theta packing, K components) as ONE grid - they depend on the parameters and the minibatch only.
 * scalar_table != NULL (a step replayed from a HIP graph): (table_rows, 2) 64-bit words [Philox key | CVI step size (f32), Adam step
 * size (f32)] filled by the host for the coming steps; the launch copies row *counter (u64, device) to dst16 - the 16 bytes the
 * later launches read, as vmp_svae_step_scalars writes them - and increments the counter: no eager launch per replay.         */
int    vmp_mlp_gauss_head_fwd_prep(const float* x, const float* W0, const float* b0, const float* W1, const float* b1,
                                   const float* W2, const float* b2, const float* Ws, const float* bs1, const float* bs2,
                                   int64_t R, int L, int Dy, int U, float var_scale, float* out1, float* out2,
                                   const float* mu_k, const float* L_raw, const float* pi_raw, const float* alpha,
                                   const float* A, const float* b, const float* beta, const float* v_hat, int K, float* Lk,
                                   float* P, float* bias, float* m, float* W, float* kappa, double* logpi,
                                   const void* scalar_table, int table_rows, void* counter, void* dst16, void* stream);
/* vmp_svae_step_scalars (below) and the copy of the minibatch (n_floats fp32 words, y_src -> y_dst) in one launch. */
int    vmp_svae_step_inputs(void* dst16, uint64_t philox_key, float cvi_step, float adam_step, const float* y_src,
                            float* y_dst, int64_t n_floats, void* stream);
/* Number of partial rows the fused MLP backward kernel writes for `rows` input rows (N*K*S for the decoder). */
int    vmp_decoder_bwd_blocks(int64_t rows);
/* vmp_decoder_elbo without its second launch: dx, ll as there; the (vmp_decoder_bwd_blocks(N*K*S), param_words) fp32
 * parameter partials are left in ws for vmp_svae_step_final.  The scalar tail runs inside vmp_svae_estep_bwd_tail.  */
int    vmp_decoder_elbo_lazy(const float* x, const float* y, const float* log_z, float sigma, const float* W0,
                             const float* b0, const float* W1, const float* b1, const float* W2, const float* b2,
                             const float* Ws, const float* bs1, const float* bs2, int64_t N, int K, int S, int L, int Dy,
                             int U, float* dx, float* ll, void* ws, size_t ws_bytes, void* stream);
/* vmp_mlp_gauss_head_bwd without the partial reduction ((vmp_decoder_bwd_blocks(R), param_words) partials stay in ws). */
int    vmp_mlp_gauss_head_bwd_lazy(const float* x, const float* g_out1, const float* g_out2, float var_scale,
                                   const float* W0, const float* b0, const float* W1, const float* b1, const float* W2,
                                   const float* b2, const float* Ws, const float* bs1, const float* bs2, int64_t R, int L,
                                   int Dy, int U, float* dx, void* ws, size_t ws_bytes, void* stream);
/* 1 when vmp_svae_estep_bwd_tail covers the shape (the minibatch form: <= 256 tiles of 64 / K rows, S <= 16). */
int    vmp_svae_bwd_tail_applies(int64_t N, int K, int L, int S);
/* vmp_svae_elbo_tail's (N,K) pass + vmp_svae_estep_bwd_n in ONE launch, Gaussian theta: dLoss/dlog_z and dLoss/dT' are
 * formed per cell inside the kernel (sigma * d elbo, as the tail) and never reach memory.  Gx = sigma * d elbo / dx from
 * the decoder.  Outputs: g_eta1, g_eta2d, partials (one row per tile: vmp_svae_bwd_blocks_for) as vmp_svae_estep_bwd_n;
 * r (N,K) = exp(log_z); tail_part (tiles, 2) fp64 = per-tile [sum r A / 2S, sum r (T' + log z)].                     */
int    vmp_svae_estep_bwd_tail(const float* eta1, const float* eta2d, const float* hk, const float* Pk, const float* bias,
                               const float* mk, const float* Wk, const float* x, const float* lz, const float* T_prime,
                               const float* ll, float sigma, const float* Gx, int64_t N, int K, int L, int S,
                               float* g_eta1, float* g_eta2d, float* partials, size_t partial_bytes, float* r,
                               double* tail_part, size_t tail_bytes, void* stream);
/* vmp_svae_prep_fwd that also leaves log softmax(pi_raw) (K) fp64 in logpi (may be NULL): the backward forms below read it
 * instead of the other components' pi_raw. */
int    vmp_svae_prep_fwd2(const float* mu_k, const float* L_raw, const float* pi_raw, const float* alpha, const float* A,
                          const float* b, const float* beta, const float* v_hat, int K, int L, float* Lk, float* P,
                          float* bias, float* m, float* W, float* kappa, double* logpi, void* stream);
/* vmp_svae_bwd_reduce (phi side) + vmp_svae_phi_prep_bwd in one launch (stand-alone form of what vmp_svae_step_final does for
 * phi_gmm): block k sums component k's rows of `partials` (nblk, K, vmp_svae_bwd_partial_words) in the order of
 * vmp_svae_bwd_reduce and differentiates the recognition unpacking on them. */
int    vmp_svae_bwd_reduce_prep(const float* partials, int nblk, const float* mu_k, const float* L_raw, const float* pi_raw,
                                const double* logpi, int K, int L, float* g_mu, float* g_Lraw, float* g_piraw, void* stream);
/* The closing launch.  dec_* / enc_*: parameter partials of the two MLPs (rows: *_blocks; sizes: in, units, out = (L,U,Dy)
 * for the decoder, (Dy,U,L) for the encoder), their 9 parameter / Adam-m / Adam-v tensors and 9 gradient outputs, in the
 * order [W0,b0,W1,b1,W2,b2,Ws,bs1,bs2]; partials (nblk rows) of vmp_svae_estep_bwd_tail and logpi of vmp_svae_prep_fwd2;
 * phi_*: the three phi_gmm tensors (mu_k (K,L), L_k (K,L,L), log_pi_k (K)), gradient outputs phi_g; x_samples (N,L), r (N,K): the M-step's inputs (N <= 512); prior, theta, theta_star: 5 natural
 * NIW / Dirichlet tensors each [alpha, A, b, beta, v_hat] (theta updated in place, theta_star may be NULL);
 * rho / rho_dev, lr_t / lr_t_dev as vmp_svae_cvi_update / vmp_adam_step; stats_out (K, 2+L+L*L) fp64;
 * tail_part (tail_n, 2) from vmp_svae_estep_bwd_tail -> scalars [elbo, rec, reg].                                   */
int    vmp_svae_step_final(const float* dec_part, int dec_blocks, int dec_in, int dec_units, int dec_out,
                           float* const* dec_p, float* const* dec_m, float* const* dec_v, float* const* dec_g,
                           const float* enc_part, int enc_blocks, int enc_in, int enc_units, int enc_out,
                           float* const* enc_p, float* const* enc_m, float* const* enc_v, float* const* enc_g,
                           const float* partials, int nblk, const double* logpi,
                           float* const* phi_p, float* const* phi_g, float* const* phi_m, float* const* phi_v,
                           const float* x_samples, const float* r, int64_t N, const float* const* prior,
                           float* const* theta, float* const* theta_star, const float* rho_dev, float rho, int K, int L,
                           double* stats_out, const double* tail_part, int tail_n, int Dy, float* scalars, double beta1,
                           double beta2, double eps, double lr_t, const float* lr_t_dev, void* stream);

/* The closing launch of a DATA-PARALLEL minibatch step (one process per GPU; experiments.py:247-260, tf_utils.py:52-87): the block
 * roles of vmp_svae_step_final, but nothing is updated - this rank's M-step moments, its 21 gradients and its three scalars go as
 * doubles into xbuf = [moments (K, 2+L+L*L) | phi_gmm mu_k, L_k, log_pi_k | encoder net (9) | decoder net (9) | elbo, rec, reg],
 * the packed buffer the step's ONE all-reduce sums; vmp_svae_cvi_update and vmp_adam_step_packed follow it.  The fp32
 * gradients are also left in dec_g / enc_g / phi_g.  xbuf_doubles >= the layout's length.                              */
int    vmp_svae_step_pack(double* xbuf, size_t xbuf_doubles, const float* dec_part, int dec_blocks, int dec_in, int dec_units,
                          int dec_out, float* const* dec_p, float* const* dec_g, const float* enc_part, int enc_blocks,
                          int enc_in, int enc_units, int enc_out, float* const* enc_p, float* const* enc_g,
                          const float* partials, int nblk, const double* logpi, float* const* phi_p, float* const* phi_g,
                          const float* x_samples, const float* r, int64_t N, int K, int L, const double* tail_part, int tail_n,
                          int Dy, float* scalars, void* stream);

/* Writes the 16 bytes [Philox key (u64) | CVI step size (f32) | Adam step size (f32)] that a graph-captured training step
 * reads at run time (vmp_svae_estep_fwd_rng_dev / vmp_svae_subsample_rng seed_dev, vmp_svae_cvi_update rho_dev,
 * vmp_adam_step lr_t_dev = dst16 + 0 / 8 / 12): one launch, values passed by value.                                  */
int    vmp_svae_step_scalars(void* dst16, uint64_t philox_key, float cvi_step, float adam_step, void* stream);

/* tf.train.AdamOptimizer's update (TF 1.3; experiments.py:264-265) of n_tensors fp32 tensors in one launch per 32
 * tensors:  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr_t m / (sqrt(v) + eps), where the caller supplies the
 * bias-corrected step size lr_t = lr sqrt(1-b2^t)/(1-b1^t) - by value, or through lr_t_dev (a device float that
 * overrides it: graph-captured steps refresh that word instead of re-capturing).  params/grads/m/v: HOST arrays of
 * n_tensors device pointers; sizes: element counts.  The scalars are doubles: 1 - b is formed in fp64, then rounded.                                                                */
int    vmp_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* m, float* const* v,
                     const int64_t* sizes, double beta1, double beta2, double eps, double lr_t, const float* lr_t_dev,
                     void* stream);

/* Data-parallel training step (experiments.py:247-265; helpers/tf_utils.py:52-87 average_gradients): the tower gather
 * becomes ONE packed fp64 buffer per rank, summed by one all-reduce (vmp_pack_allreduce / torch.distributed).
 * vmp_pack_f64: dst = [src_0 | src_1 | ...] converted to fp64, one launch per 32 tensors (src: HOST array of device
 *   pointers; src_is_f64[t] != 0: tensor t already holds doubles).
 * vmp_adam_step_packed: vmp_adam_step whose gradient of tensor t is gscale * gbuf[goffsets[t] ...] (gscale = 1 / ranks:
 *   the mean over the towers, formed in fp64 and rounded once); grads_out (nullable, or NULL entries): the averaged fp32
 *   gradient is also stored there (what the trainer reports).                                                              */
int    vmp_pack_f64(int n_tensors, const void* const* src, const int* src_is_f64, const int64_t* sizes, double* dst, void* stream);
int    vmp_adam_step_packed(int n_tensors, float* const* params, const double* gbuf, const int64_t* goffsets, double gscale,
                            float* const* grads_out, float* const* m, float* const* v, const int64_t* sizes, double beta1,
                            double beta2, double eps, double lr_t, const float* lr_t_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stand-alone per-cell log-densities (forward only; the training step uses the fused kernels above)
 * ------------------------------------------------------------------------------------------------
 * vmp_gauss_logprob_nat_per_samp : gaussian.log_probability_nat_per_samp (distributions/gaussian.py:74-105)
 *     x (N,K,S,D), eta1 (N,K,D), eta2 (N,K,D,D) -> out (N,K,S)
 * vmp_gauss_logprob_nat          : gaussian.log_probability_nat (gaussian.py:30-71), normalised over k
 *     x (N,D), eta1 (N,K,D), eta2 (N,K,D,D), log_weights (K) or NULL -> out (N,K)
 * vmp_student_t_logprob          : student_t.log_probability_per_samp (distributions/student_t.py:7-39,59-61)
 *     y (N,K,S,D), mu (K,D), W (K,D,D) lower with W^T W = sigma^-1, cst (K) = lgamma((v+D)/2) - lgamma(v/2)
 *     - D/2 log(pi v) - 1/2 logdet sigma, nu (K) -> out (N,K,S)   (the K distinct scale matrices are factorised
 *     once on the host side instead of N*K*S times, student_t.py:26-36)                                        */
int    vmp_gauss_logprob_nat_per_samp(const float* x, const float* eta1, const float* eta2, int64_t N, int K, int S,
                                      int D, float* out, void* stream);
int    vmp_gauss_logprob_nat(const float* x, const float* eta1, const float* eta2, const float* log_weights,
                             int64_t N, int K, int D, float* out, void* stream);
/* Stand-alone expected Mahalanobis distance (gmm.compute_expct_mahalanobis_dist models/gmm.py:84-94,
 * compute_dev_missing_data :97-114, smm.expct_mahalanobis_dist models/smm.py:88-96):
 * out (N,K) = v_k (x_n - m_k)^T P_k (x_n - m_k) + D / beta_k, entries with miss_mask (N,D) != 0 dropped.          */
int    vmp_mix_mahalanobis(const float* x, const float* m, const float* P, const float* v, const float* beta,
                           const uint8_t* miss_mask, int64_t N, int D, int K, float* out, void* stream);
int    vmp_student_t_logprob(const float* y, const float* mu, const float* W, const float* cst, const float* nu,
                             int64_t N, int K, int S, int D, float* out, void* stream);
/* Adjoints of the two per-sample densities above - what TF's autodiff does through gaussian.py:74-105 and student_t.py:7-39
 * when the reference differentiates compute_elbo (svae.py:236-243, 291-300; experiments.py:232):
 *   vmp_gauss_logprob_nat_per_samp_bwd: g (N,K,S) upstream -> gx (N,K,S,D), geta1 (N,K,D), geta2 (N,K,D,D) (symmetric:
 *       sum_s g_s (x_s x_s^T - E[x x^T]), the exponential-family identity; eta2 is read symmetrised as in the forward);
 *   vmp_student_t_logprob_bwd: gy (N,K,S,D) and per-block partial sums over (n,s) of the gradients w.r.t. the K-sized
 *       (mu_k (D) | W_k lower packed (D(D+1)/2) | cst_k): partials (vmp_student_t_bwd_blocks(N,S), K, D + D(D+1)/2 + 1),
 *       summed over the first axis by the caller; nu is treated as a constant (the reference's DoF is not trainable,
 *       experiments.py:163-165).                                                                                       */
int    vmp_gauss_logprob_nat_per_samp_bwd(const float* x, const float* eta1, const float* eta2, const float* g, int64_t N,
                                          int K, int S, int D, float* gx, float* geta1, float* geta2, void* stream);
int    vmp_student_t_bwd_blocks(int64_t N, int S);
int    vmp_student_t_logprob_bwd(const float* y, const float* mu, const float* W, const float* nu, const float* g, int64_t N,
                                 int K, int S, int D, float* gy, float* partials, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Evaluation metrics: the (N,K,S,Dy)-sized part of losses.weighted_mse (losses.py:9-38) and
 * losses.diagonal_gaussian_logprob (losses.py:83-145)
 * ------------------------------------------------------------------------------------------------
 *   mse (N,K) = mean_s sum_d (y - mean)^2                                             (may be NULL)
 *   lse (N,K) = log 1/S sum_s exp( logw_nk(s) - 1/2 sum_d mask_nd [(y-mean)^2/var + log var + log 2pi] )   (may be NULL)
 * logw: (N,K), or (N,K,S) when logw_per_sample, or NULL; mask (N,Dy) uint8 or NULL (losses.py:118-124).
 * mask_mse != 0: the squared error also counts masked entries only,
 *   mse = mean_s sum_d mask_nd (y - mean)^2   = the per-cell part of losses.imputation_mse (losses.py:148-170).   */
int    vmp_eval_cell_metrics(const float* y, const float* mean, const float* var, const float* logw,
                             int logw_per_sample, const uint8_t* mask, int mask_mse, int64_t N, int K, int S, int Dy,
                             float* mse, float* lse, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU exchange (replaces the in-graph tower gather + mean of experiments.py:247-260 and
 * helpers/tf_utils.py:52-87): ONE in-place all-reduce(sum) over RCCL / xGMI of the packed fp64 buffer
 *   T1: raw moments (K, 2+D+D*D);   T3: [ raw moments | flat gradients | elbo, neg_rec_err, regulariser ]
 * after which every rank applies the identical K-sized update.  `comm` is an RCCL communicator (ncclComm_t); a host
 * that has none (the torch host uses torch.distributed's) builds one with the three helpers: rank 0 calls
 * vmp_comm_unique_id and ships the 128-byte id to the other ranks out of band, every rank calls vmp_comm_init_rank
 * with its device current.  RCCL is resolved at run time (the copy the process has already loaded, else librccl.so.1):
 * the library has no link-time dependency on it and single-GPU hosts never load it.                              */
#define VMP_COMM_ID_BYTES 128
int    vmp_comm_unique_id(void* id_out /* VMP_COMM_ID_BYTES */);
int    vmp_comm_init_rank(void** comm_out, int nranks, const void* id, int rank);
int    vmp_comm_destroy(void* comm);
int    vmp_pack_allreduce(void* comm, double* buf, size_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * T1 data-parallel iteration in ONE launch per rank (round 3).  The RCCL form above costs three dependent launches
 * plus the collective (local reduction -> all-reduce -> posterior); here the finalize kernel of every rank
 *   (1) reduces its own per-block partials (as vmp_mix_finalize_ws),
 *   (2) PUSHES its un-shifted fp64 moments of component k into slot [parity][rank][k] of EVERY rank's exchange buffer
 *       (plain stores, system-scope release, then a sequence word = iteration + 1),
 *   (3) waits until the sequence words of all ranks for component k have arrived in its OWN buffer (bounded spin on
 *       local memory), and sums the slots in rank order 0..G-1 - the same order on every rank, so all ranks obtain
 *       bit-identical moments -, then
 *   (4) continues with the posterior update and the E-step pack (as vmp_mix_finalize).
 * Replaces the tower gather + M-step on the parameter device of experiments.py:247-260 for the pure mixture loop
 * (gmm.py:258-263 over sharded rows).  Exchange buffers: one per rank, vmp_exch_bytes(G, K, D) bytes of uncached
 * device memory (vmp_exch_alloc), exported with vmp_exch_export (VMP_EXCH_HANDLE_BYTES, shipped to the peers out of band)
 * and mapped by every peer with vmp_exch_open; peers[g] = rank g's buffer as mapped in THIS process (peers[rank] = the
 * local allocation).  `iteration` must increase by one per call on every rank (slots are double-buffered by parity).
 * `status` (device int, may be NULL) is set to 1 if a wait timed out (~4 s): the results of that call are invalid.   */
#define VMP_EXCH_MAX_RANKS 16
#define VMP_EXCH_HANDLE_BYTES 64
size_t vmp_exch_bytes(int nranks, int K, int D);
int    vmp_exch_alloc(void** buf_out, size_t bytes);
int    vmp_exch_free(void* buf);
int    vmp_exch_export(void* buf, void* handle_out /* VMP_EXCH_HANDLE_BYTES */);
int    vmp_exch_open(const void* handle, void** peer_out);
int    vmp_exch_close(void* peer);
int    vmp_mix_finalize_exchange(const void* workspace, const float* pivot, int64_t N, int D, int K, int flavour,
                                 const float* alpha0, const float* beta0, const float* m0, const float* C0,
                                 const float* v0, const float* kappa, float* alpha, float* beta, float* m, float* C,
                                 float* v, float* xbar, float* S, float* pi, float* pack, double* stats_out,
                                 void* const* peers, int nranks, int rank, unsigned long long iteration, int* status,
                                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VMP_HIP_H */
