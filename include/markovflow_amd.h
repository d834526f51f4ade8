/*
 * markovflow_amd - C ABI of the MI355X (gfx950) Kalman / block-tridiagonal hot path.
 *
 * This is the drop-in boundary.  In the reference the arithmetic of this path is reached through
 * eight Python imports from the third-party TensorFlow-op library banded_matrices
 * (/root/reference/markovflow/block_tri_diag.py:22-31) plus batched TensorFlow small-matrix ops
 * (markovflow/state_space_model.py:431-483, markovflow/kalman_filter.py:86-271).  Each entry point
 * below names the reference call it replaces.
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE pointer, contiguous row-major, layout exactly the
 *     reference's: matrices [B, T, d, d], vectors [B, T, d]; `sub` has T-1 blocks per series and
 *     block k couples block k+1 (row) with block k (column);
 *   - suffix _f32 / _f64 selects the scalar type; both are first class;
 *   - nothing is allocated inside: scratch comes from the caller (`*_workspace_bytes` + `ws`);
 *   - `stream` is a hipStream_t (passed as void*); all work is enqueued, nothing synchronises;
 *   - `info` (nullable) is a device int, zeroed by the caller, that the kernels raise on a non-positive pivot (LAPACK info > 0
 *     style; results are then NaN from the failing block on).  The word NAMES the first failing block where the raising kernel
 *     knows it - mf_info_flat_index(word) = series * blocks_per_series + block, the smallest such index over everything that
 *     raised on this word (the Cholesky and log-likelihood kernels of the lane, streamed and panel forms) - and is 1 where it does
 *     not (reduction levels, composite kernels): LAPACK's `info = 1 + flat index`, SURVEY 8(b), is `1 + mf_info_flat_index(word)`;
 *   - return value: 0 ok; -k = argument k (1-based) invalid; -100 = state dimension not instantiated for this entry
 *     point (register kernels 1..9, row kernels 10..15, wave / tile engines up to 64 in fp32 and 32 in fp64 - see
 *     mf_max_state_dim*, mf_row_operators_cover); -101 = this fused / streamed variant does not cover the call (the caller
 *     takes the general entry point); -1000 = launch failure;
 *   - re-entrant, no global state.
 */
#ifndef MARKOVFLOW_AMD_H
#define MARKOVFLOW_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Library / build identification. */
int mf_version(void);
/* Queues a 4-byte copy of the device word `info` into `host_mirror` (pinned host memory) on `stream`: after any synchronisation
 * with the stream the mirror holds the final word of every factorising launch queued before (the reference raises inside the
 * Cholesky op, block_tri_diag.py:423-436; here the failure crosses the bus in stream order, never outside it). */
int mf_info_mirror(int* host_mirror, const int* info, void* stream);
/* Flat index (series * blocks_per_series + block) of the first block whose elimination met a non-positive pivot, decoded from an
 * `info` word; -1 when the word is 0 (no failure) or 1 (a failure whose block the raising kernel could not name). */
int64_t mf_info_flat_index(int info_word);
int mf_max_state_dim(void);              /* every entry point, fp32 and fp64: register-resident kernels (9)       */
/* 1 when the register / row kernels run EVERY operator for this shape: d <= 9 always; 10 <= d <= 15 (row kernels only: one
 * 16-lane row per chunk; many series or short chains run them with one chunk per series) for every chain of at least two
 * blocks - otherwise, and for d >= 16, the LDS-tile / MFMA engine takes the call (same entry points; the chain-layout and
 * fused variants then return -100 / -101).  With more than four outputs the observation side (log-likelihood, precision
 * assembly, gradient step) goes to the tile engine at any d. */
int mf_row_operators_cover(int64_t B, int64_t T, int d, int elem_size);
int mf_max_state_dim_f32_loglik(void);   /* mf_kf_loglik_f32 only: LDS-tiled MFMA kernels for 10 <= d <= 64       */
int mf_max_state_dim_f64_loglik(void);   /* mf_kf_loglik_f64 only: f64 MFMA for 10 <= d <= 64 (panel kernels > 32) */
int mf_max_state_dim_f64_tile_ops(void); /* every other fp64 entry point on the MFMA engines: 10 <= d <= 32          */

/*
 * KalmanFilter.log_likelihood, per series, fully fused
 *   replaces markovflow/kalman_filter.py:184-255 and everything it calls:
 *   _k_inv_post (:86-101), StateSpaceModel._build_precision (state_space_model.py:431-483),
 *   SymmetricBlockTriDiagonal.cholesky (block_tri_diag.py:423-436 -> banded cholesky_band),
 *   marginal_means (state_space_model.py:232-251), LowerTriangularBlockTriDiagonal.solve
 *   (block_tri_diag.py:339-351 -> banded solve_triang_mat), abs_log_det (:353-366),
 *   log_det_precision (state_space_model.py:343-373).
 * Inputs: mu0 [B,d], cholP0 [B,d,d], A [B,T-1,d,d], b [B,T-1,d], cholQ [B,T-1,d,d],
 *         H [B,T,m,d], y [B,T,m], Rinv [m,m] (rinv_per_step=0, KalmanFilter) or [B,T,m,m]
 *         (rinv_per_step=1, KalmanFilterWithSites / WithSparseSites), 1 <= m <= 32 (up to four outputs: the register / row
 *         kernels; more: the LDS-tile kernels, at any state dimension).
 * State dimension: 1..9 in fp32 and fp64: one lane per (series, time-chunk) with the state in registers up to d = 6 (fp32: 8),
 *         above that one 16-lane DPP row per (series, time-chunk) with one matrix row per lane (csrc/mf_row.hpp -
 *         BASELINE config 4, d = 9) - the rows also take 10 <= d <= 15, there with up to EIGHT outputs; beyond that (and with
 *         more outputs than the register / row kernels take, at any d) 1 <= d <= 64 (fp32) or 32 (fp64) with 1 <= m <= 32 run one workgroup per (series,
 *         time-chunk) on LDS tiles and f32 / f64 MFMA (csrc/mf_big.hpp - BASELINE config 5, state_dim = 64).
 * Output: out[s] = add_const + term1 + term2 + 1/2 log|K^-1| - log|L|  (kalman_filter.py:233-253), i.e. the
 *         per-series log-likelihood; the terms that do not depend on the chain,
 *         -1/2 m T log(2 pi) + 1/2 log|Sigma^-1|  (kalman_filter.py:229-231,249-253), are passed in add_const.
 * chunks: number of time partitions per series (0 = choose automatically).
 * prof_start / prof_stop: optional hipEvent_t (NULL = off) recorded on `stream` immediately before and after
 *         the dominant (level-0) kernel, so a caller can time that kernel alone.
 */
size_t mf_kf_loglik_workspace_bytes(int64_t B, int64_t T, int d, int elem_size, int64_t chunks);
int mf_kf_loglik_f64(int64_t B, int64_t T, int d, int m, const double* mu0, const double* cholP0,
                     const double* A, const double* b, const double* cholQ, const double* H, const double* y,
                     const double* Rinv, int rinv_per_step, double add_const, double* out, void* ws,
                     size_t ws_bytes, int* info, int64_t chunks, void* prof_start, void* prof_stop, void* stream);
int mf_kf_loglik_f32(int64_t B, int64_t T, int d, int m, const float* mu0, const float* cholP0,
                     const float* A, const float* b, const float* cholQ, const float* H, const float* y,
                     const float* Rinv, int rinv_per_step, float add_const, float* out, void* ws,
                     size_t ws_bytes, int* info, int64_t chunks, void* prof_start, void* prof_stop, void* stream);

/*
 * SymmetricBlockTriDiagonal.cholesky  (block_tri_diag.py:423-436; banded cholesky_band +
 * block_to_band/band_to_block layout shuffles, which do not exist here).  Natural order:
 * L_0 = chol(D_0), W_{k-1} = S_{k-1} L_{k-1}^-T, L_k = chol(D_k - W_{k-1} W_{k-1}^T).
 * sub / lsub may be NULL (block-diagonal matrix).  Only the lower triangle of diag is read.
 * With many series one lane walks each chain.  With few, long chains (BASELINE config 3: B=1, T=100000) the
 * factorisation is parallelised in time by a multi-level partitioned elimination that still returns the
 * natural-order factor (csrc/mf_btd_par.hpp); that path needs scratch: ws_bytes >=
 * mf_btd_cholesky_workspace_bytes(...) (0 = the serial kernel will be used; ws may then be NULL).
 * The same holds for 10 <= d <= 64 (f32) / 32 (f64) on the LDS-tile / MFMA engine: few series are cut into chunks in time
 * (csrc/mf_bigpar_impl.hpp) - also for mf_btd_solve, mf_btd_diag_of_inverse, mf_btd_udl and mf_ssm_marginal_means, each with the
 * workspace its own query names; a NULL or short workspace selects the one-workgroup-per-series kernels, never an error.
 */
size_t mf_btd_cholesky_workspace_bytes(int64_t B, int64_t T, int d, int elem_size);
int mf_btd_cholesky_f64(int64_t B, int64_t T, int d, const double* diag, const double* sub, double* ldiag,
                        double* lsub, void* ws, size_t ws_bytes, int* info, void* stream);
int mf_btd_cholesky_f32(int64_t B, int64_t T, int d, const float* diag, const float* sub, float* ldiag,
                        float* lsub, void* ws, size_t ws_bytes, int* info, void* stream);

/*
 * LowerTriangularBlockTriDiagonal.solve  (block_tri_diag.py:339-351; banded solve_triang_mat):
 * out = L^-1 rhs (transpose=0) or L^-T rhs (transpose=1).  rhs/out are [Br,T,d]; rhs series r uses the
 * factor of series r % Bl (Br a multiple of Bl: extra leading dims, state_space_model.py:307-322).
 * Few, long chains are solved in parallel in time (the substitution is an affine recursion, composed per chunk
 * and scanned over the chunks); scratch as for the Cholesky: mf_btd_solve_workspace_bytes (0 = serial kernel).
 */
size_t mf_btd_solve_workspace_bytes(int64_t Bl, int64_t Br, int64_t T, int d, int elem_size);
int mf_btd_solve_f64(int64_t Bl, int64_t Br, int64_t T, int d, const double* ldiag, const double* lsub,
                     const double* rhs, double* out, int transpose, void* ws, size_t ws_bytes, void* stream);
int mf_btd_solve_f32(int64_t Bl, int64_t Br, int64_t T, int d, const float* ldiag, const float* lsub,
                     const float* rhs, float* out, int transpose, void* ws, size_t ws_bytes, void* stream);

/*
 * BlockTriDiagonal.dense_mult  (block_tri_diag.py:175-199; banded product_band_mat).
 * mode 0: M x for lower-triangular M; 1: M^T x; 2: symmetric M x (lower triangle mirrored).
 */
int mf_btd_matvec_f64(int64_t Bl, int64_t Br, int64_t T, int d, const double* diag, const double* sub,
                      const double* x, double* out, int mode, void* stream);
int mf_btd_matvec_f32(int64_t Bl, int64_t Br, int64_t T, int d, const float* diag, const float* sub,
                      const float* x, float* out, int mode, void* stream);

/* LowerTriangularBlockTriDiagonal.abs_log_det  (block_tri_diag.py:353-366): out [B]. */
int mf_btd_logdet_f64(int64_t B, int64_t T, int d, const double* ldiag, double* out, void* stream);
int mf_btd_logdet_f32(int64_t B, int64_t T, int d, const float* ldiag, float* out, void* stream);

/*
 * Fused scalar form of cholesky + solve + abs_log_det on an explicit (diag, sub, rhs):
 * out[s] = 1/2 |L^-1 rhs|^2 - log|L|, computed by partitioned (block cyclic reduction style)
 * elimination in time; this is what the log-likelihood of the sites variants reduces to.
 */
size_t mf_btd_logdet_quad_workspace_bytes(int64_t B, int64_t T, int d, int elem_size);
int mf_btd_logdet_quad_f64(int64_t B, int64_t T, int d, const double* diag, const double* sub, const double* rhs,
                           double* out, void* ws, size_t ws_bytes, int* info, void* stream);
int mf_btd_logdet_quad_f32(int64_t B, int64_t T, int d, const float* diag, const float* sub, const float* rhs,
                           float* out, void* ws, size_t ws_bytes, int* info, void* stream);

/*
 * LowerTriangularBlockTriDiagonal.block_diagonal_of_inverse  (block_tri_diag.py:318-337; banded
 * inverse_from_cholesky_band): diagonal blocks of (L L^T)^-1 into odiag [B,T,d,d]; if osub != NULL also
 * the sub-diagonal blocks [B,T-1,d,d] (what ssm_gaussian_transformations.py:453-458 reads).  The backward
 * (Takahashi) recursion is a congruence recursion Sigma_k = N_k + G_k^T Sigma_{k+1} G_k; for few series it is
 * composed per time-chunk and scanned (csrc/mf_btd_par.hpp), with scratch from the caller.
 */
size_t mf_btd_diag_of_inverse_workspace_bytes(int64_t B, int64_t T, int d, int elem_size);   /* 0 = serial kernel */
int mf_btd_diag_of_inverse_f64(int64_t B, int64_t T, int d, const double* ldiag, const double* lsub,
                               double* odiag, double* osub, void* ws, size_t ws_bytes, void* stream);
int mf_btd_diag_of_inverse_f32(int64_t B, int64_t T, int d, const float* ldiag, const float* lsub, float* odiag,
                               float* osub, void* ws, size_t ws_bytes, void* stream);

/*
 * StateSpaceModel.marginal_covariances and subsequent_covariances (markovflow/state_space_model.py:254-275, 326-341) in one
 * scan: Sigma_0 = P0, Sigma_{k+1} = A_k Sigma_k A_k^T + Q_k, out_sub[k] = Cov(x_{k+1}, x_k) = A_k Sigma_k (nullable).  The
 * reference takes the block diagonal of the inverse of the assembled precision; the forward recursion yields the same blocks
 * without assembling or factorising it.  Parallel in time for B < 4096 (workspace = the diag-of-inverse query), one lane per
 * series otherwise.  T >= 2.  State dimension 1..9 in registers; 10..64 (fp32) / 10..32 (fp64) on the LDS-tile / MFMA engine,
 * partitioned in time: chunk maps S -> M S M^T + N, a walk over the chunk boundaries, re-start per chunk (three launches;
 * workspace from the same query, one workgroup per series without it); -100 above.
 * cholP0 [B,d,d], A, cholQ [B,T-1,d,d]; out_cov [B,T,d,d], out_sub [B,T-1,d,d].
 */
int mf_ssm_marginal_covariances_f64(int64_t B, int64_t T, int d, const double* cholP0, const double* A, const double* cholQ,
                                    double* out_cov, double* out_sub, void* ws, size_t ws_bytes, void* stream);
int mf_ssm_marginal_covariances_f32(int64_t B, int64_t T, int d, const float* cholP0, const float* A, const float* cholQ,
                                    float* out_cov, float* out_sub, void* ws, size_t ws_bytes, void* stream);

/*
 * GaussMarkovDistribution.marginals (+ StateSpaceModel.subsequent_covariances)  (gauss_markov.py:107-117,
 * state_space_model.py:232-262,326-341) with ONE kernel writing all three: mu_{k+1} = A_k mu_k + b_k rides along the covariance
 * recursion of mf_ssm_marginal_covariances, A is read once.  Many series (B >= 4096) or a short chain (T < 64): one sweep per
 * series, no workspace.  Few long chains: the up / down sweeps of the two scans in time, then one emit kernel that restarts both
 * recursions at the chunk boundaries (workspace: mf_ssm_marginals_workspace_bytes; -101 without it - the caller then uses
 * mf_ssm_marginal_means + mf_ssm_marginal_covariances).  State dimension 10..64 (fp32) / 10..32 (fp64): the means ride along
 * the three passes of the time-partitioned covariance recursion (same workspace query).
 * mu0 [B,d], b [B,T-1,d]; out_mean [B,T,d], out_cov [B,T,d,d], out_sub [B,T-1,d,d] (nullable).
 */
size_t mf_ssm_marginals_workspace_bytes(int64_t B, int64_t T, int d, int elem_size);
int mf_ssm_marginals_f64(int64_t B, int64_t T, int d, const double* mu0, const double* cholP0, const double* A, const double* b,
                         const double* cholQ, double* out_mean, double* out_cov, double* out_sub, void* ws, size_t ws_bytes,
                         void* stream);
int mf_ssm_marginals_f32(int64_t B, int64_t T, int d, const float* mu0, const float* cholP0, const float* A, const float* b,
                         const float* cholQ, float* out_mean, float* out_cov, float* out_sub, void* ws, size_t ws_bytes,
                         void* stream);

/*
 * SymmetricBlockTriDiagonal.upper_diagonal_lower  (block_tri_diag.py:438-545, the tf.while_loop):
 * ut [B,T-1,d,d] = U_k^T, chol_d [B,T,d,d] = chol(Delta_k).
 * With eta != NULL ([B,T,d]) it additionally runs the rest of
 * BaseKalmanFilter.posterior_state_space_model (kalman_filter.py:159-174) in the same sweep:
 * m_post [B,T,d] = [mu0', b'_1...] and chol_dinv [B,T,d,d] = chol(Delta_k^-1) = [cholP0', cholQ'_1...].
 * chain_layout = 1 (needs eta; state dimension <= 9, else -101) writes the posterior chain the way StateSpaceModel takes it,
 * so that nothing has to be sliced, copied or negated afterwards: ut holds the posterior transitions A'_k = -(U_k^T), m_post
 * is [B*d values of mu0' | B*(T-1)*d values of b'] and chol_dinv [B*d*d values of cholP0' | B*(T-1)*d*d values of cholQ'];
 * chol_d may be NULL in this mode (the chain does not contain it: one d x d block per step less to write).
 * Few series: Delta_k are the natural-order pivots of the block-REVERSED matrix, so the parallel-in-time Cholesky
 * hierarchy is reused with reversed indexing, and the posterior offsets are an affine scan (scratch from the caller,
 * mf_btd_udl_workspace_bytes; 0 / NULL = one lane per series).
 */
size_t mf_btd_udl_workspace_bytes(int64_t B, int64_t T, int d, int elem_size);
int mf_btd_udl_f64(int64_t B, int64_t T, int d, const double* diag, const double* sub, double* ut, double* chol_d,
                   const double* eta, double* m_post, double* chol_dinv, int chain_layout, void* ws, size_t ws_bytes,
                   int* info, void* stream);
int mf_btd_udl_f32(int64_t B, int64_t T, int d, const float* diag, const float* sub, float* ut, float* chol_d,
                   const float* eta, float* m_post, float* chol_dinv, int chain_layout, void* ws, size_t ws_bytes,
                   int* info, void* stream);

/*
 * StateSpaceModel._build_precision (state_space_model.py:431-483), optionally + H^T R^-1 H
 * (kalman_filter.py:86-101) and the posterior information vector
 * eta = G^T Sigma^-1 y + K^-1 mu (kalman_filter.py:153-156).
 * H == NULL: prior precision only.  y == NULL: no observation term in eta.  eta == NULL: not computed
 * (mu0, b may then be NULL).  diag [B,T,d,d], sub [B,T-1,d,d], eta [B,T,d].
 */
int mf_ssm_precision_f64(int64_t B, int64_t T, int d, int m, const double* mu0, const double* cholP0,
                         const double* A, const double* b, const double* cholQ, const double* H, const double* y,
                         const double* Rinv, int rinv_per_step, double* diag, double* sub, double* eta,
                         void* stream);
int mf_ssm_precision_f32(int64_t B, int64_t T, int d, int m, const float* mu0, const float* cholP0, const float* A,
                         const float* b, const float* cholQ, const float* H, const float* y, const float* Rinv,
                         int rinv_per_step, float* diag, float* sub, float* eta, void* stream);

/*
 * StateSpaceModel.marginal_means / sample  (state_space_model.py:232-251,298-324): solves
 * (A^-1 block) x = offs, i.e. x_0 = offs_0, x_k = A_k x_{k-1} + offs_k.  offs/out [Br,T,d]; series r uses
 * the transitions of series r % Bl.  Scratch (parallel-in-time scan for few series): the size
 * mf_btd_solve_workspace_bytes(Bl, Br, T, d, elem_size) returns (0 / NULL = serial kernel).
 */
int mf_ssm_marginal_means_f64(int64_t Bl, int64_t Br, int64_t T, int d, const double* A, const double* offs,
                              double* out, void* ws, size_t ws_bytes, void* stream);
int mf_ssm_marginal_means_f32(int64_t Bl, int64_t Br, int64_t T, int d, const float* A, const float* offs,
                              float* out, void* ws, size_t ws_bytes, void* stream);

/*
 * Block-wise d x d products out[s,k] = X[s,k] Y[s,k] over [B, n] blocks; x_stride / y_stride = blocks per series in
 * the allocations of X / Y (>= n: a leading slice of a longer chain needs no copy).  Replaces the batched tf.matmul of
 * StateSpaceModel.subsequent_covariances (state_space_model.py:326-341), Cov(x_{k+1}, x_k) = A_k P_k.
 */
int mf_block_matmul_f64(int64_t B, int64_t n, int d, const double* X, int64_t x_stride, const double* Y,
                        int64_t y_stride, double* out, void* stream);
int mf_block_matmul_f32(int64_t B, int64_t n, int d, const float* X, int64_t x_stride, const float* Y,
                        int64_t y_stride, float* out, void* stream);

/*
 * SDE kernel -> state space model tensors on the device (the step right before the path).
 * For a concatenation (block-diagonal state: Sum / IndependentMultiOutput / a single kernel) of `ncomp` Matern components
 * of order orders[c] in {1, 3, 5} (Matern-1/2, 3/2, 5/2; state sizes 1, 2, 3) with lam = sqrt(order) / lengthscale and
 * variance var:  A[s,k] = exp(F dt[s,k]) in closed form (markovflow/kernels/matern.py:66-86,299-324,434-460, block diagonal
 * as kernels/sde_kernel.py:592-610), Q = Pinf - A Pinf A^T + jitter I (sde_kernel.py:421-446) and cholQ = its lower Cholesky
 * factor with an exactly zero Q passed through as zero (state_space_model.py:634-656).
 * orders is a HOST array of ncomp ints; lam / var are device arrays [ncomp] (per_series = 0) or [B, ncomp] (per_series = 1);
 * dt [B, n]; outputs A, cholQ (nullable), Q (nullable): [B, n, d, d] with d = sum of the component sizes.
 */
int mf_sde_matern_transitions_f64(int64_t B, int64_t n, int ncomp, const int* orders, const double* lam, const double* var,
                                  int per_series, const double* dt, double jitter, double* A, double* cholQ, double* Q,
                                  void* stream);
int mf_sde_matern_transitions_f32(int64_t B, int64_t n, int ncomp, const int* orders, const float* lam, const float* var,
                                  int per_series, const float* dt, float jitter, float* A, float* cholQ, float* Q,
                                  void* stream);

/*
 * Reverse mode of mf_sde_matern_transitions_* with respect to the hyper-parameters (the reference differentiates matern.py /
 * sde_kernel.py:421-446 and the Cholesky through TensorFlow: the GPR training step): for incoming gradients g_A, g_cholQ
 * [B,n,d,d] (either may be NULL) writes out[b, k, c, 0 / 1] = d/d lam_c, d/d var_c of <g_A[b,k], A_k> + <g_cholQ[b,k], chol Q_k>;
 * the caller sums over k (and over b for shared hyper-parameters).  One lane per (series, transition), the generator's closed
 * forms evaluated in forward mode (csrc/mf_sde.hip: Dual2).
 */
int mf_sde_matern_transitions_grad_f64(int64_t B, int64_t n, int ncomp, const int* orders, const double* lam, const double* var,
                                       int per_series, const double* dt, double jitter, const double* g_A, const double* g_cholQ,
                                       double* out, void* stream);
int mf_sde_matern_transitions_grad_f32(int64_t B, int64_t n, int ncomp, const int* orders, const float* lam, const float* var,
                                       int per_series, const float* dt, float jitter, const float* g_A, const float* g_cholQ,
                                       float* out, void* stream);
/* The same with the incoming gradients as packed records (see mf_gpr_matern_loglik_grad): only the entries the generator reads. */
int mf_sde_matern_transitions_grad_packed_f64(int64_t B, int64_t n, int ncomp, const int* orders, const double* lam,
                                              const double* var, int per_series, const double* dt, double jitter,
                                              const double* g_packed, double* out, void* stream);
int mf_sde_matern_transitions_grad_packed_f32(int64_t B, int64_t n, int ncomp, const int* orders, const float* lam,
                                              const float* var, int per_series, const float* dt, float jitter,
                                              const float* g_packed, float* out, void* stream);
/* The stationary prior of the same kernels, chol(Pinf + jitter) (markovflow/kernels/sde_kernel.py:402-419): out [B,ncomp,2] =
 * the contraction of g_cholP0 [B,d,d] (only the lower triangles of its diagonal blocks are read) with d chol / d (lam, var) of
 * every component - per series; a caller with shared hyper-parameters (per_series = 0) sums over the batch. */
int mf_sde_matern_prior_chol_grad_f64(int64_t B, int ncomp, const int* orders, const double* lam, const double* var, int per_series,
                                      double jitter, const double* g_cholP0, double* out, void* stream);
int mf_sde_matern_prior_chol_grad_f32(int64_t B, int ncomp, const int* orders, const float* lam, const float* var, int per_series,
                                      float jitter, const float* g_cholP0, float* out, void* stream);

/*
 * GaussianProcessRegression.log_likelihood (markovflow/models/gaussian_process_regression.py:150-160) for a Matern kernel or
 * a Sum of two, with the kernel -> state-space-model step FUSED into the Kalman sweep: A_k and chol(Q_k) are generated in
 * registers from dt_k = t_{k+1} - t_k, so a step reads 16 bytes (t, y) instead of the materialised tensors.  One output,
 * zero mean function, H = [1 0 .. | 1 0 ..].  Arguments as for mf_sde_matern_transitions_* - orders: HOST array, ncomp <= 2 - plus
 * time points t [B,T] (strictly increasing), observations y [B,T], rinv [1] (device: 1 / noise variance); out[s] as
 * mf_kf_loglik_*: add_const carries -T/2 log 2 pi + T/2 log rinv.  Workspace: the size mf_kf_loglik_workspace_bytes returns.
 * Returns -101 when the component signature is not instantiated - supported: (1), (3), (5), (3,3), (5,3), (3,5), (5,5) on the
 * register kernels (d <= 6) and ANY concatenation of up to 15 components with 7 <= d <= 15 on the row kernels
 * (csrc/mf_row_gpr.hpp: every lane generates its own row of chol Q_k and column of A_k) - the caller then materialises the
 * model (mf_sde_matern_transitions + mf_kf_loglik).
 *
 * mf_gpr_matern_multi_loglik_*: the same for IndependentMultiOutput (kernels/sde_kernel.py:826-880: one output per component,
 * H[o] = e_{first state of component o}; BASELINE config 4 = three Matern-5/2 components, three outputs): y [B,T,m] with
 * m = ncomp <= 4 (<= 8 for d >= 10), rinv [m,m]; 7 <= d <= 15 (row kernels), -101 otherwise.
 */
int mf_gpr_matern_loglik_f64(int64_t B, int64_t T, int ncomp, const int* orders, const double* lam, const double* var,
                             int per_series, const double* t, const double* y, const double* rinv, double jitter,
                             double add_const, double* out, void* ws, size_t ws_bytes, int* info, int64_t chunks,
                             void* prof_start, void* prof_stop, void* stream);
int mf_gpr_matern_loglik_f32(int64_t B, int64_t T, int ncomp, const int* orders, const float* lam, const float* var,
                             int per_series, const float* t, const float* y, const float* rinv, float jitter,
                             float add_const, float* out, void* ws, size_t ws_bytes, int* info, int64_t chunks,
                             void* prof_start, void* prof_stop, void* stream);
int mf_gpr_matern_multi_loglik_f64(int64_t B, int64_t T, int ncomp, const int* orders, const double* lam, const double* var,
                                   int per_series, const double* t, const double* y, int m, const double* rinv, double jitter,
                                   double add_const, double* out, void* ws, size_t ws_bytes, int* info, int64_t chunks,
                                   void* prof_start, void* prof_stop, void* stream);
int mf_gpr_matern_multi_loglik_f32(int64_t B, int64_t T, int ncomp, const int* orders, const float* lam, const float* var,
                                   int per_series, const float* t, const float* y, int m, const float* rinv, float jitter,
                                   float add_const, float* out, void* ws, size_t ws_bytes, int* info, int64_t chunks,
                                   void* prof_start, void* prof_stop, void* stream);

/*
 * Last line of BaseKalmanFilter.log_likelihood (markovflow/kalman_filter.py:229-231,249-255) as ONE kernel:
 *   out[0] = sum_s per_series[s] + B * ( host_const - num_points * sum_i log chol_obs[i][i] + extra_const[0] )
 * per_series [B]: the output of the per-series entry point above; chol_obs [m,m] (nullable): Cholesky factor of the shared
 * observation covariance, contributing 1/2 T log|R^-1|; extra_const (nullable): one device scalar (e.g. the summed
 * log-determinants of per-step site precisions); host_const: -1/2 m T log(2 pi).  Accumulates in double.
 */
int mf_kf_loglik_total_f64(int64_t B, const double* per_series, int m, const double* chol_obs, int64_t num_points,
                           const double* extra_const, double host_const, double* out, void* stream);
int mf_kf_loglik_total_f32(int64_t B, const float* per_series, int m, const float* chol_obs, int64_t num_points,
                           const float* extra_const, float host_const, float* out, void* stream);

/*
 * Posterior prediction of the state at new time points: conditional_predict of
 * markovflow/conditionals.py:29-83 with _conditional_statistics_from_transitions (:122-203) and base_conditional_predict
 * (:380-420) fused, one lane per (series, new point).  idx [B,Np] (int64, device): insertion index of each new point
 * among the N training points (0..N); A_mt, Q_mt [B,Np,d,d]: transition from the previous training point (or from
 * "minus infinity") to the new point; A_tp, Q_tp: from the new point to the next training point (mf_sde_matern_transitions);
 * means [B,N,d], covs [B,N,d,d], subsequent_covs [B,N-1,d,d] = Cov(x_{k+1}, x_k): the posterior's marginals; prior_mean [B,d],
 * prior_cov [B,d,d]: the stationary state, used beyond both ends (pairwise_marginals, conditionals.py:424-485).
 * out_mean [B,Np,d], out_cov [B,Np,d,d] (NULL: means only).  State dimension 1..9.
 */
int mf_sde_conditional_predict_f64(int64_t B, int64_t N, int64_t Np, int d, const int64_t* idx, const double* A_mt,
                                   const double* Q_mt, const double* A_tp, const double* Q_tp, const double* means,
                                   const double* covs, const double* subsequent_covs, const double* prior_mean,
                                   const double* prior_cov, double* out_mean, double* out_cov, int* info, void* stream);
int mf_sde_conditional_predict_f32(int64_t B, int64_t N, int64_t Np, int d, const int64_t* idx, const float* A_mt,
                                   const float* Q_mt, const float* A_tp, const float* Q_tp, const float* means,
                                   const float* covs, const float* subsequent_covs, const float* prior_mean,
                                   const float* prior_cov, float* out_mean, float* out_cov, int* info, void* stream);

/*
 * conditional_statistics of markovflow/conditionals.py:87-120 (_conditional_statistics_from_transitions, :122-203): for every
 * new time point the statistics of p(x_t | x_-, x_+) = N(P_t [x_-, x_+], T_t) from the transitions x_- -> x_t (A_mt, Q_mt) and
 * x_t -> x_+ (A_tp, Q_tp), each [n, d, d]:  E = Q_mt A_tp^T (Q_tp + A_tp Q_mt A_tp^T)^-1,  D = A_mt - E A_tp A_mt,
 * projections = [D | E] ([n, d, 2d]),  covariances = T = Q_mt - Q_mt A_tp^T (...)^-1 A_tp Q_mt ([n, d, d]).  One lane per point,
 * d <= 9 (-100 beyond).  `info`: raised when Q_tp + A_tp Q_mt A_tp^T is not positive definite.
 */
int mf_sde_conditional_statistics_f64(int64_t n, int d, const double* A_mt, const double* Q_mt, const double* A_tp,
                                      const double* Q_tp, double* projections, double* covariances, int* info, void* stream);
int mf_sde_conditional_statistics_f32(int64_t n, int d, const float* A_mt, const float* Q_mt, const float* A_tp, const float* Q_tp,
                                      float* projections, float* covariances, int* info, void* stream);

/*
 * Reverse mode through the operators: banded_matrices registers a gradient for cholesky_band and inverse_from_cholesky_band
 * (markovflow/block_tri_diag.py:22-31), which is how the CVI models differentiate dist_p.precision -> naturals_to_ssm_params
 * (models/variational_cvi.py:105-136).
 *   mf_btd_cholesky_grad: (g_ldiag [B,T,d,d] lower | NULL, g_lsub [B,T-1,d,d] | NULL) = gradients w.r.t. the factor's blocks ->
 *     (g_diag [B,T,d,d] symmetric, g_sub [B,T-1,d,d]) = gradients w.r.t. the symmetric matrix' blocks.  Local adjoint of the
 *     dense Cholesky per block + the congruence recursion Z_k = C_k + G_k^T Z_{k+1} G_k, G_k = W_k L_k^-1 (parallel in time for
 *     few series) + an axpy.
 *   mf_btd_diag_of_inverse_grad: sigma = the forward's diagonal blocks of (L L^T)^-1; (g_diag | NULL, g_sub | NULL) = gradients
 *     w.r.t. the diagonal / sub-diagonal blocks of the inverse -> (g_ldiag lower, g_lsub).  The block Takahashi recursion run
 *     forward in reverse mode, A_{k+1} = Qbar_{k+1} + G_k A_k G_k^T, between two local kernels.
 * lsub == NULL: block-diagonal factor.  Workspace: mf_btd_grad_workspace_bytes (0: state dimension without these kernels,
 * the entry points then return -100).  d <= 9 (register kernels) and 10 <= d <= 15 where the row scan takes the recursion;
 * 10 <= d <= 32 otherwise (round 6, csrc/mf_adj.hip, 16 x 16 MFMA register tiles; the reference differentiates these operators at
 * d = 30, T = 1001): with a workspace of mf_btd_grad_workspace_bytes and at least 32 blocks, terms local in time (a wavefront per
 * block) around one congruence recursion per adjoint, partitioned in time; otherwise (shorter chains, ws == NULL) one wavefront
 * per series walks the block recurrences.
 */
size_t mf_btd_grad_workspace_bytes(int64_t B, int64_t T, int d, int elem_size);
int mf_btd_cholesky_grad_f64(int64_t B, int64_t T, int d, const double* ldiag, const double* lsub, const double* g_ldiag,
                             const double* g_lsub, double* g_diag, double* g_sub, void* ws, size_t ws_bytes, void* stream);
int mf_btd_cholesky_grad_f32(int64_t B, int64_t T, int d, const float* ldiag, const float* lsub, const float* g_ldiag,
                             const float* g_lsub, float* g_diag, float* g_sub, void* ws, size_t ws_bytes, void* stream);
int mf_btd_diag_of_inverse_grad_f64(int64_t B, int64_t T, int d, const double* ldiag, const double* lsub, const double* sigma,
                                    const double* g_diag, const double* g_sub, double* g_ldiag, double* g_lsub, void* ws,
                                    size_t ws_bytes, void* stream);
int mf_btd_diag_of_inverse_grad_f32(int64_t B, int64_t T, int d, const float* ldiag, const float* lsub, const float* sigma,
                                    const float* g_diag, const float* g_sub, float* g_ldiag, float* g_lsub, void* ws,
                                    size_t ws_bytes, void* stream);

/*
 * Gradient of the Kalman log-likelihood with respect to every tensor of the model (the reference
 * gets it from TensorFlow's reverse mode over the banded ops: tests/integration/models/test_gaussian_process_regression.py:117-130,
 * markovflow/models/variational_cvi.py:138-161 for the sites variants).  Fisher's identity,
 * grad log p(y) = E_{x|y}[grad log p(x, y)], evaluated from the smoothed marginals the caller passes in:
 * post_mean [B,T,d], post_cov [B,T,d,d], post_cross [B,T-1,d,d] = Cov(x_{k+1}, x_k) - exact, local in time, one lane per
 * (series, time point).  Outputs, all per series (not summed over the batch): g_mu0 [B,d], g_cholP0 [B,d,d] (lower),
 * g_A [B,T-1,d,d], g_b [B,T-1,d], g_cholQ [B,T-1,d,d] (lower), g_H [B,T,m,d], g_y [B,T,m], and g_omega [B,T,m,m] =
 * E[r r^T] with r = y - H x: the derivative of the log-likelihood with respect to the observation PRECISION of time point k
 * is -1/2 g_omega[k] (+ 1/2 R_k from the log-determinant, which the caller owns).  Rinv: shared [m,m] (rinv_per_step = 0) or
 * [B,T,m,m] (1: KalmanFilterWithSites / WithSparseSites).  H = NULL: no emission model - the expected score of the bare chain
 * under the given moments (used for the q2 half of the KL gradient); y, Rinv, g_H, g_y, g_omega are then ignored.
 * weights [B] (nullable): the incoming gradient of every series' value, applied to all outputs (required for d > 9 or m > 4).
 * State dimension 1..9 with m <= 4: register kernels; 10..15 with m <= 4: row kernels; otherwise (d <= 64 fp32 / 32 fp64, m <= d
 * rounded up to 16) one LDS-tile workgroup per (series, time point) (csrc/mf_biggrad_impl.hpp).
 */
int mf_kf_loglik_grad_f64(int64_t B, int64_t T, int d, int m, const double* mu0, const double* cholP0, const double* A,
                          const double* b, const double* cholQ, const double* H, const double* y, const double* Rinv,
                          int rinv_per_step, const double* post_mean, const double* post_cov, const double* post_cross,
                          double* g_mu0, double* g_cholP0, double* g_A, double* g_b, double* g_cholQ, double* g_H,
                          double* g_y, double* g_omega, const double* weights, int* info, void* stream);
int mf_kf_loglik_grad_f32(int64_t B, int64_t T, int d, int m, const float* mu0, const float* cholP0, const float* A,
                          const float* b, const float* cholQ, const float* H, const float* y, const float* Rinv,
                          int rinv_per_step, const float* post_mean, const float* post_cov, const float* post_cross,
                          float* g_mu0, float* g_cholP0, float* g_A, float* g_b, float* g_cholQ, float* g_H, float* g_y,
                          float* g_omega, const float* weights, int* info, void* stream);

/*
 * The same gradients for FEW, LONG series WITHOUT the smoothed marginals in memory (csrc/mf_grad_lds.hpp): five streamed passes,
 * all partitioned in time - the three of the streamed posterior chain above: chunk summaries, scan, emit (only chol(Q') and b' of
 * the chain are read back), a scan that meets the two sides of every chunk boundary in the smoothed marginal of the chunk's first block,
 * and a forward pass per (series, chunk) that carries (m_k, S_k, Cov(x_{k+1}, x_k)) in registers and writes every gradient once.
 * Replaces mf_kf_posterior_chain -> mf_ssm_marginal_means / _covariances -> mf_kf_loglik_grad for state dimensions 1..6,
 * m <= 3 (per-step precision: m = 1) and 16-byte aligned A, cholQ, g_A, g_cholQ; -101: not this route's call (the caller keeps
 * the three-call route).  Outputs and weights as for mf_kf_loglik_grad; the upper triangles of g_cholP0, g_cholQ are zeros;
 * g_b, g_H, g_y, g_omega may be NULL (not wanted: not computed into memory - a tenth of the pass's traffic).
 * ws: mf_kf_loglik_grad_streamed_workspace_bytes (0: not this route's call); it holds the posterior chain, (4 d^2 + 3 d) s bytes
 * per step.  chunks: time partitions per series, 0 = automatic (>= 2).  prof_start / prof_stop: optional hipEvent_t recorded
 * around the kernels.
 * fwd_ws (nullable): the workspace of the mf_kf_loglik call that evaluated the SAME inputs, untouched since, together with the
 * partition mf_kf_loglik_plan reports for that call (path 2: the streaming level-0 kernel, whose per-chunk summaries are the
 * first thing in its workspace).  The Schur complement of a chunk's interior does not depend on the direction of the
 * elimination, so those summaries replace the first two passes and the boundary scan: three passes instead of five.
 * Reference: TensorFlow reverse mode through markovflow/kalman_filter.py:184-255.
 */
size_t mf_kf_loglik_grad_streamed_workspace_bytes(int64_t B, int64_t T, int d, int m, int rinv_per_step, int elem_size,
                                                  int64_t chunks);
int mf_kf_loglik_grad_streamed_f64(int64_t B, int64_t T, int d, int m, const double* mu0, const double* cholP0, const double* A,
                                   const double* b, const double* cholQ, const double* H, const double* y, const double* Rinv,
                                   int rinv_per_step, const double* weights, double* g_mu0, double* g_cholP0, double* g_A,
                                   double* g_b, double* g_cholQ, double* g_H, double* g_y, double* g_omega, void* ws,
                                   size_t ws_bytes, int* info, int64_t chunks, const void* fwd_ws, int64_t fwd_chunks_per_series,
                                   int64_t fwd_chunk_length, void* prof_start, void* prof_stop, void* stream);
int mf_kf_loglik_grad_streamed_f32(int64_t B, int64_t T, int d, int m, const float* mu0, const float* cholP0, const float* A,
                                   const float* b, const float* cholQ, const float* H, const float* y, const float* Rinv,
                                   int rinv_per_step, const float* weights, float* g_mu0, float* g_cholP0, float* g_A,
                                   float* g_b, float* g_cholQ, float* g_H, float* g_y, float* g_omega, void* ws,
                                   size_t ws_bytes, int* info, int64_t chunks, const void* fwd_ws, int64_t fwd_chunks_per_series,
                                   int64_t fwd_chunk_length, void* prof_start, void* prof_stop, void* stream);
/*
 * posterior_state_space_model AFTER log_likelihood on the same inputs - the smoother reuses the filter's pass: fwd_ws is the
 * workspace of that mf_kf_loglik call (untouched since; partition from mf_kf_loglik_plan, path 2), whose per-chunk summaries
 * replace the first pass of the streamed kernels (mf_kf_posterior_chain with a workspace): one scan over the summaries for the
 * boundary states, then the emit pass alone (3.9-5.4 ms instead of 5.4-6.7 at B=1024, T=10000, d=6 fp64).  Same outputs.
 * ws: mf_kf_posterior_chain_from_filter_workspace_bytes (0: not covered).  -101: not this route's call.
 */
size_t mf_kf_posterior_chain_from_filter_workspace_bytes(int64_t B, int64_t T, int d, int m, int rinv_per_step, int elem_size,
                                                         int64_t fwd_chunks_per_series);
int mf_kf_posterior_chain_from_filter_f64(int64_t B, int64_t T, int d, int m, const double* mu0, const double* cholP0,
                                          const double* A, const double* b, const double* cholQ, const double* H, const double* y,
                                          const double* Rinv, int rinv_per_step, double* a_post, double* mu0_post, double* b_post,
                                          double* cholP0_post, double* cholQ_post, void* ws, size_t ws_bytes, int* info,
                                          const void* fwd_ws, int64_t fwd_chunks_per_series, int64_t fwd_chunk_length,
                                          void* prof_start, void* prof_stop, void* stream);
int mf_kf_posterior_chain_from_filter_f32(int64_t B, int64_t T, int d, int m, const float* mu0, const float* cholP0,
                                          const float* A, const float* b, const float* cholQ, const float* H, const float* y,
                                          const float* Rinv, int rinv_per_step, float* a_post, float* mu0_post, float* b_post,
                                          float* cholP0_post, float* cholQ_post, void* ws, size_t ws_bytes, int* info,
                                          const void* fwd_ws, int64_t fwd_chunks_per_series, int64_t fwd_chunk_length,
                                          void* prof_start, void* prof_stop, void* stream);

/*
 * The posterior of the same GP-regression models as a state space model (markovflow/models/gaussian_process_regression.py:130-148
 * builds its posterior process from it) without materialising the prior's tensors: boundary states from the summaries the fused
 * forward mf_gpr_matern_loglik left in its workspace (explicit chunk count, partition as for mf_gpr_matern_loglik_grad), then the
 * emit pass of the streamed posterior chain with the transitions generated in registers (csrc/mf_gpr_grad.hpp).  Outputs as
 * mf_kf_posterior_chain.  ws: mf_gpr_matern_posterior_chain_workspace_bytes.  -101: signature or partition not covered.
 */
size_t mf_gpr_matern_posterior_chain_workspace_bytes(int64_t B, int64_t T, int d, int elem_size, int64_t fwd_chunks_per_series);
int mf_gpr_matern_posterior_chain_f64(int64_t B, int64_t T, int ncomp, const int* orders, const double* lam, const double* var,
                                      int per_series, const double* t, const double* y, const double* rinv, double jitter,
                                      double* a_post, double* mu0_post, double* b_post, double* cholP0_post, double* cholQ_post,
                                      void* ws, size_t ws_bytes, int* info, const void* fwd_ws, int64_t fwd_chunks_per_series,
                                      int64_t fwd_chunk_length, void* stream);
int mf_gpr_matern_posterior_chain_f32(int64_t B, int64_t T, int ncomp, const int* orders, const float* lam, const float* var,
                                      int per_series, const float* t, const float* y, const float* rinv, float jitter,
                                      float* a_post, float* mu0_post, float* b_post, float* cholP0_post, float* cholQ_post,
                                      void* ws, size_t ws_bytes, int* info, const void* fwd_ws, int64_t fwd_chunks_per_series,
                                      int64_t fwd_chunk_length, void* stream);

/*
 * The backward of the fused GP-regression log-likelihood mf_gpr_matern_loglik: a Sum of one or two Matern components, one output - the
 * training step
 * of markovflow/models/gaussian_process_regression.py:150-160 under a GradientTape - with the kernel -> state-space-model step
 * fused into its passes (csrc/mf_gpr_grad.hpp): the transitions are generated in registers from (t, hyper-parameters) inside the
 * emit pass and the gradient pass of the streamed backward, 16 bytes of model per step; the model tensors never exist.
 * fwd_ws: the workspace of the mf_gpr_matern_loglik call on the SAME inputs with an EXPLICIT chunk count C, untouched since;
 * its partition is fwd_chunk_length = ceil((T-1) / C), fwd_chunks_per_series = ceil((T-1) / fwd_chunk_length) >= 2.
 * Outputs: g_packed [B,T-1,rec] - the derivative of sum_s weights[s] log p(y_s) with respect to the transitions and the Cholesky
 * factors of the process covariances, ONE record per transition: [the k x k diagonal block of g_A of every component, row-major |
 * the lower triangle of the diagonal block of g_cholQ of every component, row-major], each part padded to 16 bytes (30 values at
 * d = 3 + 3 in fp64 instead of 72) - the input of mf_sde_matern_transitions_grad_packed, which reduces it to the hyper-parameters -,
 * g_cholP0 [B,d,d] (the stationary prior's factor), g_omega [B,T] (nullable; E[r^2] per time point: the
 * derivative with respect to the noise PRECISION is -1/2 of it, + 1/2 R from the log-determinant, which the caller owns).
 * ws: mf_gpr_matern_loglik_grad_workspace_bytes.  -101: signature or partition not covered (materialise instead).
 */
size_t mf_gpr_matern_loglik_grad_workspace_bytes(int64_t B, int64_t T, int d, int elem_size, int64_t fwd_chunks_per_series);
int mf_gpr_matern_loglik_grad_f64(int64_t B, int64_t T, int ncomp, const int* orders, const double* lam, const double* var,
                                  int per_series, const double* t, const double* y, const double* rinv, double jitter,
                                  const double* weights, double* g_packed, double* g_cholP0, double* g_omega,
                                  void* ws, size_t ws_bytes, int* info, const void* fwd_ws, int64_t fwd_chunks_per_series,
                                  int64_t fwd_chunk_length, void* stream);
int mf_gpr_matern_loglik_grad_f32(int64_t B, int64_t T, int ncomp, const int* orders, const float* lam, const float* var,
                                  int per_series, const float* t, const float* y, const float* rinv, float jitter,
                                  const float* weights, float* g_packed, float* g_cholP0, float* g_omega,
                                  void* ws, size_t ws_bytes, int* info, const void* fwd_ws, int64_t fwd_chunks_per_series,
                                  int64_t fwd_chunk_length, void* stream);
/* The level-0 kernel (path: 0 row kernels, 1 spike-in-LDS, 2 streaming, 3 direct loads) and time partition mf_kf_loglik chooses
 * for a call; aligned16: A and cholQ are 16-byte aligned. */
int mf_kf_loglik_plan(int64_t B, int64_t T, int d, int m, int rinv_per_step, int elem_size, int64_t chunks, int aligned16,
                      int* path, int64_t* chunks_per_series, int64_t* chunk_length);

/*
 * Gradient of  KL(q1 || q2)  between two state space models (markovflow/state_space_model.py:528-593; differentiated by
 * TensorFlow in the reference, pinned by tests/integration/models/test_variational.py:123-132) with respect to the parameters
 * of q1: the adjoints of q1's marginal mean and covariance, lam_k = n_k + A^T lam_{k+1}, M_k = N_k + A^T M_{k+1} A, in three
 * kernels - per-step inputs (parallel), a light sequential recursion per series, per-step parameter gradients (parallel);
 * csrc/mf_kl_grad.hpp.  ws: mf_ssm_adjoint_workspace_bytes.
 * means_1 [B,T,d], covs_1 [B,T,d,d]: marginals of q1 (mf_ssm_marginal_means / mf_ssm_marginal_covariances).  Outputs per series:
 * g_mu0 [B,d], g_cholP0 [B,d,d] (lower), g_A [B,T-1,d,d], g_b [B,T-1,d], g_cholQ [B,T-1,d,d] (lower), scaled by weights [B]
 * (nullable).  adj_N [B,T,d,d], adj_n [B,T,d] (both or neither): the inputs of the recursion when the forward kept them
 * (mf_ssm_kl_divergence: out_N, out_n); NULL: they are formed here.
 * The gradient with respect to q2 is MINUS mf_kf_loglik_grad with H = NULL, q2's parameters and q1's moments.
 * State dimension 1..9.
 */
int mf_ssm_kl_grad_f64(int64_t B, int64_t T, int d, const double* mu0_1, const double* cholP0_1, const double* A_1,
                       const double* b_1, const double* cholQ_1, const double* mu0_2, const double* cholP0_2, const double* A_2,
                       const double* b_2, const double* cholQ_2, const double* means_1, const double* covs_1,
                       const double* weights, const double* adj_N, const double* adj_n, double* g_mu0, double* g_cholP0,
                       double* g_A, double* g_b, double* g_cholQ, void* ws, size_t ws_bytes, int* info, void* stream);
int mf_ssm_kl_grad_f32(int64_t B, int64_t T, int d, const float* mu0_1, const float* cholP0_1, const float* A_1,
                       const float* b_1, const float* cholQ_1, const float* mu0_2, const float* cholP0_2, const float* A_2,
                       const float* b_2, const float* cholQ_2, const float* means_1, const float* covs_1, const float* weights,
                       const float* adj_N, const float* adj_n, float* g_mu0, float* g_cholP0, float* g_A, float* g_b,
                       float* g_cholQ, void* ws, size_t ws_bytes, int* info, void* stream);

/*
 * KalmanFilter._r_inv (markovflow/kalman_filter.py:341-348): R^-1 = (L L^T)^-1 from the Cholesky factor L [m,m] of the shared
 * observation covariance (the reference: tf.linalg.cholesky_solve against the identity), out [m,m].  m <= 32; one launch.
 */
int mf_obs_precision_from_chol_f64(int m, const double* chol, double* out, int* info, void* stream);
int mf_obs_precision_from_chol_f32(int m, const float* chol, float* out, int* info, void* stream);

/*
 * BaseKalmanFilter.posterior_state_space_model (markovflow/kalman_filter.py:109-182) fused: the posterior precision and the
 * information vector (state_space_model.py:431-483, kalman_filter.py:86-101,149-156) are assembled block by block INSIDE the
 * backward U D U^T recursion (block_tri_diag.py:438-545) and the five tensors of the posterior chain are written directly:
 * a_post [B,T-1,d,d] (= -U^T), mu0_post [B,d], b_post [B,T-1,d], cholP0_post [B,d,d], cholQ_post [B,T-1,d,d].  Nothing else
 * touches HBM: the precision, its factor and the solves of the reference's route (mf_ssm_precision + mf_btd_udl here) never exist.
 * Rinv shared [m,m] or per step [B,T,m,m]; m <= 4; state dimension 1..9.  Two forms:
 *   ws == NULL           one lane per series, ONE backward sweep: (4 d^2 + 3 d + m d + m) s bytes per step.  For batches that
 *                        fill the chip (thousands of series) and for short chains.
 *   ws != NULL           (mf_kf_posterior_chain_workspace_bytes > 0 and a workspace of that size; d <= 6, m <= 3 or m = 1 with
 *                        per-step precisions, 16-byte aligned tensors - otherwise the call quietly takes the first form):
 *                        partitioned in TIME like mf_kf_loglik and streamed by LDS-DMA (csrc/mf_post_lds.hpp).  Pass 1: a lane
 *                        per (series, chunk) eliminates its chunk backwards with the fill-in carried to the chunk's right end;
 *                        pass 2: a scan over the chunk summaries gives the recursion's state at every chunk boundary; pass 3:
 *                        every chunk restarts the recursion there and writes the chain.  The inputs are read twice, coalesced,
 *                        every output once.  `chunks` = chunks per series (0 = fill the chip); prof_start / prof_stop: optional
 *                        hipEvent_t recorded on `stream` around the three kernels.
 */
size_t mf_kf_posterior_chain_workspace_bytes(int64_t B, int64_t T, int d, int m, int rinv_per_step, int elem_size,
                                             int64_t chunks);
int mf_kf_posterior_chain_f64(int64_t B, int64_t T, int d, int m, const double* mu0, const double* cholP0, const double* A,
                              const double* b, const double* cholQ, const double* H, const double* y, const double* Rinv,
                              int rinv_per_step, double* a_post, double* mu0_post, double* b_post, double* cholP0_post,
                              double* cholQ_post, void* ws, size_t ws_bytes, int* info, int64_t chunks, void* prof_start,
                              void* prof_stop, void* stream);
int mf_kf_posterior_chain_f32(int64_t B, int64_t T, int d, int m, const float* mu0, const float* cholP0, const float* A,
                              const float* b, const float* cholQ, const float* H, const float* y, const float* Rinv,
                              int rinv_per_step, float* a_post, float* mu0_post, float* b_post, float* cholP0_post,
                              float* cholQ_post, void* ws, size_t ws_bytes, int* info, int64_t chunks, void* prof_start,
                              void* prof_stop, void* stream);

/*
 * KL(q1 || q2) between two state space models, one scalar per series (markovflow/state_space_model.py:528-593), fused: ONE
 * forward sweep per series carries q1's marginal mean and covariance in registers and accumulates the divergence from the
 * local form  1/2 sum_k [ tr(Q2^-1 Q1) + tr(Q2^-1 dA S_k dA^T) + eps_k^T Q2^-1 eps_k ] - T d / 2 + log-determinants
 * (csrc/mf_kl_grad.hpp) - the ten parameter tensors are read once and nothing else touches HBM.  One lane per series when the
 * batch fills the chip.  With few, long chains (mf_ssm_kl_workspace_bytes > 0 and a workspace of that size) q1's marginals come
 * from the scans in time and the same local terms are formed by one lane per (series, step) and summed per series; on that
 * route the marginals of q1 can be kept: out_means [B,T,d], out_covs [B,T,d,d], out_cross [B,T-1,d,d] = Cov(x_{k+1}, x_k) (means
 * and covariances together, the cross-covariances optionally on top; -15 otherwise) - exactly what mf_ssm_kl_grad /
 * mf_kf_loglik_grad need; the sweep per series carries the same quantities in registers and writes them on request.
 * out_N [B,T,d,d], out_n [B,T,d] (both or neither; either route): N_k = dA_k^T Q2_k^-1 dA_k, n_k = dA_k^T Q2_k^-1 eps_k, the
 * inputs of the adjoint recursion of mf_ssm_kl_grad - by-products of the forward sweep that save the backward a kernel.
 * State dimension 1..15 (10..15: where mf_row_operators_cover holds).
 * State dimension 16..32, T > 1, every by-product pointer NULL and a workspace of mf_ssm_kl_workspace_bytes: the value alone, in
 * the reference's operator form (the block rows of q2's precision against q1's marginal / subsequent covariances,
 * state_space_model.py:569-593) evaluated in ONE walk per (series, chunk of the time axis) on 16 x 16 MFMA register tiles
 * (csrc/mf_wave_ops.hpp, wave_kl_walk_kernel): q1's moment recursion, q2's means and the block terms together, neither the moments
 * nor the precision in memory; -100 with by-products requested (their consumers, mf_ssm_kl_grad_*, stop at d = 9), -15 without
 * the workspace.  `info` is not written on this route (a non-positive diagonal of a factor yields NaN in `out`).
 */
size_t mf_ssm_kl_workspace_bytes(int64_t B, int64_t T, int d, int elem_size);
int mf_ssm_kl_divergence_f64(int64_t B, int64_t T, int d, const double* mu0_1, const double* cholP0_1, const double* A_1,
                             const double* b_1, const double* cholQ_1, const double* mu0_2, const double* cholP0_2,
                             const double* A_2, const double* b_2, const double* cholQ_2, double* out, double* out_means,
                             double* out_covs, double* out_cross, double* out_N, double* out_n, void* ws, size_t ws_bytes,
                             int* info, void* stream);
int mf_ssm_kl_divergence_f32(int64_t B, int64_t T, int d, const float* mu0_1, const float* cholP0_1, const float* A_1,
                             const float* b_1, const float* cholQ_1, const float* mu0_2, const float* cholP0_2, const float* A_2,
                             const float* b_2, const float* cholQ_2, float* out, float* out_means, float* out_covs,
                             float* out_cross, float* out_N, float* out_n, void* ws, size_t ws_bytes, int* info, void* stream);

/*
 * KL(q1 || q2) for 16 <= d <= 32 from q1's MOMENTS (markovflow/state_space_model.py:528-593: the reference assembles q2's precision,
 * multiplies it block by block with q1's marginal / subsequent covariances and sums, :569-573).  One wavefront per (series, block)
 * forms block row k of q2's precision on register tiles (as mf_ssm_precision does) and reduces it on the spot against
 * covs_1 [B,T,d,d], cross_1 [B,T-1,d,d] = Cov(x_{k+1}, x_k) and mean_diff [B,T,d] = mu_2 - mu_1; the two log-determinants ride along
 * (q1's factors are read on their diagonals only); a second small kernel sums the T terms of a series in a fixed order.  q2's
 * precision never exists in memory.  ws: mf_ssm_kl_from_moments_workspace_bytes (B T scalars).  -100 outside 16 <= d <= 32.
 */
size_t mf_ssm_kl_from_moments_workspace_bytes(int64_t B, int64_t T, int d, int elem_size);
int mf_ssm_kl_from_moments_f64(int64_t B, int64_t T, int d, const double* cholP0_1, const double* cholQ_1, const double* cholP0_2,
                               const double* A_2, const double* cholQ_2, const double* covs_1, const double* cross_1,
                               const double* mean_diff, double* out, void* ws, size_t ws_bytes, void* stream);
int mf_ssm_kl_from_moments_f32(int64_t B, int64_t T, int d, const float* cholP0_1, const float* cholQ_1, const float* cholP0_2,
                               const float* A_2, const float* cholQ_2, const float* covs_1, const float* cross_1,
                               const float* mean_diff, float* out, void* ws, size_t ws_bytes, void* stream);

/*
 * Adjoint of the marginal recursion  m_{k+1} = A_k m_k + b_k,  S_{k+1} = A_k S_k A_k^T + Q_k  (markovflow/state_space_model.py:232-262,
 * differentiated by TensorFlow in the reference: the expected log-likelihood of every variational model goes through
 * `marginals`, models/variational.py:150, models/sparse_variational.py:178-192).  Given the incoming gradients g_means [B,T,d] and
 * g_covs [B,T,d,d] (either may be NULL = zero) of a scalar with respect to the marginal means / covariances, and the
 * marginals themselves (means, covs), the gradients with respect to mu0, cholP0 (lower), A, b and cholQ (lower): the same
 * recursion as mf_ssm_kl_grad with (N_k, n_k) = (g_covs_k + g_covs_k^T, g_means_k), then a parallel kernel.  ws:
 * mf_ssm_adjoint_workspace_bytes.  State dimension 1..9.
 */
int mf_ssm_marginals_grad_f64(int64_t B, int64_t T, int d, const double* cholP0, const double* A, const double* cholQ,
                              const double* means, const double* covs, const double* g_means, const double* g_covs,
                              double* g_mu0, double* g_cholP0, double* g_A, double* g_b, double* g_cholQ, void* ws,
                              size_t ws_bytes, void* stream);
int mf_ssm_marginals_grad_f32(int64_t B, int64_t T, int d, const float* cholP0, const float* A, const float* cholQ,
                              const float* means, const float* covs, const float* g_means, const float* g_covs, float* g_mu0,
                              float* g_cholP0, float* g_A, float* g_b, float* g_cholQ, void* ws, size_t ws_bytes, void* stream);

/* Workspace of mf_ssm_kl_grad_* and mf_ssm_marginals_grad_*: per (series, block) the inputs (N_k, n_k) and the results
 * (M_k, lambda_k) of the adjoint recursion - (2 d^2 + 2 d) elements. */
size_t mf_ssm_adjoint_workspace_bytes(int64_t B, int64_t T, int d, int elem_size);

#ifdef __cplusplus
}
#endif
#endif /* MARKOVFLOW_AMD_H */
