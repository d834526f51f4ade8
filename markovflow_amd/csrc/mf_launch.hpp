// Per-state-dimension launch tables.  One translation unit (mf_inst.hip, compiled with -DMF_D=<d>)
// instantiates every kernel for that d and both scalar types and exports its table; mf_api.hip
// dispatches the extern "C" entry points of include/markovflow_amd.h through these tables.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "mf_env.hpp"

namespace mf {

template <typename T> struct OpsTable {
    size_t (*kf_loglik_ws)(long B, long Tn, long chunks);
    int (*kf_loglik)(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ,
                     const T* H, const T* y, const T* Rinv, int rinv_per_step, T add_const, T* out, void* ws,
                     size_t ws_bytes, int* info, long chunks, hipEvent_t ev0, hipEvent_t ev1, hipStream_t st);
    size_t (*btd_logdet_quad_ws)(long B, long n, long chunks);
    int (*btd_logdet_quad)(long B, long n, const T* diag, const T* sub, const T* rhs, T* out, void* ws,
                           size_t ws_bytes, int* info, long chunks, hipStream_t st);
    size_t (*btd_cholesky_ws)(long B, long n);
    int (*btd_cholesky)(long B, long n, const T* diag, const T* sub, T* ldiag, T* lsub, void* ws, size_t ws_bytes,
                        int* info, hipStream_t st);
    size_t (*btd_solve_ws)(long Bl, long Br, long n);
    int (*btd_solve)(long Bl, long Br, long n, const T* ldiag, const T* lsub, const T* rhs, T* out, int transpose,
                     void* ws, size_t ws_bytes, hipStream_t st);
    int (*btd_matvec)(long Bl, long Br, long n, const T* diag, const T* sub, const T* x, T* out, int mode,
                      hipStream_t st);
    int (*btd_logdet)(long B, long n, const T* ldiag, T* out, hipStream_t st);
    size_t (*btd_diag_of_inverse_ws)(long B, long n);
    int (*btd_diag_of_inverse)(long B, long n, const T* ldiag, const T* lsub, T* odiag, T* osub, void* ws, size_t ws_bytes,
                               hipStream_t st);
    int (*ssm_marginal_covs)(long B, long n, const T* cholP0, const T* A, const T* cholQ, T* ocov, T* osub, void* ws,
                             size_t ws_bytes, hipStream_t st);
    size_t (*btd_udl_ws)(long B, long n);
    int (*btd_udl)(long B, long n, const T* diag, const T* sub, T* ut, T* chol_d, const T* eta, T* m_post,
                   T* chol_dinv, int chain_layout, void* ws, size_t ws_bytes, int* info, hipStream_t st);
    int (*ssm_precision)(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b,
                         const T* cholQ, const T* H, const T* y, const T* Rinv, int rinv_per_step, T* diag, T* sub,
                         T* eta, hipStream_t st);
    int (*ssm_means)(long Bl, long Br, long Tn, const T* A, const T* offs, T* out, void* ws, size_t ws_bytes,
                     hipStream_t st);
    int (*block_matmul)(long B, long n, const T* X, long xs, const T* Y, long ys, T* out, hipStream_t st);
    int (*gpr_loglik)(long B, long Tn, int ncomp, const int* orders, const T* lam, const T* var, int per_series, const T* t,
                      const T* y, int m, int multi, const T* rinv, T jitter, T add_const, T* out, void* ws, size_t ws_bytes,
                      int* info, long chunks, hipEvent_t ev0, hipEvent_t ev1, hipStream_t st);
    int (*sde_predict)(long B, long N, long Np, const long long* idx, const T* Amt, const T* Qmt, const T* Atp, const T* Qtp,
                       const T* means, const T* covs, const T* subseq, const T* m0, const T* P0, T* omean, T* ocov,
                       int* info, hipStream_t st);
    int (*kf_grad)(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
                   const T* y, const T* Rinv, const T* pm, const T* pS, const T* pX, T* gmu0, T* gC0, T* gA, T* gb, T* gC,
                   T* gH, T* gy, T* gOm, const T* weights, int rinv_per_step, int* info, hipStream_t st);
    int (*kl_grad)(long B, long Tn, const T* mu0_1, const T* C0_1, const T* A_1, const T* b_1, const T* C_1, const T* mu0_2,
                   const T* C0_2, const T* A_2, const T* b_2, const T* C_2, const T* pm, const T* pS, const T* weights,
                   const T* in_N, const T* in_n, T* gmu0, T* gC0, T* gA, T* gb, T* gC, void* ws, size_t ws_bytes, int* info,
                   hipStream_t st);
    int (*posterior_chain)(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ,
                           const T* H, const T* y, const T* Rinv, int rinv_per_step, T* a_post, T* mu0_post, T* b_post,
                           T* cp0_post, T* cq_post, int* info, hipStream_t st);
    int (*kl)(long B, long Tn, const T* mu0_1, const T* C0_1, const T* A_1, const T* b_1, const T* C_1, const T* mu0_2,
              const T* C0_2, const T* A_2, const T* b_2, const T* C_2, T* out, T* out_means, T* out_covs, T* out_cross,
              T* out_N, T* out_n, void* ws, size_t ws_bytes, int* info, hipStream_t st);
    int (*marginals_grad)(long B, long Tn, const T* C0, const T* A, const T* C, const T* pm, const T* pS, const T* gm,
                          const T* gS, T* gmu0, T* gC0, T* gA, T* gb, T* gC, void* ws, size_t ws_bytes, hipStream_t st);
    size_t (*adjoint_ws)(long B, long Tn);
    int (*ssm_marginals)(long B, long n, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, T* omean,
                         T* ocov, T* osub, void* ws, size_t ws_bytes, hipStream_t st);
    size_t (*kl_ws)(long B, long Tn);
    size_t (*marginals_ws)(long B, long n);
    // the level-0 kernel and time partition kf_loglik chooses: path 2 = the streaming kernel (mf_kf_lds.hpp), whose summaries
    // (P per series, L transitions each) sit at the start of the workspace
    int (*kf_loglik_plan)(long B, long Tn, int m, int rinv_per_step, long chunks, int aligned16, int* path, long* P, long* L);
    // conditional_statistics (conditionals.py:87-203): P_t [n, d, 2d], T_t [n, d, d] from the transitions around every new point
    int (*sde_cond_stats)(long n, const T* Amt, const T* Qmt, const T* Atp, const T* Qtp, T* proj, T* cov, int* info, hipStream_t st);
    // reverse mode of cholesky / block_diagonal_of_inverse (block_tri_diag.py:22-31: banded_matrices' registered gradients)
    size_t (*btd_grad_ws)(long B, long n);
    int (*btd_cholesky_grad)(long B, long n, const T* ldiag, const T* lsub, const T* g_ldiag, const T* g_lsub, T* g_diag, T* g_sub,
                             void* ws, size_t ws_bytes, hipStream_t st);
    int (*btd_diag_of_inverse_grad)(long B, long n, const T* ldiag, const T* lsub, const T* sigma, const T* g_diag, const T* g_sub,
                                    T* g_ldiag, T* g_lsub, void* ws, size_t ws_bytes, hipStream_t st);
};

// streamed, time-partitioned posterior chain (mf_post_lds.hpp, instantiated by mf_post_inst.hip for d = 1 ... MF_MAX_D_POST)
template <typename T> struct PostOps {
    size_t (*ws)(long B, long Tn, int m, int rinv_per_step, long chunks);      // 0: not this route's call
    // a_post = NULL: the chain without its transitions (the streamed backward of log_likelihood does not read them)
    int (*chain)(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
                 const T* y, const T* Rinv, int rinv_per_step, T* a_post, T* mu0_post, T* b_post, T* cp0_post, T* cq_post,
                 void* ws, size_t ws_bytes, int* info, long chunks, hipEvent_t ev0, hipEvent_t ev1, hipStream_t st);
    int (*plan)(long B, long Tn, int m, int rinv_per_step, long chunks, long* P, long* L);   // the partition `chain` makes of `chunks`
    // pass 3 alone on the partition (P, L), from boundary states the caller has put into the workspace (PostWs: bPsi, bpsi)
    int (*emit)(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
                const T* y, const T* Rinv, int rinv_per_step, T* a_post, T* mu0_post, T* b_post, T* cp0_post, T* cq_post,
                void* ws, size_t ws_bytes, int* info, long P, long L, hipStream_t st);
};
constexpr int MF_MAX_D_POST = 6;

// streamed backward of KalmanFilter.log_likelihood (mf_grad_lds.hpp, mf_grad_inst.hip; the same state dimensions)
template <typename T> struct GradOps {
    size_t (*ws)(long B, long Tn, int m, int rinv_per_step, long chunks);      // 0: not this route's call
    int (*run)(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
               const T* y, const T* Rinv, int rinv_per_step, const T* weights, T* g_mu0, T* g_cholP0, T* g_A, T* g_b, T* g_cholQ,
               T* g_H, T* g_y, T* g_Om, void* ws, size_t ws_bytes, int* info, long chunks, const void* fwd_ws, long fwd_P,
               long fwd_L, hipEvent_t ev0, hipEvent_t ev1, hipStream_t st);
    // posterior_state_space_model from the summaries a preceding mf_kf_loglik call on the same inputs left in its workspace (the
    // smoother reuses the filter's pass): boundary states by a scan over those summaries, then the emit pass alone
    size_t (*post_from_fwd_ws)(long B, long Tn, int m, int rinv_per_step, long fwd_P);
    int (*post_from_fwd)(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
                         const T* y, const T* Rinv, int rinv_per_step, T* a_post, T* mu0_post, T* b_post, T* cp0_post, T* cq_post,
                         void* ws, size_t ws_bytes, int* info, const void* fwd_ws, long fwd_P, long fwd_L, hipEvent_t ev0,
                         hipEvent_t ev1, hipStream_t st);
    // GPR with the kernel -> state-space-model step fused (mf_gpr_grad.hpp); -101: signature / partition not covered
    size_t (*gpr_ws)(long B, long Tn, long fwd_P);
    int (*gpr_run)(long B, long Tn, int ncomp, const int* orders, const T* lam, const T* var, int per_series, const T* t,
                   const T* y, const T* rinv, T jitter, const T* weights, T* g_packed, T* g_cholP0, T* g_Om, void* ws,
                   size_t ws_bytes, int* info, const void* fwd_ws, long fwd_P, long fwd_L, hipStream_t st);
    // ... and its posterior_state_space_model from the fused forward's summaries
    size_t (*gpr_post_ws)(long B, long Tn, long fwd_P);
    int (*gpr_post_run)(long B, long Tn, int ncomp, const int* orders, const T* lam, const T* var, int per_series, const T* t,
                        const T* y, const T* rinv, T jitter, T* a_post, T* mu0_post, T* b_post, T* cp0_post, T* cq_post, void* ws,
                        size_t ws_bytes, int* info, const void* fwd_ws, long fwd_P, long fwd_L, hipStream_t st);
};

constexpr int MF_MAX_D = 9;        // largest state dimension with a register-resident (lane per chunk) instantiation
constexpr int MF_MAX_D_ROW = 15;   // largest state dimension of the row kernels (one 16-lane row per chunk; 10 ... 15: only those)
constexpr int MF_MAX_D_BIG = 64;      // largest state dimension of the LDS-tiled MFMA path, fp32 (log-likelihood only)
constexpr int MF_MAX_D_BIG_F64 = 32;  // the same in fp64 (seven d x d tiles must fit the 160 KB of LDS)
constexpr int MF_MAX_D_LOGLIK_F64 = 64;   // mf_kf_loglik_f64 alone: the panel kernels (mf_panel.hpp) carry fp64 to d = 64

// mf_big_inst.hip
size_t big_kf_loglik_ws(long B, long Tn, int d, long chunks, int elem_size);
size_t big_marginal_covs_ws(long B, long n, int d, int elem_size);
int big_kf_loglik_f64(long B, long Tn, int d, int m, const double* mu0, const double* cholP0, const double* A,
                      const double* b, const double* cholQ, const double* H, const double* y, const double* Rinv,
                      int rinv_per_step, double add_const, double* out, void* ws, size_t ws_bytes, int* info, long chunks,
                      hipEvent_t ev0, hipEvent_t ev1, hipStream_t st);
int big_kf_loglik_f32(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                      const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step,
                      float add_const, float* out, void* ws, size_t ws_bytes, int* info, long chunks, hipEvent_t ev0,
                      hipEvent_t ev1, hipStream_t st);

// mf_wave_inst.hip: level 0 of the log-likelihood on register tiles, one wavefront per (series, chunk), 16 <= d <= 32, m <= 4
// (mf_wave.hpp); the reduced system has the layout of the tile engine's, whose levels take it from there.  -101: not covered.
template <typename T> struct RedSys;
bool wave_covers(int d, int m);
int wave_waves_per_simd(int d, int elem_size);
int wave_kf_level0_f64(long B, long Tn, int d, int m, const double* mu0, const double* cholP0, const double* A, const double* b,
                       const double* cholQ, const double* H, const double* y, const double* Rinv, int rinv_per_step, long P, long L,
                       const RedSys<double>& out, int* info, hipStream_t st);
int wave_kf_level0_f32(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                       const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step, long P, long L,
                       const RedSys<float>& out, int* info, hipStream_t st);

// mf_panel_inst.hip: the log-likelihood on register PANELS, one workgroup of d / 16 wavefronts per (series, chunk), 32 < d <= 64
// (mf_panel.hpp): level 0 and the reduction levels behind it (same reduced system as the tile engine's).
bool panel_covers(int d, int m);
int panel_kf_level0_f64(long B, long Tn, int d, int m, const double* mu0, const double* cholP0, const double* A, const double* b,
                        const double* cholQ, const double* H, const double* y, const double* Rinv, int rinv_per_step, long P, long L,
                        const RedSys<double>& out, int* info, hipStream_t st);
int panel_kf_level0_f32(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                        const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step, long P, long L,
                        const RedSys<float>& out, int* info, hipStream_t st);
// StateSpaceModel._build_precision (+ H^T R^-1 H, + information vector) for 32 < d <= 64 on the panel kernels; -101: not covered
int panel_ssm_precision_f64(long B, long Tn, int d, int m, const double* mu0, const double* cholP0, const double* A, const double* b,
                            const double* cholQ, const double* H, const double* y, const double* Rinv, int rinv_per_step, double* diag,
                            double* sub, double* eta, hipStream_t st);
int panel_ssm_precision_f32(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                            const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step, float* diag,
                            float* sub, float* eta, hipStream_t st);
// one reduction level RedSys(in.n) -> RedSys(P) (final: P = 1, out_scalar[s] = add_const + the series' value)
int panel_red_f64(const RedSys<double>& in, const RedSys<double>& out, long B, long P, int d, double add_const, double* out_scalar,
                  int* info, int final_level, hipStream_t st);
int panel_red_f32(const RedSys<float>& in, const RedSys<float>& out, long B, long P, int d, float add_const, float* out_scalar,
                  int* info, int final_level, hipStream_t st);

// mf_adj.hip: reverse mode through cholesky / block_diagonal_of_inverse for 10 <= d <= 32 on register MFMA tiles: parallel in time
// (local terms + congruence scans; needs adj_grad_ws bytes) or, without a workspace / for short chains, a wavefront per series
bool adj_covers(int d);
size_t adj_grad_ws(long B, long n, int d, int elem_size);
template <typename T>
int adj_cholesky_grad(long B, long n, int d, const T* ldiag, const T* lsub, const T* g_ldiag, const T* g_lsub, T* g_diag, T* g_sub,
                      void* ws, size_t ws_bytes, hipStream_t st);
template <typename T>
int adj_diag_of_inverse_grad(long B, long n, int d, const T* ldiag, const T* lsub, const T* sigma, const T* g_diag, const T* g_sub,
                             T* g_ldiag, T* g_lsub, void* ws, size_t ws_bytes, hipStream_t st);

#define MF_DECLARE_BIG(SUF, T)                                                                                               \
    int big_cholesky_##SUF(long B, long n, int d, const T* diag, const T* sub, T* ldiag, T* lsub, void* ws, size_t ws_bytes,  \
                           int* info, hipStream_t st);                                                                       \
    int big_solve_##SUF(long Bl, long Br, long n, int d, const T* ldiag, const T* lsub, const T* rhs, T* out, int transpose,  \
                        void* ws, size_t ws_bytes, hipStream_t st);                                                          \
    int big_matvec_##SUF(long Bl, long Br, long n, int d, const T* diag, const T* sub, const T* x, T* out, int mode,          \
                         hipStream_t st);                                                                                    \
    int big_logdet_##SUF(long B, long n, int d, const T* ldiag, T* out, hipStream_t st);                                     \
    int big_diag_of_inverse_##SUF(long B, long n, int d, const T* ldiag, const T* lsub, T* odiag, T* osub, void* ws,          \
                                  size_t ws_bytes, hipStream_t st);                                                          \
    int big_udl_##SUF(long B, long n, int d, const T* diag, const T* sub, T* ut, T* chol_d, const T* eta, T* m_post,          \
                      T* chol_dinv, void* ws, size_t ws_bytes, int* info, hipStream_t st);                                   \
    int big_ssm_precision_##SUF(long B, long Tn, int d, int m, const T* mu0, const T* cholP0, const T* A, const T* b,         \
                                const T* cholQ, const T* H, const T* y, const T* Rinv, int rinv_per_step, T* diag, T* sub,    \
                                T* eta, hipStream_t st);                                                                     \
    int big_means_##SUF(long Bl, long Br, long Tn, int d, const T* A, const T* offs, T* out, void* ws, size_t ws_bytes,       \
                        hipStream_t st);                                                                                     \
    int big_block_matmul_##SUF(long B, long n, int d, const T* X, long xs, const T* Y, long ys, T* out, hipStream_t st);    \
    int big_kf_grad_##SUF(long B, long Tn, int d, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, \
                          const T* H, const T* y, const T* Rinv, int rinv_per_step, const T* mean, const T* cov,             \
                          const T* cross, const T* w, T* g_mu0, T* g_cholP0, T* g_A, T* g_b, T* g_cholQ, T* g_H, T* g_y,     \
                          T* g_om, hipStream_t st);                                                                          \
    int big_marginal_covs_##SUF(long B, long n, int d, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ,  \
                                T* omean, T* ocov, T* osub, void* ws, size_t ws_bytes, hipStream_t st);
size_t big_btd_par_ws(long B, long n, int d, int chain, int elem_size);
size_t big_btd_solve_ws(long Bl, long Br, long n, int d, int elem_size);
size_t big_btd_tak_ws(long B, long n, int d, int elem_size);
MF_DECLARE_BIG(f32, float)
MF_DECLARE_BIG(f64, double)
#undef MF_DECLARE_BIG

#define MF_DECLARE_TABLES(D)                          \
    const OpsTable<float>* ops_f32_d##D();            \
    const OpsTable<double>* ops_f64_d##D();
MF_DECLARE_TABLES(1) MF_DECLARE_TABLES(2) MF_DECLARE_TABLES(3) MF_DECLARE_TABLES(4) MF_DECLARE_TABLES(5)
MF_DECLARE_TABLES(6) MF_DECLARE_TABLES(7) MF_DECLARE_TABLES(8) MF_DECLARE_TABLES(9)
MF_DECLARE_TABLES(10) MF_DECLARE_TABLES(11) MF_DECLARE_TABLES(12) MF_DECLARE_TABLES(13) MF_DECLARE_TABLES(14) MF_DECLARE_TABLES(15)
#undef MF_DECLARE_TABLES
#define MF_DECLARE_POST(D)                            \
    const PostOps<float>* post_ops_f32_d##D();        \
    const PostOps<double>* post_ops_f64_d##D();       \
    const GradOps<float>* grad_ops_f32_d##D();        \
    const GradOps<double>* grad_ops_f64_d##D();
MF_DECLARE_POST(1) MF_DECLARE_POST(2) MF_DECLARE_POST(3) MF_DECLARE_POST(4) MF_DECLARE_POST(5) MF_DECLARE_POST(6)
#undef MF_DECLARE_POST

}  // namespace mf
