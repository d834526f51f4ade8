// Entry points of mf_wave_inst.hip (declared in mf_launch.hpp) as one overloaded name, for the type-generic launcher of the tile
// engine (mf_big_impl.hpp is included once per scalar type).
#pragma once
#include "mf_launch.hpp"

namespace mf {
inline int wave_kf_level0(long B, long Tn, int d, int m, const double* mu0, const double* cholP0, const double* A, const double* b,
                          const double* cholQ, const double* H, const double* y, const double* Rinv, int rinv_per_step, long P, long L,
                          const RedSys<double>& out, int* info, hipStream_t st) {
    return wave_kf_level0_f64(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, out, info, st);
}
inline int wave_kf_level0(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                          const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step, long P, long L,
                          const RedSys<float>& out, int* info, hipStream_t st) {
    return wave_kf_level0_f32(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, out, info, st);
}
inline int panel_kf_level0(long B, long Tn, int d, int m, const double* mu0, const double* cholP0, const double* A, const double* b,
                           const double* cholQ, const double* H, const double* y, const double* Rinv, int rinv_per_step, long P, long L,
                           const RedSys<double>& out, int* info, hipStream_t st) {
    return panel_kf_level0_f64(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, out, info, st);
}
inline int panel_kf_level0(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                           const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step, long P, long L,
                           const RedSys<float>& out, int* info, hipStream_t st) {
    return panel_kf_level0_f32(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, out, info, st);
}
inline int panel_red(const RedSys<double>& in, const RedSys<double>& out, long B, long P, int d, double add_const, double* out_scalar,
                     int* info, int final_level, hipStream_t st) {
    return panel_red_f64(in, out, B, P, d, add_const, out_scalar, info, final_level, st);
}
inline int panel_red(const RedSys<float>& in, const RedSys<float>& out, long B, long P, int d, float add_const, float* out_scalar,
                     int* info, int final_level, hipStream_t st) {
    return panel_red_f32(in, out, B, P, d, add_const, out_scalar, info, final_level, st);
}
// mf_wave_inst.hip: precision assembly on register tiles (wave_ssm_precision_kernel); -101: not covered
int wave_ssm_precision_f64(long B, long Tn, int d, int m, const double* mu0, const double* cholP0, const double* A, const double* b,
                           const double* cholQ, const double* H, const double* y, const double* Rinv, int rinv_per_step, double* diag,
                           double* sub, double* eta, hipStream_t st);
int wave_ssm_precision_f32(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                           const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step, float* diag,
                           float* sub, float* eta, hipStream_t st);
inline int wave_ssm_precision(long B, long Tn, int d, int m, const double* mu0, const double* cholP0, const double* A, const double* b,
                              const double* cholQ, const double* H, const double* y, const double* Rinv, int rinv_per_step, double* diag,
                              double* sub, double* eta, hipStream_t st) {
    return wave_ssm_precision_f64(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, diag, sub, eta, st);
}
inline int wave_ssm_precision(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                              const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step, float* diag,
                              float* sub, float* eta, hipStream_t st) {
    return wave_ssm_precision_f32(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, diag, sub, eta, st);
}
// mf_wave_inst.hip: the per-block terms of KL(chain 1 || chain 2) on register tiles + their sum (wave_ssm_kl_terms_kernel); ws: B Tn scalars
int wave_ssm_kl_f64(long B, long Tn, int d, const double* cp0_1, const double* cq_1, const double* cp0_2, const double* a_2,
                    const double* cq_2, const double* cov, const double* cross, const double* mdiff, double* out, void* ws,
                    size_t ws_bytes, hipStream_t st);
int wave_ssm_kl_f32(long B, long Tn, int d, const float* cp0_1, const float* cq_1, const float* cp0_2, const float* a_2,
                    const float* cq_2, const float* cov, const float* cross, const float* mdiff, float* out, void* ws, size_t ws_bytes,
                    hipStream_t st);
// mf_panel_inst.hip: the up-sweep of the time-partitioned factorisations for 32 < d <= 64 (panel_red_kernel in operator mode); -101: not covered
// (piv != NULL: the pass over the chunk ends as well - the natural-order pivot of the last block of every chunk, [B, P, d, d])
int panel_chol_up_f64(long B, long n, int d, long P, long L, const double* diag, const double* sub, double* oDv, double* oGU, double* oF,
                      double* piv, int rev, int* info, hipStream_t st);
int panel_chol_up_f32(long B, long n, int d, long P, long L, const float* diag, const float* sub, float* oDv, float* oGU, float* oF,
                      float* piv, int rev, int* info, hipStream_t st);
inline int panel_chol_up(long B, long n, int d, long P, long L, const double* diag, const double* sub, double* oDv, double* oGU, double* oF,
                         double* piv, int rev, int* info, hipStream_t st) {
    return panel_chol_up_f64(B, n, d, P, L, diag, sub, oDv, oGU, oF, piv, rev, info, st);
}
inline int panel_chol_up(long B, long n, int d, long P, long L, const float* diag, const float* sub, float* oDv, float* oGU, float* oF,
                         float* piv, int rev, int* info, hipStream_t st) {
    return panel_chol_up_f32(B, n, d, P, L, diag, sub, oDv, oGU, oF, piv, rev, info, st);
}
// mf_panel_inst.hip: the emit pass of the time-partitioned cholesky for 32 < d <= 64 (panel_chol_emit_kernel); -101: not covered
int panel_chol_emit_f64(long B, long n, int d, long P, long L, const double* diag, const double* sub, const double* piv, double* ldiag,
                        double* lsub, int* info, hipStream_t st);
int panel_chol_emit_f32(long B, long n, int d, long P, long L, const float* diag, const float* sub, const float* piv, float* ldiag,
                        float* lsub, int* info, hipStream_t st);
inline int panel_chol_emit(long B, long n, int d, long P, long L, const double* diag, const double* sub, const double* piv, double* ldiag,
                           double* lsub, int* info, hipStream_t st) {
    return panel_chol_emit_f64(B, n, d, P, L, diag, sub, piv, ldiag, lsub, info, st);
}
inline int panel_chol_emit(long B, long n, int d, long P, long L, const float* diag, const float* sub, const float* piv, float* ldiag,
                           float* lsub, int* info, hipStream_t st) {
    return panel_chol_emit_f32(B, n, d, P, L, diag, sub, piv, ldiag, lsub, info, st);
}
inline int panel_ssm_precision(long B, long Tn, int d, int m, const double* mu0, const double* cholP0, const double* A, const double* b,
                               const double* cholQ, const double* H, const double* y, const double* Rinv, int rinv_per_step, double* diag,
                               double* sub, double* eta, hipStream_t st) {
    return panel_ssm_precision_f64(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, diag, sub, eta, st);
}
inline int panel_ssm_precision(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                               const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step, float* diag,
                               float* sub, float* eta, hipStream_t st) {
    return panel_ssm_precision_f32(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, diag, sub, eta, st);
}
// mf_wave_inst.hip: StateSpaceModel.kl_divergence in one walk per (series, chunk) (wave_kl_walk_kernel); -101: not covered, -15: workspace
size_t wave_ssm_kl_fused_ws(long B, long n, int d, int elem_size);
int wave_ssm_kl_fused_f64(long B, long n, int d, const double* mu0_1, const double* cp0_1, const double* a_1, const double* b_1,
                          const double* cq_1, const double* mu0_2, const double* cp0_2, const double* a_2, const double* b_2,
                          const double* cq_2, double* out, void* ws, size_t ws_bytes, hipStream_t st);
int wave_ssm_kl_fused_f32(long B, long n, int d, const float* mu0_1, const float* cp0_1, const float* a_1, const float* b_1,
                          const float* cq_1, const float* mu0_2, const float* cp0_2, const float* a_2, const float* b_2,
                          const float* cq_2, float* out, void* ws, size_t ws_bytes, hipStream_t st);
// mf_wave_inst.hip: solve with the time axis walked serially inside a wavefront (mf_wave_ops.hpp); -101: not covered
int wave_btd_solve_f64(long Bl, long Br, long n, int d, const double* ldiag, const double* lsub, const double* rhs, double* out,
                       int transpose, void* ws, size_t ws_bytes, hipStream_t st);
int wave_btd_solve_f32(long Bl, long Br, long n, int d, const float* ldiag, const float* lsub, const float* rhs, float* out, int transpose,
                       void* ws, size_t ws_bytes, hipStream_t st);
inline int wave_btd_solve(long Bl, long Br, long n, int d, const double* ldiag, const double* lsub, const double* rhs, double* out,
                          int transpose, void* ws, size_t ws_bytes, hipStream_t st) {
    return wave_btd_solve_f64(Bl, Br, n, d, ldiag, lsub, rhs, out, transpose, ws, ws_bytes, st);
}
inline int wave_btd_solve(long Bl, long Br, long n, int d, const float* ldiag, const float* lsub, const float* rhs, float* out, int transpose,
                          void* ws, size_t ws_bytes, hipStream_t st) {
    return wave_btd_solve_f32(Bl, Br, n, d, ldiag, lsub, rhs, out, transpose, ws, ws_bytes, st);
}
// marginal_means at 16 <= d <= 32 partitioned in time: the chunk maps and the mean every chunk starts from (mf_wave_ops.hpp), in the
// solve's workspace; *P = 1: not partitioned (nothing launched).  The walk per chunk is the caller's (bigop_means_kernel).
int wave_means_boundaries_f64(long Bl, long Br, long n, int d, const double* A, const double* offs, void* ws, size_t ws_bytes, long* P,
                              long* Lc, const double** m_in, hipStream_t st);
int wave_means_boundaries_f32(long Bl, long Br, long n, int d, const float* A, const float* offs, void* ws, size_t ws_bytes, long* P,
                              long* Lc, const float** m_in, hipStream_t st);
inline int wave_means_boundaries(long Bl, long Br, long n, int d, const double* A, const double* offs, void* ws, size_t ws_bytes, long* P,
                                 long* Lc, const double** m_in, hipStream_t st) {
    return wave_means_boundaries_f64(Bl, Br, n, d, A, offs, ws, ws_bytes, P, Lc, m_in, st);
}
inline int wave_means_boundaries(long Bl, long Br, long n, int d, const float* A, const float* offs, void* ws, size_t ws_bytes, long* P,
                                 long* Lc, const float** m_in, hipStream_t st) {
    return wave_means_boundaries_f32(Bl, Br, n, d, A, offs, ws, ws_bytes, P, Lc, m_in, st);
}
// workspace of the time-partitioned wave solve (0: not partitioned / not covered)
size_t wave_btd_solve_ws(long Bl, long Br, long n, int d, int elem_size);
// mf_wave_inst.hip: the factorisations with one wavefront per series walking the time axis (mf_wave_ops.hpp); -101: not covered.
// Overloaded on the scalar type; defined for double and float.
template <typename T> int wave_btd_cholesky(long B, long n, int d, const T* diag, const T* sub, T* ldiag, T* lsub, void* ws, size_t ws_bytes,
                                            int* info, hipStream_t st);
template <typename T> int wave_btd_udl(long B, long n, int d, const T* diag, const T* sub, T* ut, T* chol_d, const T* eta, T* m_post,
                                       T* chol_dinv, void* ws, size_t ws_bytes, int* info, hipStream_t st);
// workspace of the time-partitioned posterior chain on the wave kernels (0: not partitioned / not covered)
size_t wave_udl_ws(long B, long n, int d, int elem_size);
template <typename T> int wave_btd_diag_of_inverse(long B, long n, int d, const T* ldiag, const T* lsub, T* odiag, T* osub, void* ws,
                                                   size_t ws_bytes, hipStream_t st);
// marginal means (omean | NULL), covariances and subsequent covariances (osub | NULL) of a chain of n time points
template <typename T> int wave_ssm_marginals(long B, long n, int d, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ,
                                             T* omean, T* ocov, T* osub, void* ws, size_t ws_bytes, hipStream_t st);
size_t wave_marg_ws(long B, long n, int d, int elem_size);
// the local step of the log-likelihood's gradient from the smoothed moments (mf_wave_grad.hpp); H == NULL: no observation terms
template <typename T>
int wave_kf_grad(long B, long Tn, int d, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H, const T* y,
                 const T* Rinv, int rinv_per_step, const T* mean, const T* cov, const T* cross, const T* w, T* g_mu0, T* g_cholP0, T* g_A,
                 T* g_b, T* g_cholQ, T* g_H, T* g_y, T* g_om, hipStream_t st);
}  // namespace mf
