// Large-state-dimension instantiations of the Kalman log-likelihood (mf_big.hpp: LDS tiles + MFMA), fp32 (d <= 64) and
// fp64 (d <= 32), and the entry points the C ABI dispatches to.
#include "mf_big.hpp"
#include "mf_launch.hpp"

namespace mf {

size_t big_kf_loglik_ws(long B, long Tn, int d, long chunks, int elem_size) {
    return elem_size == 4 ? big::kf_loglik_ws(B, Tn, d, chunks) : bigd::kf_loglik_ws(B, Tn, d, chunks);
}

int big_kf_loglik_f32(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                      const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step,
                      float add_const, float* out, void* ws, size_t ws_bytes, int* info, long chunks, hipEvent_t ev0,
                      hipEvent_t ev1, hipStream_t st) {
    return big::kf_loglik(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, add_const, out, ws, ws_bytes,
                          info, chunks, ev0, ev1, st);
}

int big_kf_loglik_f64(long B, long Tn, int d, int m, const double* mu0, const double* cholP0, const double* A,
                      const double* b, const double* cholQ, const double* H, const double* y, const double* Rinv,
                      int rinv_per_step, double add_const, double* out, void* ws, size_t ws_bytes, int* info, long chunks,
                      hipEvent_t ev0, hipEvent_t ev1, hipStream_t st) {
    return bigd::kf_loglik(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, add_const, out, ws, ws_bytes,
                           info, chunks, ev0, ev1, st);
}


#define MF_BIG_EXPORT(SUF, T, NS)                                                                                            \
    int big_cholesky_##SUF(long B, long n, int d, const T* diag, const T* sub, T* ldiag, T* lsub, void* ws, size_t ws_bytes,   \
                           int* info, hipStream_t st) {                                                                      \
        return NS::op_cholesky_par(B, n, d, diag, sub, ldiag, lsub, ws, ws_bytes, info, st);                                 \
    }                                                                                                                        \
    int big_solve_##SUF(long Bl, long Br, long n, int d, const T* ldiag, const T* lsub, const T* rhs, T* out, int transpose,  \
                        void* ws, size_t ws_bytes, hipStream_t st) {                                                         \
        return NS::op_solve_par(Bl, Br, n, d, ldiag, lsub, rhs, out, transpose, ws, ws_bytes, st);                           \
    }                                                                                                                        \
    int big_matvec_##SUF(long Bl, long Br, long n, int d, const T* diag, const T* sub, const T* x, T* out, int mode,          \
                         hipStream_t st) {                                                                                   \
        return NS::op_matvec(Bl, Br, n, d, diag, sub, x, out, mode, st);                                                     \
    }                                                                                                                        \
    int big_logdet_##SUF(long B, long n, int d, const T* ldiag, T* out, hipStream_t st) {                                    \
        return NS::op_logdet(B, n, d, ldiag, out, st);                                                                       \
    }                                                                                                                        \
    int big_diag_of_inverse_##SUF(long B, long n, int d, const T* ldiag, const T* lsub, T* odiag, T* osub, void* ws,          \
                                  size_t ws_bytes, hipStream_t st) {                                                         \
        return NS::op_diag_of_inverse_par(B, n, d, ldiag, lsub, odiag, osub, ws, ws_bytes, st);                              \
    }                                                                                                                        \
    int big_udl_##SUF(long B, long n, int d, const T* diag, const T* sub, T* ut, T* chol_d, const T* eta, T* m_post,          \
                      T* chol_dinv, void* ws, size_t ws_bytes, int* info, hipStream_t st) {                                  \
        return NS::op_udl_par(B, n, d, diag, sub, ut, chol_d, eta, m_post, chol_dinv, ws, ws_bytes, info, st);               \
    }                                                                                                                        \
    int big_ssm_precision_##SUF(long B, long Tn, int d, int m, const T* mu0, const T* cholP0, const T* A, const T* b,         \
                                const T* cholQ, const T* H, const T* y, const T* Rinv, int rinv_per_step, T* diag, T* sub,    \
                                T* eta, hipStream_t st) {                                                                    \
        return NS::op_ssm_precision(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, diag, sub, eta, st);   \
    }                                                                                                                        \
    int big_means_##SUF(long Bl, long Br, long Tn, int d, const T* A, const T* offs, T* out, void* ws, size_t ws_bytes,       \
                        hipStream_t st) {                                                                                    \
        return NS::op_means_par(Bl, Br, Tn, d, A, offs, out, ws, ws_bytes, st);                                              \
    }                                                                                                                        \
    int big_block_matmul_##SUF(long B, long n, int d, const T* X, long xs, const T* Y, long ys, T* out, hipStream_t st) {     \
        return NS::op_block_matmul(B, n, d, X, xs, Y, ys, out, st);                                                          \
    }                                                                                                                        \
    int big_kf_grad_##SUF(long B, long Tn, int d, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, \
                          const T* H, const T* y, const T* Rinv, int rinv_per_step, const T* mean, const T* cov,             \
                          const T* cross, const T* w, T* g_mu0, T* g_cholP0, T* g_A, T* g_b, T* g_cholQ, T* g_H, T* g_y,     \
                          T* g_om, hipStream_t st) {                                                                         \
        return NS::op_kf_grad(NS::BigGradArgs{B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, mean, cov,   \
                                              cross, w, g_mu0, g_cholP0, g_A, g_b, g_cholQ, g_H, g_y, g_om}, st);            \
    }                                                                                                                        \
    int big_marginal_covs_##SUF(long B, long n, int d, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ,  \
                                T* omean, T* ocov, T* osub, void* ws, size_t ws_bytes, hipStream_t st) {                     \
        return NS::op_marginal_covs(B, n, d, mu0, cholP0, A, b, cholQ, omean, ocov, osub, ws, ws_bytes, st);                 \
    }
size_t big_marginal_covs_ws(long B, long n, int d, int elem_size) {
    return elem_size == 4 ? big::marginal_covs_ws(B, n, d) : bigd::marginal_covs_ws(B, n, d);
}
// workspace of the time-partitioned factorisations (mf_bigpar_impl.hpp); chain: the posterior chain's right-hand-side maps too
size_t big_btd_par_ws(long B, long n, int d, int chain, int elem_size) {
    return elem_size == 4 ? big::bigpar_ws_any(B, n, d, chain != 0) : bigd::bigpar_ws_any(B, n, d, chain != 0);
}
size_t big_btd_solve_ws(long Bl, long Br, long n, int d, int elem_size) {
    const size_t tile = elem_size == 4 ? big::bigpar_solve_ws(Bl, Br, n, d) : bigd::bigpar_solve_ws(Bl, Br, n, d);
    const size_t wave = wave_btd_solve_ws(Bl, Br, n, d, elem_size);
    return tile > wave ? tile : wave;
}
size_t big_btd_tak_ws(long B, long n, int d, int elem_size) {
    return elem_size == 4 ? big::bigpar_tak_ws(B, n, d) : bigd::bigpar_tak_ws(B, n, d);
}
MF_BIG_EXPORT(f32, float, big)
MF_BIG_EXPORT(f64, double, bigd)
#undef MF_BIG_EXPORT

}  // namespace mf
