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

}  // namespace mf
