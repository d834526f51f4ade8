// extern "C" entry points of include/markovflow_amd.h: argument checks + dispatch on the state dimension.
#include "../../include/markovflow_amd.h"
#include "mf_launch.hpp"
#include "mf_wave_api.hpp"

#include <cstdlib>

namespace {

template <typename T> const mf::OpsTable<T>* table_for(int d);
#define MF_CASE(D) case D: return mf::ops_f32_d##D();
template <> const mf::OpsTable<float>* table_for<float>(int d) {
    switch (d) {
        MF_CASE(1) MF_CASE(2) MF_CASE(3) MF_CASE(4) MF_CASE(5) MF_CASE(6) MF_CASE(7) MF_CASE(8) MF_CASE(9)
        MF_CASE(10) MF_CASE(11) MF_CASE(12) MF_CASE(13) MF_CASE(14) MF_CASE(15)
        default: return nullptr;
    }
}
#undef MF_CASE
#define MF_CASE(D) case D: return mf::ops_f64_d##D();
template <> const mf::OpsTable<double>* table_for<double>(int d) {
    switch (d) {
        MF_CASE(1) MF_CASE(2) MF_CASE(3) MF_CASE(4) MF_CASE(5) MF_CASE(6) MF_CASE(7) MF_CASE(8) MF_CASE(9)
        MF_CASE(10) MF_CASE(11) MF_CASE(12) MF_CASE(13) MF_CASE(14) MF_CASE(15)
        default: return nullptr;
    }
}
#undef MF_CASE

template <typename T> const mf::PostOps<T>* post_table_for(int d);
#define MF_CASE(D) case D: return mf::post_ops_f32_d##D();
template <> const mf::PostOps<float>* post_table_for<float>(int d) {
    switch (d) {
        MF_CASE(1) MF_CASE(2) MF_CASE(3) MF_CASE(4) MF_CASE(5) MF_CASE(6)
        default: return nullptr;
    }
}
#undef MF_CASE
#define MF_CASE(D) case D: return mf::post_ops_f64_d##D();
template <> const mf::PostOps<double>* post_table_for<double>(int d) {
    switch (d) {
        MF_CASE(1) MF_CASE(2) MF_CASE(3) MF_CASE(4) MF_CASE(5) MF_CASE(6)
        default: return nullptr;
    }
}
#undef MF_CASE

template <typename T> const mf::GradOps<T>* grad_table_for(int d);
#define MF_CASE(D) case D: return mf::grad_ops_f32_d##D();
template <> const mf::GradOps<float>* grad_table_for<float>(int d) {
    switch (d) {
        MF_CASE(1) MF_CASE(2) MF_CASE(3) MF_CASE(4) MF_CASE(5) MF_CASE(6)
        default: return nullptr;
    }
}
#undef MF_CASE
#define MF_CASE(D) case D: return mf::grad_ops_f64_d##D();
template <> const mf::GradOps<double>* grad_table_for<double>(int d) {
    switch (d) {
        MF_CASE(1) MF_CASE(2) MF_CASE(3) MF_CASE(4) MF_CASE(5) MF_CASE(6)
        default: return nullptr;
    }
}
#undef MF_CASE

inline hipStream_t S(void* s) { return static_cast<hipStream_t>(s); }

template <typename T>
int kf_loglik(int64_t B, int64_t Tn, int d, int m, const T* mu0, const T* cholP0, const T* A, const T* b,
              const T* cholQ, const T* H, const T* y, const T* Rinv, int rinv_per_step, T add_const, T* out, void* ws,
              size_t ws_bytes, int* info, int64_t chunks, void* ev0, void* ev1, void* stream) {
    if (B < 0) return -1;
    if (Tn < 1) return -2;
    if (d < 1) return -3;
    const auto* t = table_for<T>(d);
    // experiment knob: state dimensions >= MF_BIG_FROM take the LDS-tile / MFMA path even where a register-resident
    // instantiation exists
    static const int big_from = [] { const char* e = mf::mf_knob("MF_BIG_FROM"); return e ? std::atoi(e) : 1000; }();
    // (more than four outputs: the tile engine at ANY state dimension - its tiles pad d to 16 - so that the observation
    // dimension is not capped at the register kernels' four)
    const bool big = (d > mf::MF_MAX_D || d >= big_from || m > 4) && d <= (sizeof(T) == 4 ? mf::MF_MAX_D_BIG : mf::MF_MAX_D_LOGLIK_F64);
    if (!t && !big) return -100;
    if (m < 1 || m > (big ? 32 : 4)) return -4;
    if (B == 0) return 0;
    if (!mu0) return -5;
    if (!cholP0) return -6;
    if (Tn > 1 && (!A || !b || !cholQ)) return -7;
    if (!H) return -10;
    if (!y) return -11;
    if (!Rinv) return -12;
    if (!out) return -15;
    if (t && d < big_from && m <= (d > mf::MF_MAX_D ? 8 : 4)) {
        // d <= 9: every plan has a kernel.  10 <= d <= 15: the row kernels when the plan is theirs (at most four outputs, offsets
        // within a buffer descriptor), else -100 and the tile engine below takes the call
        const int rc = t->kf_loglik(B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, add_const, out, ws, ws_bytes,
                                    info, chunks, static_cast<hipEvent_t>(ev0), static_cast<hipEvent_t>(ev1), S(stream));
        if (rc != -100 || !big) return rc;
    }
    if constexpr (sizeof(T) == 4)
        return mf::big_kf_loglik_f32(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, add_const, out,
                                     ws, ws_bytes, info, chunks, static_cast<hipEvent_t>(ev0),
                                     static_cast<hipEvent_t>(ev1), S(stream));
    else
        return mf::big_kf_loglik_f64(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, add_const, out,
                                     ws, ws_bytes, info, chunks, static_cast<hipEvent_t>(ev0),
                                     static_cast<hipEvent_t>(ev1), S(stream));
}


// sum of the per-series values + B x constant terms (kalman_filter.py:229-231,249-255): one workgroup
template <typename T>
__global__ void __launch_bounds__(256) loglik_total_kernel(long B, const T* __restrict__ per_series, int m,
                                                           const T* __restrict__ chol_obs, long num_points,
                                                           const T* __restrict__ extra, T host_const, T* __restrict__ out) {
    __shared__ double part[4];
    double acc = 0.0;
    for (long s = threadIdx.x; s < B; s += 256) acc += (double)per_series[s];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = (double)host_const;
        if (chol_obs) {
            double ld = 0.0;
            for (int i = 0; i < m; ++i) ld += log(fabs((double)chol_obs[i * m + i]));
            c -= (double)num_points * ld;
        }
        if (extra) c += (double)extra[0];
        out[0] = (T)(part[0] + part[1] + part[2] + part[3] + (double)B * c);
    }
}

template <typename T>
int loglik_total(int64_t B, const T* per_series, int m, const T* chol_obs, int64_t num_points, const T* extra, T host_const,
                 T* out, void* stream) {
    if (B < 0) return -1;
    if (B > 0 && !per_series) return -2;
    if (chol_obs && m < 1) return -3;
    if (!out) return -8;
    hipLaunchKernelGGL((loglik_total_kernel<T>), dim3(1), dim3(256), 0, S(stream), (long)B, per_series, m, chol_obs,
                       (long)num_points, extra, host_const, out);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// the fused GPR routes share one body: `m` outputs, `multi` = one output per component
template <typename T>
int gpr_matern_loglik(int64_t B, int64_t Tn, int ncomp, const int* orders, const T* lam, const T* var, int per_series,
                      const T* t_pts, const T* y, int m, int multi, const T* rinv, T jitter, T add_const, T* out, void* ws,
                      size_t ws_bytes, int* info, int64_t chunks, void* prof_start, void* prof_stop, void* stream) {
    if (B < 0) return -1;
    if (Tn < 1) return -2;
    if (ncomp < 1 || !orders) return -3;
    int d = 0;
    for (int c = 0; c < ncomp; ++c) {
        if (orders[c] != 1 && orders[c] != 3 && orders[c] != 5) return -4;
        d += (orders[c] + 1) / 2;
    }
    const auto* t = table_for<T>(d);
    if (!t) return -101;
    if (m < 1 || m > (d > mf::MF_MAX_D ? 8 : 4)) return -101;
    if (B == 0) return 0;
    if (!lam) return -5;
    if (!var) return -6;
    if (!t_pts) return -8;
    if (!y) return -9;
    if (!rinv) return -10;
    if (!out) return -13;
    return t->gpr_loglik(B, Tn, ncomp, orders, lam, var, per_series, t_pts, y, m, multi, rinv, jitter, add_const, out, ws,
                         ws_bytes, info, chunks, static_cast<hipEvent_t>(prof_start), static_cast<hipEvent_t>(prof_stop),
                         S(stream));
}

}  // namespace

extern "C" {

int mf_kf_loglik_total_f64(int64_t B, const double* per_series, int m, const double* chol_obs, int64_t num_points,
                           const double* extra_const, double host_const, double* out, void* stream) {
    return loglik_total<double>(B, per_series, m, chol_obs, num_points, extra_const, host_const, out, stream);
}
int mf_kf_loglik_total_f32(int64_t B, const float* per_series, int m, const float* chol_obs, int64_t num_points,
                           const float* extra_const, float host_const, float* out, void* stream) {
    return loglik_total<float>(B, per_series, m, chol_obs, num_points, extra_const, host_const, out, stream);
}

int mf_version(void) { return 8; }
// The caller's copy of an `info` word into its pinned host mirror, queued behind whatever `stream` holds (see the header): one
// hipMemcpyAsync - the Python layer used to spend ~20 us per factorising call on the same copy through torch.
int mf_info_mirror(int* host_mirror, const int* info, void* stream) {
    if (!host_mirror) return -1;
    if (!info) return -2;
    return hipMemcpyAsync(host_mirror, info, sizeof(int), hipMemcpyDeviceToHost, S(stream)) == hipSuccess ? 0 : -1000;
}
// flat index (series x blocks per series + block) an `info` word names; -1: no failure, or a failure whose block is unknown (word 1)
int64_t mf_info_flat_index(int info_word) { return info_word >= 2 ? (int64_t)(0x7fffffff - info_word) : -1; }   // (mf_small.hpp: MF_INFO_TOP)
int mf_max_state_dim(void) { return mf::MF_MAX_D; }

size_t mf_kf_loglik_workspace_bytes(int64_t B, int64_t T, int d, int elem_size, int64_t chunks) {
    if (B < 1 || T < 1 || d < 1) return 0;
    if (elem_size == 4) {
        const auto* t = table_for<float>(d);
        const size_t small = t ? t->kf_loglik_ws(B, T, chunks) : 0;
        size_t large = (d <= mf::MF_MAX_D_BIG) ? mf::big_kf_loglik_ws(B, T, d, chunks, 4) : 0;   // (d <= 9 too: more than four outputs)
        return small > large ? small : large;
    }
    const auto* t = table_for<double>(d);
    const size_t small = t ? t->kf_loglik_ws(B, T, chunks) : 0;
    size_t large = (d <= mf::MF_MAX_D_LOGLIK_F64) ? mf::big_kf_loglik_ws(B, T, d, chunks, 8) : 0;
    return small > large ? small : large;
}
int mf_row_operators_cover(int64_t B, int64_t T, int d, int elem_size) {
    if (B < 1 || T < 1 || d < 1 || d > mf::MF_MAX_D_ROW) return 0;
    if (d <= mf::MF_MAX_D) return 1;
    // 10 <= d <= 15 are compiled with the row kernels only, and the operators run in row form where they are partitioned in time:
    // exactly the shapes for which the factorisation asks for a workspace
    if (elem_size == 4) { const auto* t = table_for<float>(d); return t && t->btd_cholesky_ws(B, T) > 0 ? 1 : 0; }
    const auto* t = table_for<double>(d);
    return t && t->btd_cholesky_ws(B, T) > 0 ? 1 : 0;
}
int mf_max_state_dim_f32_loglik(void) { return mf::MF_MAX_D_BIG; }
int mf_max_state_dim_f64_loglik(void) { return mf::MF_MAX_D_LOGLIK_F64; }
int mf_max_state_dim_f64_tile_ops(void) { return mf::MF_MAX_D_BIG_F64; }

int mf_kf_loglik_f64(int64_t B, int64_t T, int d, int m, const double* mu0, const double* cholP0, const double* A,
                     const double* b, const double* cholQ, const double* H, const double* y, const double* Rinv,
                     int rinv_per_step, double add_const, double* out, void* ws, size_t ws_bytes, int* info,
                     int64_t chunks, void* prof_start, void* prof_stop, void* stream) {
    return kf_loglik<double>(B, T, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, add_const, out, ws,
                             ws_bytes, info, chunks, prof_start, prof_stop, stream);
}
int mf_kf_loglik_f32(int64_t B, int64_t T, int d, int m, const float* mu0, const float* cholP0, const float* A,
                     const float* b, const float* cholQ, const float* H, const float* y, const float* Rinv,
                     int rinv_per_step, float add_const, float* out, void* ws, size_t ws_bytes, int* info,
                     int64_t chunks, void* prof_start, void* prof_stop, void* stream) {
    return kf_loglik<float>(B, T, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, add_const, out, ws,
                            ws_bytes, info, chunks, prof_start, prof_stop, stream);
}

// `t` = table of the register / row kernels (d <= 9: all of them; 10 <= d <= 15: row kernels only) or NULL;
// `big` = the LDS-tile / MFMA kernels cover this (d, type)
#define MF_HEAD(T, B, Tn, d)                                                                        \
    if ((B) < 0) return -1;                                                                         \
    if ((Tn) < 1) return -2;                                                                        \
    if ((d) < 1) return -3;                                                                         \
    const auto* t = table_for<T>(d);                                                                \
    const bool big = (d) > mf::MF_MAX_D && (d) <= (sizeof(T) == 4 ? mf::MF_MAX_D_BIG : mf::MF_MAX_D_BIG_F64); \
    if (!t && !big) return -100;                                                                    \
    if ((B) == 0) return 0;
// the table's kernels first; for 10 <= d <= 15 (row kernels only) -100 = "this plan is not theirs": on to the tile engine
#define MF_TRY(call)                                          \
    if (t) {                                                  \
        const int rc_ = (call);                               \
        if (rc_ != -100 || !big) return rc_;                  \
    }

#define MF_DEFINE(SUF, T)                                                                                              \
    int mf_btd_cholesky_##SUF(int64_t B, int64_t Tn, int d, const T* diag, const T* sub, T* ldiag, T* lsub, void* ws,  \
                              size_t ws_bytes, int* info, void* stream) {                                              \
        MF_HEAD(T, B, Tn, d)                                                                                           \
        if (!diag) return -4;                                                                                          \
        if (!ldiag) return -6;                                                                                         \
        if (sub && !lsub) return -7;                                                                                   \
        if (Tn == 1) sub = nullptr;                                                                                    \
        MF_TRY(t->btd_cholesky(B, Tn, diag, sub, ldiag, lsub, ws, ws_bytes, info, S(stream))) \
        if (big) return mf::big_cholesky_##SUF(B, Tn, d, diag, sub, ldiag, lsub, ws, ws_bytes, info, S(stream));      \
        return -100;                          \
    }                                                                                                                  \
    int mf_btd_solve_##SUF(int64_t Bl, int64_t Br, int64_t Tn, int d, const T* ldiag, const T* lsub, const T* rhs,     \
                           T* out, int transpose, void* ws, size_t ws_bytes, void* stream) {                           \
        MF_HEAD(T, Br, Tn, d)                                                                                          \
        if (Bl < 1 || Br % Bl != 0) return -1;                                                                         \
        if (!ldiag) return -5;                                                                                         \
        if (!rhs) return -7;                                                                                           \
        if (!out) return -8;                                                                                           \
        if (Tn == 1) lsub = nullptr;                                                                                   \
        MF_TRY(t->btd_solve(Bl, Br, Tn, ldiag, lsub, rhs, out, transpose, ws, ws_bytes, S(stream))) \
        if (big) return mf::big_solve_##SUF(Bl, Br, Tn, d, ldiag, lsub, rhs, out, transpose, ws, ws_bytes, S(stream)); \
        return -100;                    \
    }                                                                                                                  \
    int mf_btd_matvec_##SUF(int64_t Bl, int64_t Br, int64_t Tn, int d, const T* diag, const T* sub, const T* x,        \
                            T* out, int mode, void* stream) {                                                          \
        MF_HEAD(T, Br, Tn, d)                                                                                          \
        if (Bl < 1 || Br % Bl != 0) return -1;                                                                         \
        if (!diag) return -5;                                                                                          \
        if (!x) return -7;                                                                                             \
        if (!out) return -8;                                                                                           \
        if (mode < 0 || mode > 2) return -9;                                                                           \
        if (Tn == 1) sub = nullptr;                                                                                    \
        MF_TRY(t->btd_matvec(Bl, Br, Tn, diag, sub, x, out, mode, S(stream))) \
        if (big) return mf::big_matvec_##SUF(Bl, Br, Tn, d, diag, sub, x, out, mode, S(stream));                       \
        return -100;                                          \
    }                                                                                                                  \
    int mf_btd_logdet_##SUF(int64_t B, int64_t Tn, int d, const T* ldiag, T* out, void* stream) {                      \
        MF_HEAD(T, B, Tn, d)                                                                                           \
        if (!ldiag) return -4;                                                                                         \
        if (!out) return -5;                                                                                           \
        MF_TRY(t->btd_logdet(B, Tn, ldiag, out, S(stream))) \
        if (big) return mf::big_logdet_##SUF(B, Tn, d, ldiag, out, S(stream));                                         \
        return -100;                                                            \
    }                                                                                                                  \
    int mf_btd_logdet_quad_##SUF(int64_t B, int64_t Tn, int d, const T* diag, const T* sub, const T* rhs, T* out,      \
                                 void* ws, size_t ws_bytes, int* info, void* stream) {                                 \
        MF_HEAD(T, B, Tn, d)                                                                                           \
        if (!diag) return -4;                                                                                          \
        if (Tn > 1 && !sub) return -5;                                                                                 \
        if (!rhs) return -6;                                                                                           \
        if (!out) return -7;                                                                                           \
        if (!t) return -100;                                                                                            \
        return t->btd_logdet_quad(B, Tn, diag, sub, rhs, out, ws, ws_bytes, info, 0, S(stream));                       \
    }                                                                                                                  \
    int mf_btd_diag_of_inverse_##SUF(int64_t B, int64_t Tn, int d, const T* ldiag, const T* lsub, T* odiag, T* osub,   \
                                     void* ws, size_t ws_bytes, void* stream) {                                        \
        MF_HEAD(T, B, Tn, d)                                                                                           \
        if (!ldiag) return -4;                                                                                         \
        if (!odiag) return -6;                                                                                         \
        if (Tn == 1) lsub = nullptr;                                                                                   \
        if (!lsub) osub = nullptr;                                                                                     \
        MF_TRY(t->btd_diag_of_inverse(B, Tn, ldiag, lsub, odiag, osub, ws, ws_bytes, S(stream))) \
        if (big) return mf::big_diag_of_inverse_##SUF(B, Tn, d, ldiag, lsub, odiag, osub, ws, ws_bytes, S(stream));   \
        return -100;                       \
    }                                                                                                                  \
    int mf_ssm_marginal_covariances_##SUF(int64_t B, int64_t Tn, int d, const T* cholP0, const T* A, const T* cholQ,   \
                                          T* out_cov, T* out_sub, void* ws, size_t ws_bytes, void* stream) {          \
        MF_HEAD(T, B, Tn, d)                                                                                           \
        if (Tn < 2) return -2;                                                                                         \
        if (!cholP0) return -4;                                                                                        \
        if (!A) return -5;                                                                                             \
        if (!cholQ) return -6;                                                                                         \
        if (!out_cov) return -7;                                                                                       \
        MF_TRY(t->ssm_marginal_covs(B, Tn, cholP0, A, cholQ, out_cov, out_sub, ws, ws_bytes, S(stream))) \
        if (big) return mf::big_marginal_covs_##SUF(B, Tn, d, nullptr, cholP0, A, nullptr, cholQ, nullptr, out_cov, out_sub, \
                                                    ws, ws_bytes, S(stream));                                          \
        return -100;               \
    }                                                                                                                  \
    int mf_ssm_marginals_##SUF(int64_t B, int64_t Tn, int d, const T* mu0, const T* cholP0, const T* A, const T* b,    \
                               const T* cholQ, T* out_mean, T* out_cov, T* out_sub, void* ws, size_t ws_bytes,        \
                               void* stream) {                                                                        \
        MF_HEAD(T, B, Tn, d)                                                                                           \
        if (Tn < 2) return -101;                                                                                       \
        if (!mu0) return -4;                                                                                           \
        if (!cholP0) return -5;                                                                                        \
        if (!A) return -6;                                                                                             \
        if (!b) return -7;                                                                                             \
        if (!cholQ) return -8;                                                                                         \
        if (!out_mean) return -9;                                                                                      \
        if (!out_cov) return -10;                                                                                      \
        MF_TRY(t->ssm_marginals(B, Tn, mu0, cholP0, A, b, cholQ, out_mean, out_cov, out_sub, ws, ws_bytes, S(stream))) \
        if (big) return mf::big_marginal_covs_##SUF(B, Tn, d, mu0, cholP0, A, b, cholQ, out_mean, out_cov, out_sub, ws, \
                                                    ws_bytes, S(stream));                                              \
        return -100; \
    }                                                                                                                  \
    int mf_btd_udl_##SUF(int64_t B, int64_t Tn, int d, const T* diag, const T* sub, T* ut, T* chol_d, const T* eta,    \
                         T* m_post, T* chol_dinv, int chain_layout, void* ws, size_t ws_bytes, int* info,              \
                         void* stream) {                                                                               \
        MF_HEAD(T, B, Tn, d)                                                                                           \
        if (!diag) return -4;                                                                                          \
        if (Tn > 1 && (!sub || !ut)) return -5;                                                                        \
        if (!chol_d && !chain_layout) return -7;                                                                       \
        if (eta && (!m_post || !chol_dinv)) return -9;                                                                 \
        if (chain_layout && (!eta || Tn < 2)) return -11;                                                              \
        MF_TRY(t->btd_udl(B, Tn, diag, sub, ut, chol_d, eta, m_post, chol_dinv, chain_layout, ws, ws_bytes, info, S(stream))) \
        if (big && chain_layout) return -101;                                                                          \
        if (big) return mf::big_udl_##SUF(B, Tn, d, diag, sub, ut, chol_d, eta, m_post, chol_dinv, ws, ws_bytes, info, \
                                       S(stream));                                                                 \
        return -100;                                                                                  \
    }                                                                                                                  \
    int mf_ssm_precision_##SUF(int64_t B, int64_t Tn, int d, int m, const T* mu0, const T* cholP0, const T* A,         \
                               const T* b, const T* cholQ, const T* H, const T* y, const T* Rinv, int rinv_per_step,   \
                               T* diag, T* sub, T* eta, void* stream) {                                                \
        MF_HEAD(T, B, Tn, d)                                                                                           \
        if (!cholP0) return -6;                                                                                        \
        if (Tn > 1 && (!A || !cholQ || !sub)) return -7;                                                               \
        if (H && !Rinv) return -12;                                                                                    \
        if (H && (m < 1 || m > 32)) return -4;                                                                         \
        if (!diag) return -14;                                                                                         \
        if (eta && (!mu0 || (Tn > 1 && !b))) return -5;                                                                \
        if (H && m > 4)      /* more than four outputs: the tile engine at any d */                                    \
            return mf::big_ssm_precision_##SUF(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step,       \
                                               diag, sub, eta, S(stream));                                             \
        MF_TRY(t->ssm_precision(B, Tn, H ? m : 1, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, diag, sub, eta, S(stream))) \
        if (big)                                                                                                       \
            return mf::big_ssm_precision_##SUF(B, Tn, d, H ? m : 1, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step,   \
                                               diag, sub, eta, S(stream));                                             \
        return -100;                                                                            \
    }                                                                                                                  \
    int mf_ssm_marginal_means_##SUF(int64_t Bl, int64_t Br, int64_t Tn, int d, const T* A, const T* offs, T* out,      \
                                    void* ws, size_t ws_bytes, void* stream) {                                         \
        MF_HEAD(T, Br, Tn, d)                                                                                          \
        if (Bl < 1 || Br % Bl != 0) return -1;                                                                         \
        if (Tn > 1 && !A) return -5;                                                                                   \
        if (!offs) return -6;                                                                                          \
        if (!out) return -7;                                                                                           \
        MF_TRY(t->ssm_means(Bl, Br, Tn, A, offs, out, ws, ws_bytes, S(stream))) \
        if (big) return mf::big_means_##SUF(Bl, Br, Tn, d, A, offs, out, ws, ws_bytes, S(stream));                     \
        return -100;                                        \
    }

#define MF_DEFINE2(SUF, T)                                                                                             \
    int mf_block_matmul_##SUF(int64_t B, int64_t n, int d, const T* X, int64_t x_stride, const T* Y, int64_t y_stride,  \
                              T* out, void* stream) {                                                                  \
        MF_HEAD(T, B, n, d)                                                                                            \
        if (!X) return -4;                                                                                             \
        if (x_stride < n) return -5;                                                                                   \
        if (!Y) return -6;                                                                                             \
        if (y_stride < n) return -7;                                                                                   \
        if (!out) return -8;                                                                                           \
        MF_TRY(t->block_matmul(B, n, X, x_stride, Y, y_stride, out, S(stream))) \
        if (big) return mf::big_block_matmul_##SUF(B, n, d, X, x_stride, Y, y_stride, out, S(stream));                 \
        return -100;                                        \
    }

#define MF_DEFINE3(SUF, T)                                                                                             \
    int mf_gpr_matern_loglik_##SUF(int64_t B, int64_t Tn, int ncomp, const int* orders, const T* lam, const T* var,    \
                                   int per_series, const T* t_pts, const T* y, const T* rinv, T jitter, T add_const,   \
                                   T* out, void* ws, size_t ws_bytes, int* info, int64_t chunks, void* prof_start,     \
                                   void* prof_stop, void* stream) {                                                    \
        return gpr_matern_loglik<T>(B, Tn, ncomp, orders, lam, var, per_series, t_pts, y, 1, 0, rinv, jitter, add_const, \
                                    out, ws, ws_bytes, info, chunks, prof_start, prof_stop, stream);                   \
    }                                                                                                                  \
    int mf_gpr_matern_multi_loglik_##SUF(int64_t B, int64_t Tn, int ncomp, const int* orders, const T* lam,            \
                                         const T* var, int per_series, const T* t_pts, const T* y, int m,              \
                                         const T* rinv, T jitter, T add_const, T* out, void* ws, size_t ws_bytes,      \
                                         int* info, int64_t chunks, void* prof_start, void* prof_stop, void* stream) { \
        if (m != ncomp) return -10;                                                                                    \
        return gpr_matern_loglik<T>(B, Tn, ncomp, orders, lam, var, per_series, t_pts, y, m, 1, rinv, jitter, add_const, \
                                    out, ws, ws_bytes, info, chunks, prof_start, prof_stop, stream);                   \
    }

#define MF_DEFINE4(SUF, T)                                                                                             \
    int mf_sde_conditional_predict_##SUF(int64_t B, int64_t N, int64_t Np, int d, const int64_t* idx, const T* A_mt,   \
                                         const T* Q_mt, const T* A_tp, const T* Q_tp, const T* means, const T* covs,   \
                                         const T* subsequent_covs, const T* prior_mean, const T* prior_cov,            \
                                         T* out_mean, T* out_cov, int* info, void* stream) {                           \
        if (B < 0) return -1;                                                                                          \
        if (N < 1) return -2;                                                                                          \
        if (Np < 0) return -3;                                                                                         \
        if (d < 1) return -4;                                                                                          \
        const auto* t = table_for<T>(d);                                                                               \
        if (!t) return -100;                                                                                           \
        if (B == 0 || Np == 0) return 0;                                                                               \
        if (!idx) return -5;                                                                                           \
        if (!A_mt || !Q_mt || !A_tp || !Q_tp) return -6;                                                               \
        if (!means) return -10;                                                                                        \
        if (out_cov && (!covs || (N > 1 && !subsequent_covs) || !prior_cov)) return -11;                               \
        if (!prior_mean) return -13;                                                                                   \
        if (!out_mean) return -15;                                                                                     \
        return t->sde_predict(B, N, Np, reinterpret_cast<const long long*>(idx), A_mt, Q_mt, A_tp, Q_tp, means, covs,  \
                              subsequent_covs, prior_mean, prior_cov, out_mean, out_cov, info, S(stream));             \
    }

#define MF_DEFINE4B(SUF, T)                                                                                            \
    int mf_sde_conditional_statistics_##SUF(int64_t n, int d, const T* A_mt, const T* Q_mt, const T* A_tp, const T* Q_tp, \
                                            T* projections, T* covariances, int* info, void* stream) {                \
        if (n < 0) return -1;                                                                                          \
        if (d < 1) return -2;                                                                                          \
        const auto* t = table_for<T>(d);                                                                               \
        if (!t) return -100;                                                                                           \
        if (n == 0) return 0;                                                                                          \
        if (!A_mt || !Q_mt || !A_tp || !Q_tp) return -3;                                                               \
        if (!projections) return -7;                                                                                   \
        if (!covariances) return -8;                                                                                   \
        return t->sde_cond_stats(n, A_mt, Q_mt, A_tp, Q_tp, projections, covariances, info, S(stream));                \
    }                                                                                                                  \
    int mf_btd_cholesky_grad_##SUF(int64_t B, int64_t n, int d, const T* ldiag, const T* lsub, const T* g_ldiag,       \
                                   const T* g_lsub, T* g_diag, T* g_sub, void* ws, size_t ws_bytes, void* stream) {    \
        if (B < 0) return -1;                                                                                          \
        if (n < 1) return -2;                                                                                          \
        if (d < 1) return -3;                                                                                          \
        const auto* t = table_for<T>(d);                                                                               \
        if (!t && !mf::adj_covers(d)) return -100;                                                                     \
        if (B == 0) return 0;                                                                                          \
        if (!ldiag) return -4;                                                                                         \
        if (!g_diag) return -8;                                                                                        \
        if (lsub && n > 1 && !g_sub) return -9;                                                                        \
        if (t) {                                                                                                       \
            const int rc = t->btd_cholesky_grad(B, n, ldiag, lsub, g_ldiag, g_lsub, g_diag, g_sub, ws, ws_bytes, S(stream)); \
            if (rc != -100 || !mf::adj_covers(d)) return rc;                                                           \
        }                                                                                                              \
        /* 10 <= d <= 32: one workgroup per series walks the chain (mf_adj.hip) */                                     \
        return mf::adj_cholesky_grad<T>(B, n, d, ldiag, lsub, g_ldiag, g_lsub, g_diag, g_sub, ws, ws_bytes, S(stream)); \
    }                                                                                                                  \
    int mf_btd_diag_of_inverse_grad_##SUF(int64_t B, int64_t n, int d, const T* ldiag, const T* lsub, const T* sigma,  \
                                          const T* g_diag, const T* g_sub, T* g_ldiag, T* g_lsub, void* ws,            \
                                          size_t ws_bytes, void* stream) {                                             \
        if (B < 0) return -1;                                                                                          \
        if (n < 1) return -2;                                                                                          \
        if (d < 1) return -3;                                                                                          \
        const auto* t = table_for<T>(d);                                                                               \
        if (!t && !mf::adj_covers(d)) return -100;                                                                     \
        if (B == 0) return 0;                                                                                          \
        if (!ldiag) return -4;                                                                                         \
        if (lsub && n > 1 && !sigma) return -6;                                                                        \
        if (!g_ldiag) return -9;                                                                                       \
        if (lsub && n > 1 && !g_lsub) return -10;                                                                      \
        if (t) {                                                                                                       \
            const int rc = t->btd_diag_of_inverse_grad(B, n, ldiag, lsub, sigma, g_diag, g_sub, g_ldiag, g_lsub, ws, ws_bytes, \
                                                       S(stream));                                                     \
            if (rc != -100 || !mf::adj_covers(d)) return rc;                                                           \
        }                                                                                                              \
        return mf::adj_diag_of_inverse_grad<T>(B, n, d, ldiag, lsub, sigma, g_diag, g_sub, g_ldiag, g_lsub, ws, ws_bytes, S(stream)); \
    }

#define MF_DEFINE5(SUF, T)                                                                                             \
    int mf_kf_loglik_grad_##SUF(int64_t B, int64_t Tn, int d, int m, const T* mu0, const T* cholP0, const T* A,        \
                                const T* b, const T* cholQ, const T* H, const T* y, const T* Rinv, int rinv_per_step,  \
                                const T* post_mean, const T* post_cov, const T* post_cross, T* g_mu0, T* g_cholP0,     \
                                T* g_A, T* g_b, T* g_cholQ, T* g_H, T* g_y, T* g_omega, const T* weights, int* info,   \
                                void* stream) {                                                                        \
        MF_HEAD(T, B, Tn, d)                                                                                           \
        if (m < 1 || m > 32) return -4;                                                                                \
        if (!mu0 || !cholP0 || (Tn > 1 && (!A || !b || !cholQ))) return -5;                                            \
        if (H && (!y || !Rinv)) return -11;                                                                            \
        if (!post_mean || !post_cov || (Tn > 1 && !post_cross)) return -14;                                            \
        if (!g_mu0 || !g_cholP0 || (Tn > 1 && (!g_A || !g_b || !g_cholQ))) return -17;                                 \
        if (H && (!g_H || !g_y || !g_omega)) return -22;                                                               \
        if ((big || m > 4) && !weights) return -25;                                                                    \
        if (H && m > 4)      /* more than four outputs: the tile kernel at any d (m <= d rounded up to 16) */           \
            return mf::big_kf_grad_##SUF(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step,             \
                                         post_mean, post_cov, post_cross, weights, g_mu0, g_cholP0, g_A, g_b,          \
                                         g_cholQ, g_H, g_y, g_omega, S(stream));                                       \
        MF_TRY(t->kf_grad(B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, post_mean, post_cov, post_cross, g_mu0, g_cholP0, g_A, g_b, g_cholQ, g_H, g_y, g_omega, weights, rinv_per_step, info, S(stream))) \
        if (big) return mf::big_kf_grad_##SUF(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step,        \
                                              post_mean, post_cov, post_cross, weights, g_mu0, g_cholP0, g_A, g_b,     \
                                              g_cholQ, g_H, g_y, g_omega, S(stream));                                  \
        return -100;    \
    }                                                                                                                  \
    int mf_kf_loglik_grad_streamed_##SUF(int64_t B, int64_t Tn, int d, int m, const T* mu0, const T* cholP0,           \
                                         const T* A, const T* b, const T* cholQ, const T* H, const T* y, const T* Rinv, \
                                         int rinv_per_step, const T* weights, T* g_mu0, T* g_cholP0, T* g_A, T* g_b,    \
                                         T* g_cholQ, T* g_H, T* g_y, T* g_omega, void* ws, size_t ws_bytes, int* info,  \
                                         int64_t chunks, const void* fwd_ws, int64_t fwd_chunks_per_series,             \
                                         int64_t fwd_chunk_length, void* prof_start, void* prof_stop, void* stream) {   \
        if (B < 1) return -1;                                                                                          \
        if (Tn < 2) return -2;                                                                                         \
        if (d < 1) return -3;                                                                                          \
        if (m < 1) return -4;                                                                                          \
        if (!mu0 || !cholP0 || !A || !b || !cholQ) return -5;                                                          \
        if (!H || !y || !Rinv) return -10;                                                                             \
        if (!g_mu0 || !g_cholP0 || !g_A || !g_cholQ) return -15;   /* g_b, g_H, g_y, g_omega: NULL = not wanted */        \
        const auto* gt = grad_table_for<T>(d);                                                                         \
        if (!gt) return -101;                                                                                          \
        return gt->run(B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, weights, g_mu0, g_cholP0, g_A,    \
                       g_b, g_cholQ, g_H, g_y, g_omega, ws, ws_bytes, info, chunks, fwd_ws, fwd_chunks_per_series,      \
                       fwd_chunk_length, static_cast<hipEvent_t>(prof_start), static_cast<hipEvent_t>(prof_stop),      \
                       S(stream));                                                                                    \
    }                                                                                                                  \
    int mf_kf_posterior_chain_from_filter_##SUF(int64_t B, int64_t Tn, int d, int m, const T* mu0, const T* cholP0,     \
                                                const T* A, const T* b, const T* cholQ, const T* H, const T* y,         \
                                                const T* Rinv, int rinv_per_step, T* a_post, T* mu0_post, T* b_post,    \
                                                T* cholP0_post, T* cholQ_post, void* ws, size_t ws_bytes, int* info,    \
                                                const void* fwd_ws, int64_t fwd_chunks_per_series,                      \
                                                int64_t fwd_chunk_length, void* prof_start, void* prof_stop,            \
                                                void* stream) {                                                         \
        if (B < 1) return -1;                                                                                          \
        if (Tn < 2) return -2;                                                                                         \
        if (d < 1) return -3;                                                                                          \
        if (m < 1) return -4;                                                                                          \
        if (!mu0 || !cholP0 || !A || !b || !cholQ) return -5;                                                          \
        if (!H || !y || !Rinv) return -10;                                                                             \
        if (!mu0_post || !cholP0_post || !a_post || !b_post || !cholQ_post) return -14;                                \
        const auto* gt = grad_table_for<T>(d);                                                                         \
        if (!gt) return -101;                                                                                          \
        return gt->post_from_fwd(B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, a_post, mu0_post,       \
                                 b_post, cholP0_post, cholQ_post, ws, ws_bytes, info, fwd_ws, fwd_chunks_per_series,    \
                                 fwd_chunk_length, static_cast<hipEvent_t>(prof_start),                                \
                                 static_cast<hipEvent_t>(prof_stop), S(stream));                                       \
    }                                                                                                                  \
    int mf_gpr_matern_posterior_chain_##SUF(int64_t B, int64_t Tn, int ncomp, const int* orders, const T* lam,          \
                                            const T* var, int per_series, const T* t, const T* y, const T* rinv,        \
                                            T jitter, T* a_post, T* mu0_post, T* b_post, T* cholP0_post,                \
                                            T* cholQ_post, void* ws, size_t ws_bytes, int* info, const void* fwd_ws,    \
                                            int64_t fwd_chunks_per_series, int64_t fwd_chunk_length, void* stream) {    \
        if (B < 1) return -1;                                                                                          \
        if (Tn < 2) return -2;                                                                                         \
        if (ncomp < 1 || !orders) return -3;                                                                           \
        if (!lam || !var || !t || !y || !rinv) return -5;                                                              \
        if (!a_post || !mu0_post || !b_post || !cholP0_post || !cholQ_post) return -12;                                \
        int d = 0;                                                                                                     \
        for (int c = 0; c < ncomp; ++c) d += (orders[c] + 1) / 2;                                                      \
        const auto* gt = grad_table_for<T>(d);                                                                         \
        if (!gt) return -101;                                                                                          \
        return gt->gpr_post_run(B, Tn, ncomp, orders, lam, var, per_series, t, y, rinv, jitter, a_post, mu0_post,       \
                                b_post, cholP0_post, cholQ_post, ws, ws_bytes, info, fwd_ws, fwd_chunks_per_series,     \
                                fwd_chunk_length, S(stream));                                                          \
    }                                                                                                                  \
    int mf_gpr_matern_loglik_grad_##SUF(int64_t B, int64_t Tn, int ncomp, const int* orders, const T* lam, const T* var, \
                                        int per_series, const T* t, const T* y, const T* rinv, T jitter,                \
                                        const T* weights, T* g_packed, T* g_cholP0, T* g_omega, void* ws,              \
                                        size_t ws_bytes, int* info, const void* fwd_ws,                                \
                                        int64_t fwd_chunks_per_series, int64_t fwd_chunk_length, void* stream) {       \
        if (B < 1) return -1;                                                                                          \
        if (Tn < 2) return -2;                                                                                         \
        if (ncomp < 1 || !orders) return -3;                                                                           \
        if (!lam || !var || !t || !y || !rinv) return -5;                                                              \
        if (!g_packed || !g_cholP0) return -13;                                                                        \
        int d = 0;                                                                                                     \
        for (int c = 0; c < ncomp; ++c) d += (orders[c] + 1) / 2;                                                      \
        const auto* gt = grad_table_for<T>(d);                                                                         \
        if (!gt) return -101;                                                                                          \
        return gt->gpr_run(B, Tn, ncomp, orders, lam, var, per_series, t, y, rinv, jitter, weights, g_packed,           \
                           g_cholP0, g_omega, ws, ws_bytes, info, fwd_ws, fwd_chunks_per_series, fwd_chunk_length,      \
                           S(stream));                                                                                 \
    }                                                                                                                  \
    int mf_ssm_kl_grad_##SUF(int64_t B, int64_t Tn, int d, const T* mu0_1, const T* cholP0_1, const T* A_1,            \
                             const T* b_1, const T* cholQ_1, const T* mu0_2, const T* cholP0_2, const T* A_2,          \
                             const T* b_2, const T* cholQ_2, const T* means_1, const T* covs_1, const T* weights,      \
                             const T* adj_N, const T* adj_n, T* g_mu0, T* g_cholP0, T* g_A, T* g_b, T* g_cholQ,        \
                             void* ws, size_t ws_bytes, int* info, void* stream) {                                     \
        MF_HEAD(T, B, Tn, d)                                                                                           \
        if (!t) return -100;                                                                                            \
        if (!mu0_1 || !cholP0_1 || (Tn > 1 && (!A_1 || !b_1 || !cholQ_1))) return -4;                                  \
        if (!mu0_2 || !cholP0_2 || (Tn > 1 && (!A_2 || !b_2 || !cholQ_2))) return -9;                                  \
        if (!means_1 || !covs_1) return -14;                                                                           \
        if (!g_mu0 || !g_cholP0 || (Tn > 1 && (!g_A || !g_b || !g_cholQ))) return -19;                                 \
        return t->kl_grad(B, Tn, mu0_1, cholP0_1, A_1, b_1, cholQ_1, mu0_2, cholP0_2, A_2, b_2, cholQ_2, means_1,      \
                          covs_1, weights, adj_N, adj_n, g_mu0, g_cholP0, g_A, g_b, g_cholQ, ws, ws_bytes, info,       \
                          S(stream));                                                                                  \
    }                                                                                                                  \
    int mf_kf_posterior_chain_##SUF(int64_t B, int64_t Tn, int d, int m, const T* mu0, const T* cholP0, const T* A,    \
                                    const T* b, const T* cholQ, const T* H, const T* y, const T* Rinv,                 \
                                    int rinv_per_step, T* a_post, T* mu0_post, T* b_post, T* cholP0_post,              \
                                    T* cholQ_post, void* ws, size_t ws_bytes, int* info, int64_t chunks,               \
                                    void* prof_start, void* prof_stop, void* stream) {                                 \
        MF_HEAD(T, B, Tn, d)                                                                                           \
        if (!t) return -100;                                                                                            \
        if (m < 1 || m > 4) return -4;                                                                                 \
        if (!mu0 || !cholP0 || (Tn > 1 && (!A || !b || !cholQ))) return -5;                                            \
        if (!H || !y || !Rinv) return -10;                                                                             \
        if (!mu0_post || !cholP0_post || (Tn > 1 && (!a_post || !b_post || !cholQ_post))) return -14;                  \
        if (ws != nullptr && ws_bytes > 0) {       /* the streamed, time-partitioned kernels (mf_post_lds.hpp) */     \
            if (const auto* pt = post_table_for<T>(d)) {                                                               \
                const int rc = pt->chain(B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, a_post,        \
                                         mu0_post, b_post, cholP0_post, cholQ_post, ws, ws_bytes, info, chunks,        \
                                         static_cast<hipEvent_t>(prof_start), static_cast<hipEvent_t>(prof_stop),      \
                                         S(stream));                                                                   \
                if (rc != -101) return rc;         /* -101: not their call (outputs, alignment): one lane per series */ \
            }                                                                                                          \
        }                                                                                                              \
        return t->posterior_chain(B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, a_post, mu0_post,     \
                                  b_post, cholP0_post, cholQ_post, info, S(stream));                                   \
    }                                                                                                                  \
    int mf_ssm_kl_divergence_##SUF(int64_t B, int64_t Tn, int d, const T* mu0_1, const T* cholP0_1, const T* A_1,      \
                                   const T* b_1, const T* cholQ_1, const T* mu0_2, const T* cholP0_2, const T* A_2,    \
                                   const T* b_2, const T* cholQ_2, T* out, T* out_means, T* out_covs,          \
                                   T* out_cross, T* out_N, T* out_n, void* ws, size_t ws_bytes, int* info,             \
                                   void* stream) {                                                                     \
        MF_HEAD(T, B, Tn, d)                                                                                           \
        if (!mu0_1 || !cholP0_1 || (Tn > 1 && (!A_1 || !b_1 || !cholQ_1))) return -4;                                  \
        if (!mu0_2 || !cholP0_2 || (Tn > 1 && (!A_2 || !b_2 || !cholQ_2))) return -9;                                  \
        if (!out) return -14;                                                                                          \
        if (!t) {   /* 16 <= d <= 32, the value alone: the moment recursion and the block terms in one walk (mf_wave_ops.hpp) */ \
            if (Tn > 1 && !out_means && !out_covs && !out_cross && !out_N && !out_n) {                                  \
                const int rc_ = mf::wave_ssm_kl_fused_##SUF(B, Tn, d, mu0_1, cholP0_1, A_1, b_1, cholQ_1, mu0_2, cholP0_2, A_2, \
                                                            b_2, cholQ_2, out, ws, ws_bytes, S(stream));                \
                if (rc_ != -101) return rc_;                                                                           \
            }                                                                                                          \
            return -100;                                                                                               \
        }                                                                                                              \
        return t->kl(B, Tn, mu0_1, cholP0_1, A_1, b_1, cholQ_1, mu0_2, cholP0_2, A_2, b_2, cholQ_2, out, out_means,    \
                     out_covs, out_cross, out_N, out_n, ws, ws_bytes, info, S(stream));                                \
    }                                                                                                                  \
    int mf_ssm_marginals_grad_##SUF(int64_t B, int64_t Tn, int d, const T* cholP0, const T* A, const T* cholQ,         \
                                    const T* means, const T* covs, const T* g_means, const T* g_covs, T* g_mu0,        \
                                    T* g_cholP0, T* g_A, T* g_b, T* g_cholQ, void* ws, size_t ws_bytes, void* stream) { \
        MF_HEAD(T, B, Tn, d)                                                                                           \
        if (!t) return -100;                                                                                            \
        if (!cholP0 || (Tn > 1 && (!A || !cholQ))) return -4;                                                          \
        if (!means || !covs) return -7;                                                                                \
        if (!g_mu0 || !g_cholP0 || (Tn > 1 && (!g_A || !g_b || !g_cholQ))) return -11;                                 \
        return t->marginals_grad(B, Tn, cholP0, A, cholQ, means, covs, g_means, g_covs, g_mu0, g_cholP0, g_A, g_b,     \
                                 g_cholQ, ws, ws_bytes, S(stream));                                                    \
    }

MF_DEFINE(f64, double)
MF_DEFINE(f32, float)
MF_DEFINE2(f64, double)
MF_DEFINE2(f32, float)
MF_DEFINE3(f64, double)
MF_DEFINE3(f32, float)
MF_DEFINE4(f64, double)
MF_DEFINE4(f32, float)
MF_DEFINE4B(f64, double)
MF_DEFINE4B(f32, float)
MF_DEFINE5(f64, double)
MF_DEFINE5(f32, float)

// KL(q1 || q2) from q1's moments on register tiles, 16 <= d <= 32 (mf_wave.hpp: wave_ssm_kl_terms_kernel); see the header
size_t mf_ssm_kl_from_moments_workspace_bytes(int64_t B, int64_t T, int d, int elem_size) {
    if (B < 1 || T < 1 || d < 16 || d > 32 || (elem_size != 4 && elem_size != 8)) return 0;
    return size_t(B) * size_t(T) * size_t(elem_size);
}
#define MF_DEFINE_KLM(SUF, T)                                                                                                     \
    int mf_ssm_kl_from_moments_##SUF(int64_t B, int64_t Tn, int d, const T* cholP0_1, const T* cholQ_1, const T* cholP0_2,         \
                                     const T* A_2, const T* cholQ_2, const T* covs_1, const T* cross_1, const T* mean_diff, T* out, \
                                     void* ws, size_t ws_bytes, void* stream) {                                                    \
        if (B < 0) return -1;                                                                                                      \
        if (Tn < 1) return -2;                                                                                                     \
        if (d < 16 || d > 32) return -100;                                                                                         \
        if (B == 0) return 0;                                                                                                      \
        if (!cholP0_1 || (Tn > 1 && !cholQ_1)) return -4;                                                                          \
        if (!cholP0_2 || (Tn > 1 && (!A_2 || !cholQ_2))) return -6;                                                                \
        if (!covs_1 || (Tn > 1 && !cross_1)) return -9;                                                                            \
        if (!mean_diff) return -11;                                                                                                \
        if (!out) return -12;                                                                                                      \
        return mf::wave_ssm_kl_##SUF(B, Tn, d, cholP0_1, cholQ_1, cholP0_2, A_2, cholQ_2, covs_1, cross_1, mean_diff, out, ws,     \
                                     ws_bytes, S(stream));                                                                         \
    }
MF_DEFINE_KLM(f64, double)
MF_DEFINE_KLM(f32, float)
#undef MF_DEFINE_KLM

static size_t big_max(size_t a, size_t b) { return a > b ? a : b; }
static bool big_dim(int d, int elem_size) {
    return d > mf::MF_MAX_D && d <= (elem_size == 4 ? mf::MF_MAX_D_BIG : mf::MF_MAX_D_BIG_F64);
}

size_t mf_btd_cholesky_workspace_bytes(int64_t B, int64_t T, int d, int elem_size) {
    if (B < 1 || T < 1) return 0;
    size_t small = 0;
    if (elem_size == 4) { if (const auto* t = table_for<float>(d)) small = t->btd_cholesky_ws(B, T); }
    else if (const auto* t = table_for<double>(d)) small = t->btd_cholesky_ws(B, T);
    // 10 <= d <= 15: sized for whichever engine takes the call (row kernels when the plan is theirs, else the tile engine)
    const size_t large = big_dim(d, elem_size) ? mf::big_btd_par_ws(B, T, d, 0, elem_size) : 0;
    return small > large ? small : large;
}
size_t mf_btd_solve_workspace_bytes(int64_t Bl, int64_t Br, int64_t T, int d, int elem_size) {
    if (Bl < 1 || Br < 1 || T < 1) return 0;
    size_t small = 0;
    if (elem_size == 4) { if (const auto* t = table_for<float>(d)) small = t->btd_solve_ws(Bl, Br, T); }
    else if (const auto* t = table_for<double>(d)) small = t->btd_solve_ws(Bl, Br, T);
    // 10 <= d <= 15: sized for whichever engine takes the call (row kernels when the plan is theirs, else the tile engine)
    const size_t large = big_dim(d, elem_size) ? mf::big_btd_solve_ws(Bl, Br, T, d, elem_size) : 0;
    return small > large ? small : large;
}

size_t mf_btd_diag_of_inverse_workspace_bytes(int64_t B, int64_t T, int d, int elem_size) {
    if (B < 1 || T < 1) return 0;
    size_t small = 0;
    if (elem_size == 4) { if (const auto* t = table_for<float>(d)) small = t->btd_diag_of_inverse_ws(B, T); }
    else if (const auto* t = table_for<double>(d)) small = t->btd_diag_of_inverse_ws(B, T);
    // 10 <= d <= 15: sized for whichever engine takes the call (row kernels when the plan is theirs, else the tile engine)
    const size_t large = big_dim(d, elem_size) ? big_max(mf::big_marginal_covs_ws(B, T, d, elem_size), mf::big_btd_tak_ws(B, T, d, elem_size)) : 0;
    return small > large ? small : large;
}

// reverse mode of cholesky / block_diagonal_of_inverse: register and row kernels only (-> 0 beyond: the Python layer keeps its
// differentiable torch route there)
size_t mf_btd_grad_workspace_bytes(int64_t B, int64_t T, int d, int elem_size) {
    if (B < 1 || T < 1) return 0;
    size_t ws = 0;
    if (elem_size == 4) { if (const auto* t = table_for<float>(d)) ws = t->btd_grad_ws(B, T); }
    else if (const auto* t = table_for<double>(d)) ws = t->btd_grad_ws(B, T);
    // 10 <= d <= 32: the parallel-in-time adjoints on register tiles (mf_adj.hip)
    if (ws == 0 && mf::adj_covers(d)) ws = mf::adj_grad_ws(B, T, d, elem_size);
    return ws;
}

size_t mf_btd_udl_workspace_bytes(int64_t B, int64_t T, int d, int elem_size) {
    if (B < 1 || T < 1) return 0;
    size_t small = 0;
    if (elem_size == 4) { if (const auto* t = table_for<float>(d)) small = t->btd_udl_ws(B, T); }
    else if (const auto* t = table_for<double>(d)) small = t->btd_udl_ws(B, T);
    // 10 <= d <= 15: sized for whichever engine takes the call (row kernels when the plan is theirs, else the tile engine)
    const size_t large = big_dim(d, elem_size) ? mf::big_btd_par_ws(B, T, d, 1, elem_size) : 0;
    return small > large ? small : large;
}

size_t mf_ssm_marginals_workspace_bytes(int64_t B, int64_t T, int d, int elem_size) {
    if (B < 1 || T < 1) return 0;
    size_t small = 0;
    if (elem_size == 4) { if (const auto* t = table_for<float>(d)) small = t->marginals_ws(B, T); }
    else if (const auto* t = table_for<double>(d)) small = t->marginals_ws(B, T);
    // 10 <= d <= 15: sized for whichever engine takes the call (row kernels when the plan is theirs, else the tile engine)
    const size_t large = big_dim(d, elem_size) ? mf::big_marginal_covs_ws(B, T, d, elem_size) : 0;
    return small > large ? small : large;
}

size_t mf_ssm_kl_workspace_bytes(int64_t B, int64_t T, int d, int elem_size) {
    if (B < 1 || T < 1) return 0;
    if (elem_size == 4) { const auto* t = table_for<float>(d); return t ? t->kl_ws(B, T) : mf::wave_ssm_kl_fused_ws(B, T, d, 4); }
    const auto* t = table_for<double>(d);
    return t ? t->kl_ws(B, T) : mf::wave_ssm_kl_fused_ws(B, T, d, 8);
}

size_t mf_ssm_adjoint_workspace_bytes(int64_t B, int64_t T, int d, int elem_size) {
    if (B < 1 || T < 1) return 0;
    if (elem_size == 4) { const auto* t = table_for<float>(d); return t ? t->adjoint_ws(B, T) : 0; }
    const auto* t = table_for<double>(d);
    return t ? t->adjoint_ws(B, T) : 0;
}

size_t mf_kf_posterior_chain_workspace_bytes(int64_t B, int64_t T, int d, int m, int rinv_per_step, int elem_size,
                                             int64_t chunks) {
    if (B < 1 || T < 2 || d < 1 || m < 1) return 0;
    if (elem_size == 4) { const auto* t = post_table_for<float>(d); return t ? t->ws(B, T, m, rinv_per_step, chunks) : 0; }
    const auto* t = post_table_for<double>(d);
    return t ? t->ws(B, T, m, rinv_per_step, chunks) : 0;
}

int mf_kf_loglik_plan(int64_t B, int64_t T, int d, int m, int rinv_per_step, int elem_size, int64_t chunks, int aligned16,
                      int* path, int64_t* chunks_per_series, int64_t* chunk_length) {
    if (B < 1 || T < 1 || d < 1 || !path || !chunks_per_series || !chunk_length) return -1;
    long P = 0, L = 0;
    int rc = -100;
    if (elem_size == 4) { if (const auto* t = table_for<float>(d)) rc = t->kf_loglik_plan(B, T, m, rinv_per_step, chunks, aligned16, path, &P, &L); }
    else if (const auto* t = table_for<double>(d)) rc = t->kf_loglik_plan(B, T, m, rinv_per_step, chunks, aligned16, path, &P, &L);
    *chunks_per_series = P;
    *chunk_length = L;
    return rc;
}

size_t mf_kf_posterior_chain_from_filter_workspace_bytes(int64_t B, int64_t T, int d, int m, int rinv_per_step, int elem_size,
                                                         int64_t fwd_chunks_per_series) {
    if (B < 1 || T < 2 || d < 1 || m < 1) return 0;
    if (elem_size == 4) { const auto* t = grad_table_for<float>(d); return t ? t->post_from_fwd_ws(B, T, m, rinv_per_step, fwd_chunks_per_series) : 0; }
    const auto* t = grad_table_for<double>(d);
    return t ? t->post_from_fwd_ws(B, T, m, rinv_per_step, fwd_chunks_per_series) : 0;
}

size_t mf_gpr_matern_posterior_chain_workspace_bytes(int64_t B, int64_t T, int d, int elem_size, int64_t fwd_chunks_per_series) {
    if (B < 1 || T < 2 || d < 1) return 0;
    if (elem_size == 4) { const auto* t = grad_table_for<float>(d); return t ? t->gpr_post_ws(B, T, fwd_chunks_per_series) : 0; }
    const auto* t = grad_table_for<double>(d);
    return t ? t->gpr_post_ws(B, T, fwd_chunks_per_series) : 0;
}

size_t mf_gpr_matern_loglik_grad_workspace_bytes(int64_t B, int64_t T, int d, int elem_size, int64_t fwd_chunks_per_series) {
    if (B < 1 || T < 2 || d < 1) return 0;
    if (elem_size == 4) { const auto* t = grad_table_for<float>(d); return t ? t->gpr_ws(B, T, fwd_chunks_per_series) : 0; }
    const auto* t = grad_table_for<double>(d);
    return t ? t->gpr_ws(B, T, fwd_chunks_per_series) : 0;
}

size_t mf_kf_loglik_grad_streamed_workspace_bytes(int64_t B, int64_t T, int d, int m, int rinv_per_step, int elem_size,
                                                  int64_t chunks) {
    if (B < 1 || T < 2 || d < 1 || m < 1) return 0;
    if (elem_size == 4) { const auto* t = grad_table_for<float>(d); return t ? t->ws(B, T, m, rinv_per_step, chunks) : 0; }
    const auto* t = grad_table_for<double>(d);
    return t ? t->ws(B, T, m, rinv_per_step, chunks) : 0;
}

size_t mf_btd_logdet_quad_workspace_bytes(int64_t B, int64_t T, int d, int elem_size) {
    if (B < 1 || T < 1 || d < 1) return 0;
    if (elem_size == 4) { const auto* t = table_for<float>(d); return t ? t->btd_logdet_quad_ws(B, T, 0) : 0; }
    const auto* t = table_for<double>(d);
    return t ? t->btd_logdet_quad_ws(B, T, 0) : 0;
}

}  // extern "C"
