// Instantiations of the panel kernels (mf_panel.hpp: one workgroup of d / 16 wavefronts per (series, chunk), register panels,
// 32 < d <= 64) and the entry points the tile engine's launcher (mf_big_impl.hpp) hands the log-likelihood to.
#include "mf_panel.hpp"
#include "mf_launch.hpp"

namespace mf {

namespace {
template <typename K> bool panel_attr(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}
template <typename T, int NT, int MT, bool EX>
int panel_launch0(const wv::WvArgs<T>& a, const RedSys<T>& out, hipStream_t st) {
    constexpr int bytes = pn::Lds<T, NT, MT>::BYTES;
    static const bool ok = panel_attr(&pn::panel_kf_chunk_kernel<T, NT, MT, EX>, bytes);
    if (!ok) return -1000;
    hipLaunchKernelGGL((pn::panel_kf_chunk_kernel<T, NT, MT, EX>), dim3((unsigned)(a.B * a.P)), dim3(64 * NT), bytes, st, a, out);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
template <typename T, int NT, int MT>
int panel_launch0_ex(const wv::WvArgs<T>& a, const RedSys<T>& out, hipStream_t st) {
    return a.d == 16 * NT ? panel_launch0<T, NT, MT, true>(a, out, st) : panel_launch0<T, NT, MT, false>(a, out, st);
}
template <typename T>
int panel_level0(long B, long Tn, int d, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
                 const T* y, const T* Rinv, int rinv_per_step, long P, long L, const RedSys<T>& out, int* info, hipStream_t st) {
    if (!panel_covers(d, m)) return -101;
    const wv::WvArgs<T> a{B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, info};
    if (d <= 32) return m <= 16 ? panel_launch0_ex<T, 2, 1>(a, out, st) : panel_launch0_ex<T, 2, 2>(a, out, st);
    if (d <= 48) return m <= 16 ? panel_launch0_ex<T, 3, 1>(a, out, st) : panel_launch0_ex<T, 3, 2>(a, out, st);
    return m <= 16 ? panel_launch0_ex<T, 4, 1>(a, out, st) : panel_launch0_ex<T, 4, 2>(a, out, st);
}
// StateSpaceModel._build_precision (+ observation terms, + information vector) on the panel kernels: chunks of 16 blocks
template <typename T, int NT, int MT, bool EX>
int panel_prec_launch(const wv::WvArgs<T>& a, const pn::PrecOut<T>& po, hipStream_t st) {
    constexpr int bytes = pn::Lds<T, NT, MT>::BYTES;
    static const bool ok = panel_attr(&pn::panel_kf_chunk_kernel<T, NT, MT, EX, true>, bytes);
    if (!ok) return -1000;
    hipLaunchKernelGGL((pn::panel_kf_chunk_kernel<T, NT, MT, EX, true>), dim3((unsigned)(a.B * a.P)), dim3(64 * NT), bytes, st, a,
                       RedSys<T>{}, po);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
template <typename T, int NT, int MT>
int panel_prec_ex(const wv::WvArgs<T>& a, const pn::PrecOut<T>& po, hipStream_t st) {
    return a.d == 16 * NT ? panel_prec_launch<T, NT, MT, true>(a, po, st) : panel_prec_launch<T, NT, MT, false>(a, po, st);
}
template <typename T>
int panel_prec(long B, long Tn, int d, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
               const T* y, const T* Rinv, int rinv_per_step, T* diag, T* sub, T* eta, hipStream_t st) {
    const int mm = H ? m : 1;
    if (!(d > 32 && d <= 64) || mm < 1 || mm > 32 || B <= 0 || Tn <= 0) return -101;
    const long nt = Tn - 1, L = 16, P = nt > 0 ? (nt + L - 1) / L : 1;
    const wv::WvArgs<T> a{B, Tn, d, mm, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, nullptr};
    const pn::PrecOut<T> po{diag, sub, eta};
    if (d <= 48) return mm <= 16 ? panel_prec_ex<T, 3, 1>(a, po, st) : panel_prec_ex<T, 3, 2>(a, po, st);
    return mm <= 16 ? panel_prec_ex<T, 4, 1>(a, po, st) : panel_prec_ex<T, 4, 2>(a, po, st);
}
template <typename T, int NT, bool FINAL, bool EX>
int panel_red_launch(const RedSys<T>& in, const RedSys<T>& out, long B, long P, int d, T add_const, T* out_scalar, int* info,
                     hipStream_t st) {
    constexpr int bytes = pn::Lds<T, NT, 1>::BYTES;
    static const bool ok = panel_attr(&pn::panel_red_kernel<T, NT, FINAL, EX>, bytes);
    if (!ok) return -1000;
    hipLaunchKernelGGL((pn::panel_red_kernel<T, NT, FINAL, EX>), dim3((unsigned)(B * P)), dim3(64 * NT), bytes, st, in, out, B, P, d,
                       add_const, out_scalar, info);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
template <typename T, int NT>
int panel_red_nt(const RedSys<T>& in, const RedSys<T>& out, long B, long P, int d, T add_const, T* out_scalar, int* info, int fin,
                 hipStream_t st) {
    const bool ex = d == 16 * NT;
    if (fin) return ex ? panel_red_launch<T, NT, true, true>(in, out, B, P, d, add_const, out_scalar, info, st)
                       : panel_red_launch<T, NT, true, false>(in, out, B, P, d, add_const, out_scalar, info, st);
    return ex ? panel_red_launch<T, NT, false, true>(in, out, B, P, d, add_const, out_scalar, info, st)
              : panel_red_launch<T, NT, false, false>(in, out, B, P, d, add_const, out_scalar, info, st);
}
template <typename T>
int panel_red_t(const RedSys<T>& in, const RedSys<T>& out, long B, long P, int d, T add_const, T* out_scalar, int* info, int fin,
                hipStream_t st) {
    if (d <= 16 || d > 64) return -101;
    if (d <= 32) return panel_red_nt<T, 2>(in, out, B, P, d, add_const, out_scalar, info, fin, st);
    return d <= 48 ? panel_red_nt<T, 3>(in, out, B, P, d, add_const, out_scalar, info, fin, st)
                   : panel_red_nt<T, 4>(in, out, B, P, d, add_const, out_scalar, info, fin, st);
}
}  // namespace

// 32 < d <= 64: every call.  16 < d <= 32: the calls with more than four outputs - up to four the wave kernels (mf_wave.hpp: the whole
// chunk in one wavefront, no barrier) are 1.5-2.3 x faster than two wavefronts with barriers between them, beyond four they do not
// exist and the LDS-tile engine took the call: 58 ms against 10 ms here at d = 32, m = 8, B = 512, T = 1000, fp64
// (profiles/r06_panel_nt2_ab.txt).
bool panel_covers(int d, int m) { return m >= 1 && m <= 32 && ((d > 32 && d <= 64) || (d > 16 && d <= 32 && m > 4)); }

int panel_kf_level0_f64(long B, long Tn, int d, int m, const double* mu0, const double* cholP0, const double* A, const double* b,
                        const double* cholQ, const double* H, const double* y, const double* Rinv, int rinv_per_step, long P, long L,
                        const RedSys<double>& out, int* info, hipStream_t st) {
    return panel_level0<double>(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, out, info, st);
}
int panel_kf_level0_f32(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                        const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step, long P, long L,
                        const RedSys<float>& out, int* info, hipStream_t st) {
    return panel_level0<float>(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, out, info, st);
}
int panel_ssm_precision_f64(long B, long Tn, int d, int m, const double* mu0, const double* cholP0, const double* A, const double* b,
                            const double* cholQ, const double* H, const double* y, const double* Rinv, int rinv_per_step, double* diag,
                            double* sub, double* eta, hipStream_t st) {
    return panel_prec<double>(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, diag, sub, eta, st);
}
int panel_ssm_precision_f32(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                            const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step, float* diag,
                            float* sub, float* eta, hipStream_t st) {
    return panel_prec<float>(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, diag, sub, eta, st);
}
namespace {
template <typename T, int NT>
int panel_chol_up_nt(const RedSys<T>& in, const RedSys<T>& out, long B, long P, long L, int d, int rev, T* pivs, int* info,
                     hipStream_t st) {
    constexpr int bytes = pn::Lds<T, NT, 1>::BYTES;
    const dim3 grid((unsigned)(B * P)), block(64 * NT);
    if (d == 16 * NT) {
        static const bool ok = panel_attr(&pn::panel_red_kernel<T, NT, false, true>, bytes);
        if (!ok) return -1000;
        hipLaunchKernelGGL((pn::panel_red_kernel<T, NT, false, true>), grid, block, bytes, st, in, out, B, P, d, T(0),
                           static_cast<T*>(nullptr), info, L, rev, pivs);
    } else {
        static const bool ok = panel_attr(&pn::panel_red_kernel<T, NT, false, false>, bytes);
        if (!ok) return -1000;
        hipLaunchKernelGGL((pn::panel_red_kernel<T, NT, false, false>), grid, block, bytes, st, in, out, B, P, d, T(0),
                           static_cast<T*>(nullptr), info, L, rev, pivs);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
// the up-sweep of the time-partitioned Cholesky / U D U^T factorisation for 32 < d <= 64 on the panel reduction kernel (operator
// mode): per chunk of L blocks the pivot of its last block, the contribution to the separator in front of it and the coupling to it,
// in the arrays the tile engine's boundary and emit passes read
template <typename T>
int panel_chol_up_t(long B, long n, int d, long P, long L, const T* diag, const T* sub, T* oDv, T* oGU, T* oF, T* piv, int rev,
                    int* info, hipStream_t st) {
    if (d <= 32 || d > 64 || !sub || P < 2) return -101;
    const RedSys<T> in{const_cast<T*>(diag), nullptr, const_cast<T*>(sub), nullptr, nullptr, nullptr, n, n - 1, -1};
    const RedSys<T> out{oDv, oGU, oF, nullptr, nullptr, nullptr, P, P, 0};
    const int rc = d <= 48 ? panel_chol_up_nt<T, 3>(in, out, B, P, L, d, rev, nullptr, info, st)
                           : panel_chol_up_nt<T, 4>(in, out, B, P, L, d, rev, nullptr, info, st);
    if (rc != 0 || !piv) return rc;
    // the chunk ends of every series: one workgroup walks the P - 1 reduced blocks and leaves the natural-order pivot of each
    const RedSys<T> red{oDv, oGU, oF, nullptr, nullptr, nullptr, P, P, 0};
    const RedSys<T> none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1, 0};
    return d <= 48 ? panel_chol_up_nt<T, 3>(red, none, B, 1, P - 1, d, 0, piv, info, st)
                   : panel_chol_up_nt<T, 4>(red, none, B, 1, P - 1, d, 0, piv, info, st);
}
}  // namespace
int panel_chol_up_f64(long B, long n, int d, long P, long L, const double* diag, const double* sub, double* oDv, double* oGU, double* oF,
                      double* piv, int rev, int* info, hipStream_t st) {
    return panel_chol_up_t<double>(B, n, d, P, L, diag, sub, oDv, oGU, oF, piv, rev, info, st);
}
int panel_chol_up_f32(long B, long n, int d, long P, long L, const float* diag, const float* sub, float* oDv, float* oGU, float* oF,
                      float* piv, int rev, int* info, hipStream_t st) {
    return panel_chol_up_t<float>(B, n, d, P, L, diag, sub, oDv, oGU, oF, piv, rev, info, st);
}
namespace {
template <typename T, int NT>
int panel_chol_emit_nt(long B, long n, int d, long P, long L, const T* diag, const T* sub, const T* piv, T* ldiag, T* lsub, int* info,
                       hipStream_t st) {
    constexpr int bytes = pn::Lds<T, NT, 1>::BYTES;
    const dim3 grid((unsigned)(B * P)), block(64 * NT);
    if (d == 16 * NT) {
        static const bool ok = panel_attr(&pn::panel_chol_emit_kernel<T, NT, true>, bytes);
        if (!ok) return -1000;
        hipLaunchKernelGGL((pn::panel_chol_emit_kernel<T, NT, true>), grid, block, bytes, st, B, n, d, P, L, diag, sub, piv, ldiag, lsub, info);
    } else {
        static const bool ok = panel_attr(&pn::panel_chol_emit_kernel<T, NT, false>, bytes);
        if (!ok) return -1000;
        hipLaunchKernelGGL((pn::panel_chol_emit_kernel<T, NT, false>), grid, block, bytes, st, B, n, d, P, L, diag, sub, piv, ldiag, lsub, info);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
template <typename T>
int panel_chol_emit_t(long B, long n, int d, long P, long L, const T* diag, const T* sub, const T* piv, T* ldiag, T* lsub, int* info,
                      hipStream_t st) {
    if (d <= 32 || d > 64 || !sub || P < 2) return -101;
    return d <= 48 ? panel_chol_emit_nt<T, 3>(B, n, d, P, L, diag, sub, piv, ldiag, lsub, info, st)
                   : panel_chol_emit_nt<T, 4>(B, n, d, P, L, diag, sub, piv, ldiag, lsub, info, st);
}
}  // namespace
int panel_chol_emit_f64(long B, long n, int d, long P, long L, const double* diag, const double* sub, const double* piv, double* ldiag,
                        double* lsub, int* info, hipStream_t st) {
    return panel_chol_emit_t<double>(B, n, d, P, L, diag, sub, piv, ldiag, lsub, info, st);
}
int panel_chol_emit_f32(long B, long n, int d, long P, long L, const float* diag, const float* sub, const float* piv, float* ldiag,
                        float* lsub, int* info, hipStream_t st) {
    return panel_chol_emit_t<float>(B, n, d, P, L, diag, sub, piv, ldiag, lsub, info, st);
}
int panel_red_f64(const RedSys<double>& in, const RedSys<double>& out, long B, long P, int d, double add_const, double* out_scalar,
                  int* info, int final_level, hipStream_t st) {
    return panel_red_t<double>(in, out, B, P, d, add_const, out_scalar, info, final_level, st);
}
int panel_red_f32(const RedSys<float>& in, const RedSys<float>& out, long B, long P, int d, float add_const, float* out_scalar,
                  int* info, int final_level, hipStream_t st) {
    return panel_red_t<float>(in, out, B, P, d, add_const, out_scalar, info, final_level, st);
}

}  // namespace mf
