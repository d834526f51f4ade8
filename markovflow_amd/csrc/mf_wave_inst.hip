// Instantiations of the wave kernels (mf_wave.hpp: one wavefront per (series, chunk), register tiles, 16 <= d <= 32) and the
// entry points the tile engine's launcher (mf_big_impl.hpp) hands its level 0 to.
#include <type_traits>

#include "mf_wave.hpp"
#include "mf_wave_ops.hpp"
#include "mf_wave_grad.hpp"
#include "mf_launch.hpp"
#include "mf_wave_api.hpp"

namespace mf {

namespace {
// wavefronts per SIMD the level-0 kernel is compiled for (its register budget; none of these spills, `make` prints the usage):
// one tile per matrix - fp64 two, fp32 four; 2 x 2 tiles - fp64 one (502 registers), fp32 two.  With two to four outputs the
// observation rows cost up to forty more registers: one wavefront fewer in fp32
template <typename T, int NT, int M> constexpr int wave_wpe() {
    return NT == 1 ? (sizeof(T) == 8 ? 2 : (M == 1 ? 4 : 3)) : (sizeof(T) == 8 ? 1 : 2);
}
template <typename T, int NT, int M>
int wave_launch(const wv::WvArgs<T>& a, const RedSys<T>& out, hipStream_t st) {
    const dim3 grid((unsigned)(a.B * a.P)), block(64);
    if (a.d == 16 * NT) hipLaunchKernelGGL((wv::wave_kf_chunk_kernel<T, NT, M, wave_wpe<T, NT, M>(), true>), grid, block, 0, st, a, out);
    else hipLaunchKernelGGL((wv::wave_kf_chunk_kernel<T, NT, M, wave_wpe<T, NT, M>(), false>), grid, block, 0, st, a, out);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
// the pivot's Cholesky factor and the next chol(Q)'s inverse in one pass (wave_kf_pair_kernel); MF_WAVE_PAIR=0: the plain kernel
template <typename T, int NT, int M>
int wave_launch_pair(const wv::WvArgs<T>& a, const RedSys<T>& out, hipStream_t st) {
    const dim3 grid((unsigned)(a.B * a.P)), block(64);
    if (a.d == 16 * NT) hipLaunchKernelGGL((wv::wave_kf_pair_kernel<T, NT, M, wave_wpe<T, NT, M>(), true>), grid, block, 0, st, a, out);
    else hipLaunchKernelGGL((wv::wave_kf_pair_kernel<T, NT, M, wave_wpe<T, NT, M>(), false>), grid, block, 0, st, a, out);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
template <typename T>
int wave_level0(long B, long Tn, int d, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
                const T* y, const T* Rinv, int rinv_per_step, long P, long L, const RedSys<T>& out, int* info, hipStream_t st) {
    if (!wave_covers(d, m)) return -101;
    const wv::WvArgs<T> a{B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, info};
    static const bool pair = [] { const char* e = mf_knob("MF_WAVE_PAIR"); return !(e && e[0] == '0'); }();
    if (pair) {
        if (d <= 16) return m == 1 ? wave_launch_pair<T, 1, 1>(a, out, st) : wave_launch_pair<T, 1, wv::WV_MAXM>(a, out, st);
        return m == 1 ? wave_launch_pair<T, 2, 1>(a, out, st) : wave_launch_pair<T, 2, wv::WV_MAXM>(a, out, st);
    }
    if (d <= 16) return m == 1 ? wave_launch<T, 1, 1>(a, out, st) : wave_launch<T, 1, wv::WV_MAXM>(a, out, st);
    return m == 1 ? wave_launch<T, 2, 1>(a, out, st) : wave_launch<T, 2, wv::WV_MAXM>(a, out, st);
}
}  // namespace

bool wave_covers(int d, int m) { return d >= 16 && d <= 32 && m >= 1 && m <= wv::WV_MAXM; }
// wavefronts per SIMD the level-0 kernel runs at (by its registers): what one round of chunks over the chip is sized for
int wave_waves_per_simd(int d, int elem_size) {
    return d <= 16 ? (elem_size == 8 ? wave_wpe<double, 1, 1>() : wave_wpe<float, 1, 1>())
                   : (elem_size == 8 ? wave_wpe<double, 2, 1>() : wave_wpe<float, 2, 1>());
}

int wave_kf_level0_f64(long B, long Tn, int d, int m, const double* mu0, const double* cholP0, const double* A, const double* b,
                       const double* cholQ, const double* H, const double* y, const double* Rinv, int rinv_per_step, long P, long L,
                       const RedSys<double>& out, int* info, hipStream_t st) {
    return wave_level0<double>(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, out, info, st);
}
int wave_kf_level0_f32(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                       const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step, long P, long L,
                       const RedSys<float>& out, int* info, hipStream_t st) {
    return wave_level0<float>(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, out, info, st);
}

// ---- StateSpaceModel._build_precision / BaseKalmanFilter._k_inv_post for 16 <= d <= 32 (wave_ssm_precision_kernel) -----------------
namespace {
template <typename T>
int wave_precision(long B, long Tn, int d, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
                   const T* y, const T* Rinv, int rinv_per_step, T* diag, T* sub, T* eta, hipStream_t st) {
    if (!wave_covers(d, H ? m : 1)) return -101;
    const wv::WvArgs<T> a{B, Tn, d, H ? m : 1, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, 1, 1, nullptr};
    const dim3 grid((unsigned)(B * Tn)), block(64);
    const bool m1 = !H || m == 1;
    auto go = [&](auto nt, auto mm, auto ex) {
        hipLaunchKernelGGL((wv::wave_ssm_precision_kernel<T, decltype(nt)::value, decltype(mm)::value, decltype(ex)::value>), grid, block, 0,
                           st, a, diag, sub, eta);
    };
    using std::integral_constant;
    using I1 = integral_constant<int, 1>;
    using I2 = integral_constant<int, 2>;
    using IM = integral_constant<int, wv::WV_MAXM>;
    using Tt = integral_constant<bool, true>;
    using Ff = integral_constant<bool, false>;
    if (d <= 16) {
        if (d == 16) { if (m1) go(I1{}, I1{}, Tt{}); else go(I1{}, IM{}, Tt{}); }
        else { if (m1) go(I1{}, I1{}, Ff{}); else go(I1{}, IM{}, Ff{}); }
    } else {
        if (d == 32) { if (m1) go(I2{}, I1{}, Tt{}); else go(I2{}, IM{}, Tt{}); }
        else { if (m1) go(I2{}, I1{}, Ff{}); else go(I2{}, IM{}, Ff{}); }
    }
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
}  // namespace
int wave_ssm_precision_f64(long B, long Tn, int d, int m, const double* mu0, const double* cholP0, const double* A, const double* b,
                           const double* cholQ, const double* H, const double* y, const double* Rinv, int rinv_per_step, double* diag,
                           double* sub, double* eta, hipStream_t st) {
    return wave_precision<double>(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, diag, sub, eta, st);
}
int wave_ssm_precision_f32(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                           const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step, float* diag,
                           float* sub, float* eta, hipStream_t st) {
    return wave_precision<float>(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, diag, sub, eta, st);
}

// ---- StateSpaceModel.kl_divergence for 16 <= d <= 32 (wave_ssm_kl_terms_kernel + row_sums_kernel); ws: B * Tn scalars --------------
namespace {
template <typename T>
int wave_kl(long B, long Tn, int d, const T* cp0_1, const T* cq_1, const T* cp0_2, const T* a_2, const T* cq_2, const T* cov,
            const T* cross, const T* mdiff, T* out, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!wave_covers(d, 1) || B <= 0 || Tn <= 0) return -101;
    if (ws == nullptr || ws_bytes < size_t(B) * Tn * sizeof(T)) return -15;
    T* terms = static_cast<T*>(ws);
    const wv::WvArgs<T> a{B, Tn, d, 1, nullptr, cp0_2, a_2, nullptr, cq_2, nullptr, nullptr, nullptr, 0, 1, 1, nullptr};
    const dim3 grid((unsigned)(B * Tn)), block(64);
    if (d <= 16) {
        if (d == 16) hipLaunchKernelGGL((wv::wave_ssm_kl_terms_kernel<T, 1, true>), grid, block, 0, st, a, cp0_1, cq_1, cov, cross, mdiff, terms);
        else hipLaunchKernelGGL((wv::wave_ssm_kl_terms_kernel<T, 1, false>), grid, block, 0, st, a, cp0_1, cq_1, cov, cross, mdiff, terms);
    } else {
        if (d == 32) hipLaunchKernelGGL((wv::wave_ssm_kl_terms_kernel<T, 2, true>), grid, block, 0, st, a, cp0_1, cq_1, cov, cross, mdiff, terms);
        else hipLaunchKernelGGL((wv::wave_ssm_kl_terms_kernel<T, 2, false>), grid, block, 0, st, a, cp0_1, cq_1, cov, cross, mdiff, terms);
    }
    hipLaunchKernelGGL((wv::row_sums_kernel<T>), dim3((unsigned)B), block, 0, st, B, Tn, static_cast<const T*>(terms), T(0.5), out);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
}  // namespace
int wave_ssm_kl_f64(long B, long Tn, int d, const double* cp0_1, const double* cq_1, const double* cp0_2, const double* a_2,
                    const double* cq_2, const double* cov, const double* cross, const double* mdiff, double* out, void* ws,
                    size_t ws_bytes, hipStream_t st) {
    return wave_kl<double>(B, Tn, d, cp0_1, cq_1, cp0_2, a_2, cq_2, cov, cross, mdiff, out, ws, ws_bytes, st);
}
int wave_ssm_kl_f32(long B, long Tn, int d, const float* cp0_1, const float* cq_1, const float* cp0_2, const float* a_2,
                    const float* cq_2, const float* cov, const float* cross, const float* mdiff, float* out, void* ws, size_t ws_bytes,
                    hipStream_t st) {
    return wave_kl<float>(B, Tn, d, cp0_1, cq_1, cp0_2, a_2, cq_2, cov, cross, mdiff, out, ws, ws_bytes, st);
}

// ---- kl_divergence at 16 <= d <= 32 in one walk per (series, chunk): wave_kl_walk_kernel (mf_wave_ops.hpp) -----------------------
void wave_marg_partition(long B, long nt, int d, int elem_size, long& P, long& L);
namespace {
template <typename T> size_t wave_kl_fused_need(long B, long n, int d, long P) {
    // q1: wM, wN, bP [B, P, d, d], wv, bm [B, P, d]; q2: wM [B, P, d, d], wv, m_in [B, P, d]; terms [B, n]
    return (size_t(B) * P * (4 * size_t(d) * d + 4 * size_t(d)) + size_t(B) * n) * sizeof(T) + 256;
}
template <typename T>
int wave_kl_fused(long B, long n, int d, const T* mu0_1, const T* cp0_1, const T* a_1, const T* b_1, const T* cq_1, const T* mu0_2,
                  const T* cp0_2, const T* a_2, const T* b_2, const T* cq_2, T* out, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!wave_covers(d, 1) || B <= 0 || n < 2) return -101;
    long P = 1, L = n - 1;
    wave_marg_partition(B, n - 1, d, (int)sizeof(T), P, L);
    if (!ws || ws_bytes < wave_kl_fused_need<T>(B, n, d, P)) return -15;
    const size_t blk = size_t(B) * P * d * d, vec = size_t(B) * P * d;
    T* p = static_cast<T*>(ws);
    wv::MargArgs<T> m1{B, n, d, mu0_1, cp0_1, a_1, b_1, cq_1, /*omean (non-NULL: the means are wanted)*/ p, nullptr, nullptr};
    m1.P = P; m1.L = L;
    m1.wM = p; m1.wN = p + blk; m1.bP = p + 2 * blk; m1.wv = p + 3 * blk; m1.bm = m1.wv + vec;
    T* q = m1.bm + vec;
    wv::MeansArgs<T> m2{B, n, d, a_2, nullptr, P, L, q, q + blk, q + blk + vec, mu0_2, b_2};
    T* terms = q + blk + 2 * vec;
    const wv::KlWalkArgs<T> ka{B, n, d, mu0_1, cp0_1, a_1, b_1, cq_1, mu0_2, cp0_2, a_2, b_2, cq_2, P, L, m1.bP, m1.bm, m2.m_in, terms};
    const dim3 chunks((unsigned)(B * P)), series((unsigned)B), block(64);
    auto go = [&](auto ntag) {
        constexpr int NT = decltype(ntag)::value;
        if (P > 1) {
            hipLaunchKernelGGL((wv::wave_marg_up_kernel<T, NT>), chunks, block, 0, st, m1);
            hipLaunchKernelGGL((wv::wave_marg_boundary_kernel<T, NT>), series, block, 0, st, m1);
            hipLaunchKernelGGL((wv::wave_means_up_kernel<T, NT>), dim3((unsigned)(B * (P - 1))), block, 0, st, m2);
            hipLaunchKernelGGL((wv::wave_means_boundary_kernel<T, NT>), series, block, 0, st, m2);
        }
        hipLaunchKernelGGL((wv::wave_kl_walk_kernel<T, NT>), chunks, block, 0, st, ka);
    };
    if (d <= 16) go(std::integral_constant<int, 1>{}); else go(std::integral_constant<int, 2>{});
    hipLaunchKernelGGL((wv::row_sums_kernel<T>), series, block, 0, st, B, n, static_cast<const T*>(terms), T(0.5), out);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
}  // namespace
size_t wave_ssm_kl_fused_ws(long B, long n, int d, int elem_size) {
    if (!wave_covers(d, 1) || B <= 0 || n < 2) return 0;
    long P = 1, L = n - 1;
    wave_marg_partition(B, n - 1, d, elem_size, P, L);
    return elem_size == 8 ? wave_kl_fused_need<double>(B, n, d, P) : wave_kl_fused_need<float>(B, n, d, P);
}
int wave_ssm_kl_fused_f64(long B, long n, int d, const double* mu0_1, const double* cp0_1, const double* a_1, const double* b_1,
                          const double* cq_1, const double* mu0_2, const double* cp0_2, const double* a_2, const double* b_2,
                          const double* cq_2, double* out, void* ws, size_t ws_bytes, hipStream_t st) {
    return wave_kl_fused<double>(B, n, d, mu0_1, cp0_1, a_1, b_1, cq_1, mu0_2, cp0_2, a_2, b_2, cq_2, out, ws, ws_bytes, st);
}
int wave_ssm_kl_fused_f32(long B, long n, int d, const float* mu0_1, const float* cp0_1, const float* a_1, const float* b_1,
                          const float* cq_1, const float* mu0_2, const float* cp0_2, const float* a_2, const float* b_2,
                          const float* cq_2, float* out, void* ws, size_t ws_bytes, hipStream_t st) {
    return wave_kl_fused<float>(B, n, d, mu0_1, cp0_1, a_1, b_1, cq_1, mu0_2, cp0_2, a_2, b_2, cq_2, out, ws, ws_bytes, st);
}

// ---- LowerTriangularBlockTriDiagonal.solve for 16 <= d <= 32: the time axis serially inside a wavefront (wave_solve_kernel) -----------
namespace {
// Chunks of the time-partitioned solve: enough (series, chunk) pairs for ~3 wavefronts of the map pass per SIMD, chunks of at least
// 32 blocks (the map pass is ~4 x a block of the walk: a triangular inversion and two products against two matrix-vector products).
// A walk of n dependent ~2 us block steps is what is being cut: B = 512, T = 1000, d = 16: 2.0 ms unpartitioned.
// Both passes read the factor: at d = 16, B = 512 the partitioned form moves 2 x 2.2 GB in 1.4 ms where the walk moved 2.2 GB in
// 2.0 ms (latency-bound: 128 wavefronts), with fewer series the gain grows with the number of chunks (B = 64: 8 x).  At d > 16 a
// block of the map pass costs ~4 x a block of the walk and takes the register file of a SIMD: only where its wavefronts have a SIMD
// each (B (P - 1) <= 1024) and at least four chunks come out.
void wave_solve_partition(long Bl, long Br, int d, long n, long& P, long& Lc) {
    P = 1; Lc = n;
    if (Bl != Br || n < 128) return;
    long want = (d <= 16 ? 3072 : 1024) / (Br > 0 ? Br : 1);
    if (want > n / 32) want = n / 32;
    if (want > 64) want = 64;
    if (want < (d <= 16 ? 2 : 4)) return;
    Lc = (n + want - 1) / want;
    P = (n + Lc - 1) / Lc;
    if (P < 2) { P = 1; Lc = n; }
}
template <typename T>
int wave_solve(long Bl, long Br, long n, int d, const T* ldiag, const T* lsub, const T* rhs, T* out, int transpose, void* ws,
               size_t ws_bytes, hipStream_t st) {
    if (!wave_covers(d, 1)) return -101;
    long P, Lc;
    wave_solve_partition(Bl, Br, d, n, P, Lc);
    static const bool nopart = getenv("MF_WAVE_SOLVE_NOPART") != nullptr;
    const size_t need = size_t(Br) * P * (size_t(d) * d + 2 * d) * sizeof(T);
    if (nopart || !lsub || !ws || ws_bytes < need) P = 1;
    // the walk is n dependent block steps of ~0.4 us: a long chain of few series belongs to the time-partitioned engine
    if (P == 1 && n > 2000 && Br < 64) return -101;
    wv::SolveArgs<T> a{Bl, Br, n, d, ldiag, lsub, rhs, out, P, Lc, nullptr, nullptr, nullptr};
    const dim3 block(64);
    if (P > 1) {
        a.wM = static_cast<T*>(ws);
        a.wv = a.wM + size_t(Br) * P * d * d;
        a.zin = a.wv + size_t(Br) * P * d;
        const dim3 gup((unsigned)(Br * (P - 1))), gb((unsigned)Br);
#define MF_UP(NT_)                                                                                                  \
        {                                                                                                           \
            if (transpose) hipLaunchKernelGGL((wv::wave_solve_up_kernel<T, NT_, true>), gup, block, 0, st, a);       \
            else hipLaunchKernelGGL((wv::wave_solve_up_kernel<T, NT_, false>), gup, block, 0, st, a);                \
            hipLaunchKernelGGL((wv::wave_solve_boundary_kernel<T, NT_>), gb, block, 0, st, a);                       \
        }
        if (d <= 16) MF_UP(1) else MF_UP(2)
#undef MF_UP
        const long units = Br * P;
        if (d <= 16) {
            const dim3 grid((unsigned)((units + 3) / 4));
            if (transpose) hipLaunchKernelGGL((wv::wave_solve_kernel<T, 1, true, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((wv::wave_solve_kernel<T, 1, false, true>), grid, block, 0, st, a);
        } else {
            const dim3 grid((unsigned)((units + 1) / 2));
            if (transpose) hipLaunchKernelGGL((wv::wave_solve_kernel<T, 2, true, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((wv::wave_solve_kernel<T, 2, false, true>), grid, block, 0, st, a);
        }
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    if (d <= 16) {
        const dim3 grid((unsigned)((Br + 3) / 4));
        if (transpose) hipLaunchKernelGGL((wv::wave_solve_kernel<T, 1, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((wv::wave_solve_kernel<T, 1, false>), grid, block, 0, st, a);
    } else {
        const dim3 grid((unsigned)((Br + 1) / 2));
        if (transpose) hipLaunchKernelGGL((wv::wave_solve_kernel<T, 2, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((wv::wave_solve_kernel<T, 2, false>), grid, block, 0, st, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
}  // namespace
namespace {
template <typename T>
int wave_means_bnd(long Bl, long Br, long n, int d, const T* A, const T* offs, void* ws, size_t ws_bytes, long* Pout, long* Lout,
                   const T** m_in, hipStream_t st) {
    *Pout = 1; *Lout = n; *m_in = nullptr;
    if (!wave_covers(d, 1) || n < 2) return 0;
    long P, Lc;
    // the solve's rule over the n - 1 transitions (the walk is one dependent ~1.1 us block step after another: 1.15 ms at T = 1000)
    wave_solve_partition(Bl, Br, d, n - 1, P, Lc);
    static const bool nopart = getenv("MF_WAVE_SOLVE_NOPART") != nullptr;
    if (nopart || P < 2 || !ws || ws_bytes < size_t(Br) * P * (size_t(d) * d + 2 * d) * sizeof(T)) return 0;
    wv::MeansArgs<T> a{Br, n, d, A, offs, P, Lc, static_cast<T*>(ws), nullptr, nullptr};
    a.wv = a.wM + size_t(Br) * P * d * d;
    a.m_in = a.wv + size_t(Br) * P * d;
    const dim3 block(64), gup((unsigned)(Br * (P - 1))), gb((unsigned)Br);
    if (d <= 16) {
        hipLaunchKernelGGL((wv::wave_means_up_kernel<T, 1>), gup, block, 0, st, a);
        hipLaunchKernelGGL((wv::wave_means_boundary_kernel<T, 1>), gb, block, 0, st, a);
    } else {
        hipLaunchKernelGGL((wv::wave_means_up_kernel<T, 2>), gup, block, 0, st, a);
        hipLaunchKernelGGL((wv::wave_means_boundary_kernel<T, 2>), gb, block, 0, st, a);
    }
    if (hipGetLastError() != hipSuccess) return -1000;
    *Pout = P; *Lout = Lc; *m_in = a.m_in;
    return 0;
}
}  // namespace
int wave_means_boundaries_f64(long Bl, long Br, long n, int d, const double* A, const double* offs, void* ws, size_t ws_bytes, long* P,
                              long* Lc, const double** m_in, hipStream_t st) {
    return wave_means_bnd<double>(Bl, Br, n, d, A, offs, ws, ws_bytes, P, Lc, m_in, st);
}
int wave_means_boundaries_f32(long Bl, long Br, long n, int d, const float* A, const float* offs, void* ws, size_t ws_bytes, long* P,
                              long* Lc, const float** m_in, hipStream_t st) {
    return wave_means_bnd<float>(Bl, Br, n, d, A, offs, ws, ws_bytes, P, Lc, m_in, st);
}
size_t wave_btd_solve_ws(long Bl, long Br, long n, int d, int elem_size) {
    if (!wave_covers(d, 1)) return 0;
    long P, Lc, Pm, Lm;
    wave_solve_partition(Bl, Br, d, n, P, Lc);
    wave_solve_partition(Bl, Br, d, n > 1 ? n - 1 : 1, Pm, Lm);                   // marginal_means (wave_means_bnd) shares the workspace
    if (Pm > P) P = Pm;
    return P > 1 ? size_t(Br) * P * (size_t(d) * d + 2 * d) * elem_size : 0;
}
int wave_btd_solve_f64(long Bl, long Br, long n, int d, const double* ldiag, const double* lsub, const double* rhs, double* out,
                       int transpose, void* ws, size_t ws_bytes, hipStream_t st) {
    return wave_solve<double>(Bl, Br, n, d, ldiag, lsub, rhs, out, transpose, ws, ws_bytes, st);
}
int wave_btd_solve_f32(long Bl, long Br, long n, int d, const float* ldiag, const float* lsub, const float* rhs, float* out, int transpose,
                       void* ws, size_t ws_bytes, hipStream_t st) {
    return wave_solve<float>(Bl, Br, n, d, ldiag, lsub, rhs, out, transpose, ws, ws_bytes, st);
}


// ---- cholesky / upper_diagonal_lower (+ posterior chain) / block_diagonal_of_inverse: one wavefront per series ---------------------
namespace {
// the walk is n dependent block steps of 2 - 5 us; few long series belong to the time-partitioned engine
inline bool wave_serial_pays(long B, long n) { return B >= 64 || n <= 256; }
}  // namespace
#define MF_WAVE_FACT(KERNEL, ARGS, WAVES)                                                                              \
    do {                                                                                                               \
        if (d <= 16) hipLaunchKernelGGL((wv::KERNEL<T, 1>), dim3((unsigned)(WAVES)), dim3(64), 0, st, ARGS);             \
        else hipLaunchKernelGGL((wv::KERNEL<T, 2>), dim3((unsigned)(WAVES)), dim3(64), 0, st, ARGS);                     \
        return hipGetLastError() == hipSuccess ? 0 : -1000;                                                            \
    } while (0)
void wave_udl_partition(long B, long n, int d, int elem_size, long& P, long& L);
size_t wave_udl_ws(long B, long n, int d, int elem_size);
template <typename T> int wave_btd_cholesky(long B, long n, int d, const T* diag, const T* sub, T* ldiag, T* lsub, void* ws, size_t ws_bytes,
                                            int* info, hipStream_t st) {
    if (!wave_covers(d, 1) || B <= 0 || n <= 0) return -101;
    if (!sub) {   // block diagonal: every block is its own series
        const wv::FactArgs<T> a{B * n, 1, d, diag, nullptr, ldiag, nullptr, nullptr, nullptr, nullptr, info};
        MF_WAVE_FACT(wave_cholesky_kernel, a, B * n);
    }
    long P = 1, L = n;
    wave_udl_partition(B, n, d, (int)sizeof(T), P, L);
    if (P > 1 && ws && ws_bytes >= wave_udl_ws(B, n, d, (int)sizeof(T))) {
        // partitioned in time: up-sweep with a spike, boundary pivots, emit (the scheme of the posterior chain below)
        wv::FactArgs<T> a{B, n, d, diag, sub, ldiag, lsub, nullptr, nullptr, nullptr, info};
        const size_t blk = size_t(B) * P * d * d;
        T* p = static_cast<T*>(ws);
        a.P = P; a.L = L;
        a.rDv = p; a.rGU = p + blk; a.rF = p + 2 * blk; a.bSig = p + 3 * blk;
        const dim3 chunks((unsigned)(B * P)), series((unsigned)B), block(64);
        if (d <= 16) {
            hipLaunchKernelGGL((wv::wave_chol_up_kernel<T, 1>), chunks, block, 0, st, a);
            hipLaunchKernelGGL((wv::wave_udl_boundary_kernel<T, 1>), series, block, 0, st, a);
            hipLaunchKernelGGL((wv::wave_cholesky_kernel<T, 1, true>), chunks, block, 0, st, a);
        } else {
            hipLaunchKernelGGL((wv::wave_chol_up_kernel<T, 2>), chunks, block, 0, st, a);
            hipLaunchKernelGGL((wv::wave_udl_boundary_kernel<T, 2>), series, block, 0, st, a);
            hipLaunchKernelGGL((wv::wave_cholesky_kernel<T, 2, true>), chunks, block, 0, st, a);
        }
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    if (!wave_serial_pays(B, n)) return -101;
    const wv::FactArgs<T> a{B, n, d, diag, sub, ldiag, lsub, nullptr, nullptr, nullptr, info};
    MF_WAVE_FACT(wave_cholesky_kernel, a, B);
}
// time partition of the posterior chain (wave_udl_up_kernel / _boundary_ / wave_udl_kernel<.., PART>): enough chunks for ~2 wavefronts
// per SIMD over the chip, chunks of at least 16 blocks, at most 256 per series (the boundary pass walks them serially)
void wave_udl_partition(long B, long n, int d, int elem_size, long& P, long& L) {
    static const long force = [] { const char* e = mf_knob("MF_WAVE_UDL_CHUNKS"); return e ? std::atol(e) : 0L; }();
    // wavefronts the chip holds at once at these kernels' registers: one tile per matrix 3 per SIMD (fp64) / 4 (fp32), 2 x 2 tiles 1 / 2
    const long target = 1024L * (d <= 16 ? (elem_size == 8 ? 3 : 4) : (elem_size == 8 ? 1 : 2));
    long want = force > 0 ? force : (target + B - 1) / B;
    if (want > 256) want = 256;
    if (want > n / 16) want = n / 16;
    if (want < 1) want = 1;
    L = (n + want - 1) / want;
    P = (n + L - 1) / L;
}
size_t wave_udl_ws(long B, long n, int d, int elem_size) {
    if (!wave_covers(d, 1) || B <= 0 || n <= 0) return 0;
    long P, L;
    wave_udl_partition(B, n, d, elem_size, P, L);
    return P > 1 ? size_t(B) * P * (4 * size_t(d) * d + 3 * size_t(d)) * elem_size + 256 : 0;
}
template <typename T> int wave_btd_udl(long B, long n, int d, const T* diag, const T* sub, T* ut, T* chol_d, const T* eta, T* m_post,
                                       T* chol_dinv, void* ws, size_t ws_bytes, int* info, hipStream_t st) {
    if (!wave_covers(d, 1) || B <= 0 || n <= 0) return -101;
    wv::FactArgs<T> a{B, n, d, diag, sub, ut, chol_d, eta, m_post, chol_dinv, info};
    const dim3 block(64);
    long P = 1, L = n;
    if (sub) wave_udl_partition(B, n, d, (int)sizeof(T), P, L);
    if (P > 1 && ws && ws_bytes >= wave_udl_ws(B, n, d, (int)sizeof(T))) {
        const size_t blk = size_t(B) * P * d * d, vec = size_t(B) * P * d;
        T* p = static_cast<T*>(ws);
        a.P = P; a.L = L;
        a.rDv = p; a.rGU = p + blk; a.rF = p + 2 * blk; a.bSig = p + 3 * blk;
        a.rtv = p + 4 * blk; a.rgU = a.rtv + vec; a.bx = a.rgU + vec;
        const dim3 chunks((unsigned)(B * P)), series((unsigned)B);
        auto go = [&](auto nt) {
            constexpr int NT = decltype(nt)::value;
            hipLaunchKernelGGL((wv::wave_udl_up_kernel<T, NT>), chunks, block, 0, st, a);
            hipLaunchKernelGGL((wv::wave_udl_boundary_kernel<T, NT>), series, block, 0, st, a);
            if (eta) hipLaunchKernelGGL((wv::wave_udl_kernel<T, NT, true, true>), chunks, block, 0, st, a);
            else hipLaunchKernelGGL((wv::wave_udl_kernel<T, NT, false, true>), chunks, block, 0, st, a);
        };
        if (d <= 16) go(std::integral_constant<int, 1>{}); else go(std::integral_constant<int, 2>{});
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    if (!wave_serial_pays(B, n)) return -101;
    const dim3 grid((unsigned)B);
    if (eta) {
        if (d <= 16) hipLaunchKernelGGL((wv::wave_udl_kernel<T, 1, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((wv::wave_udl_kernel<T, 2, true>), grid, block, 0, st, a);
    } else {
        if (d <= 16) hipLaunchKernelGGL((wv::wave_udl_kernel<T, 1, false>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((wv::wave_udl_kernel<T, 2, false>), grid, block, 0, st, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
template <typename T> int wave_btd_diag_of_inverse(long B, long n, int d, const T* ldiag, const T* lsub, T* odiag, T* osub, void* ws,
                                                   size_t ws_bytes, hipStream_t st) {
    if (!wave_covers(d, 1) || B <= 0 || n <= 0) return -101;
    if (!lsub) {
        const wv::FactArgs<T> a{B * n, 1, d, ldiag, nullptr, odiag, nullptr, nullptr, nullptr, nullptr, nullptr};
        MF_WAVE_FACT(wave_inverse_blocks_kernel, a, B * n);
    }
    long P = 1, L = n;
    wave_udl_partition(B, n, d, (int)sizeof(T), P, L);
    // (composing the maps doubles the work of the walk: two chunks per series do not pay - 7.5 -> 8.7 ms at d = 32, B = 512, T = 1000)
    if (P > 2 && ws && ws_bytes >= wave_udl_ws(B, n, d, (int)sizeof(T))) {
        // partitioned in time: composed congruence maps per chunk, Sigma above every chunk, emit
        wv::FactArgs<T> a{B, n, d, ldiag, lsub, odiag, osub, nullptr, nullptr, nullptr, nullptr};
        const size_t blk = size_t(B) * P * d * d;
        T* p = static_cast<T*>(ws);
        a.P = P; a.L = L;
        a.rDv = p; a.rGU = p + blk; a.bSig = p + 2 * blk;
        const dim3 chunks((unsigned)(B * P)), series((unsigned)B), block(64);
        auto go = [&](auto nt) {
            constexpr int NT = decltype(nt)::value;
            hipLaunchKernelGGL((wv::wave_inv_up_kernel<T, NT>), chunks, block, 0, st, a);
            hipLaunchKernelGGL((wv::wave_inv_boundary_kernel<T, NT>), series, block, 0, st, a);
            hipLaunchKernelGGL((wv::wave_inverse_blocks_kernel<T, NT, true>), chunks, block, 0, st, a);
        };
        if (d <= 16) go(std::integral_constant<int, 1>{}); else go(std::integral_constant<int, 2>{});
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    if (!wave_serial_pays(B, n)) return -101;
    const wv::FactArgs<T> a{B, n, d, ldiag, lsub, odiag, osub, nullptr, nullptr, nullptr, nullptr};
    MF_WAVE_FACT(wave_inverse_blocks_kernel, a, B);
}
// the forward recursion of the moments needs far fewer registers than the factorisations (58 per lane at one fp64 tile per matrix, 226
// at 2 x 2): more wavefronts fit, more chunks pay
void wave_marg_partition(long B, long nt, int d, int elem_size, long& P, long& L) {
    const long target = 1024L * (d <= 16 ? 6 : (elem_size == 8 ? 2 : 4));
    long want = (target + B - 1) / B;
    if (want > 256) want = 256;
    if (want > nt / 16) want = nt / 16;
    if (want < 1) want = 1;
    L = (nt + want - 1) / want;
    P = (nt + L - 1) / L;
}
size_t wave_marg_ws(long B, long n, int d, int elem_size) {
    if (!wave_covers(d, 1) || B <= 0 || n <= 1) return 0;
    long P, L;
    wave_marg_partition(B, n - 1, d, elem_size, P, L);
    return P > 1 ? size_t(B) * P * (3 * size_t(d) * d + 2 * size_t(d)) * elem_size + 256 : 0;
}
template <typename T> int wave_ssm_marginals(long B, long n, int d, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ,
                                             T* omean, T* ocov, T* osub, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!wave_covers(d, 1) || B <= 0 || n <= 0) return -101;
    wv::MargArgs<T> a{B, n, d, mu0, cholP0, A, b, cholQ, omean, ocov, osub};
    long P = 1, L = n;
    if (n > 1) wave_marg_partition(B, n - 1, d, (int)sizeof(T), P, L);
    if (P > 1 && ws && ws_bytes >= wave_marg_ws(B, n, d, (int)sizeof(T))) {
        const size_t blk = size_t(B) * P * d * d, vec = size_t(B) * P * d;
        T* p = static_cast<T*>(ws);
        a.P = P; a.L = L;
        a.wM = p; a.wN = p + blk; a.bP = p + 2 * blk; a.wv = p + 3 * blk; a.bm = a.wv + vec;
        const dim3 chunks((unsigned)(B * P)), series((unsigned)B), block(64);
        auto go = [&](auto nt) {
            constexpr int NT = decltype(nt)::value;
            hipLaunchKernelGGL((wv::wave_marg_up_kernel<T, NT>), chunks, block, 0, st, a);
            hipLaunchKernelGGL((wv::wave_marg_boundary_kernel<T, NT>), series, block, 0, st, a);
            hipLaunchKernelGGL((wv::wave_marginals_kernel<T, NT, true>), chunks, block, 0, st, a);
        };
        if (d <= 16) go(std::integral_constant<int, 1>{}); else go(std::integral_constant<int, 2>{});
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    if (!wave_serial_pays(B, n)) return -101;
    MF_WAVE_FACT(wave_marginals_kernel, a, B);
}
#undef MF_WAVE_FACT
// the local step of the log-likelihood's gradient (mf_wave_grad.hpp): one wavefront per (series, time point)
template <typename T>
int wave_kf_grad(long B, long Tn, int d, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H, const T* y,
                 const T* Rinv, int rinv_per_step, const T* mean, const T* cov, const T* cross, const T* w, T* g_mu0, T* g_cholP0, T* g_A,
                 T* g_b, T* g_cholQ, T* g_H, T* g_y, T* g_om, hipStream_t st) {
    if (!wave_covers(d, H ? m : 1) || B <= 0 || Tn <= 0) return -101;
    const wv::WvGradArgs<T> a{B, Tn, d, H ? m : 1, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, mean, cov, cross, w,
                              g_mu0, g_cholP0, g_A, g_b, g_cholQ, g_H, g_y, g_om};
    const dim3 grid((unsigned)(B * Tn)), block(64);
    const bool m1 = !H || m == 1;
    if (d <= 16) {
        if (m1) hipLaunchKernelGGL((wv::wave_kf_grad_kernel<T, 1, 1>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((wv::wave_kf_grad_kernel<T, 1, wv::WV_MAXM>), grid, block, 0, st, a);
    } else {
        if (m1) hipLaunchKernelGGL((wv::wave_kf_grad_kernel<T, 2, 1>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((wv::wave_kf_grad_kernel<T, 2, wv::WV_MAXM>), grid, block, 0, st, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
template int wave_kf_grad<double>(long, long, int, int, const double*, const double*, const double*, const double*, const double*, const double*,
                                  const double*, const double*, int, const double*, const double*, const double*, const double*, double*, double*,
                                  double*, double*, double*, double*, double*, double*, hipStream_t);
template int wave_kf_grad<float>(long, long, int, int, const float*, const float*, const float*, const float*, const float*, const float*,
                                 const float*, const float*, int, const float*, const float*, const float*, const float*, float*, float*, float*,
                                 float*, float*, float*, float*, float*, hipStream_t);
template int wave_ssm_marginals<double>(long, long, int, const double*, const double*, const double*, const double*, const double*, double*,
                                        double*, double*, void*, size_t, hipStream_t);
template int wave_ssm_marginals<float>(long, long, int, const float*, const float*, const float*, const float*, const float*, float*, float*,
                                       float*, void*, size_t, hipStream_t);
template int wave_btd_cholesky<double>(long, long, int, const double*, const double*, double*, double*, void*, size_t, int*, hipStream_t);
template int wave_btd_cholesky<float>(long, long, int, const float*, const float*, float*, float*, void*, size_t, int*, hipStream_t);
template int wave_btd_udl<double>(long, long, int, const double*, const double*, double*, double*, const double*, double*, double*, void*,
                                  size_t, int*, hipStream_t);
template int wave_btd_udl<float>(long, long, int, const float*, const float*, float*, float*, const float*, float*, float*, void*, size_t, int*,
                                 hipStream_t);
template int wave_btd_diag_of_inverse<double>(long, long, int, const double*, const double*, double*, double*, void*, size_t, hipStream_t);
template int wave_btd_diag_of_inverse<float>(long, long, int, const float*, const float*, float*, float*, void*, size_t, hipStream_t);

}  // namespace mf
