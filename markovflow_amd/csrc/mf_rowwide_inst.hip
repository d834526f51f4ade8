// The Kalman log-likelihood for 10 <= d <= 15 with the ROW kernels of mf_row.hpp (one 16-lane DPP row per (series, chunk),
// matrix rows across the lanes, lane d holds the vectors): compile with -DMF_D=<d>.  Only the log-likelihood and its reduction
// levels are instantiated here - every other operator of these dimensions runs on the LDS-tile / MFMA engine (mf_big.hpp), whose
// log-likelihood needs a whole workgroup per chunk and is 20 - 45 times slower at d = 10 ... 15 with hundreds of series
// (profiles/r03_sweep_d.txt).  The register-resident instantiations of mf_inst.hip stop at d = 9: their lane-per-chunk kernels do
// not fit a lane's registers beyond it.
#ifndef MF_D
#error "compile with -DMF_D=<state dimension>"
#endif
#include "mf_kernels.hpp"
#include "mf_row.hpp"
#include "mf_launch.hpp"

#include <type_traits>

namespace mf {
namespace {

constexpr int D = MF_D;
static_assert(D >= 10 && D + 1 <= 16, "wide row kernels: 10 <= d <= 15");
constexpr long ROW_RED_CHUNK = 6, ROW_RED_FINAL = 4;
constexpr long RED_ELEMS = 3 * D * D + 2 * D + 1;

inline long cdiv(long a, long b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }
template <typename T> size_t red_bytes(long B, long n) { return align_up(size_t(B) * n * RED_ELEMS * sizeof(T)); }
template <typename T> RedSys<T> carve(char*& p, long B, long n) {
    RedSys<T> r;
    T* base = reinterpret_cast<T*>(p);
    const long nb = B * n;
    r.Dv = base;
    r.GU = r.Dv + nb * D * D;
    r.F = r.GU + nb * D * D;
    r.tv = r.F + nb * D * D;
    r.gU = r.tv + nb * D;
    r.sc = r.gU + nb * D;
    r.n = n;
    r.f_stride = n;
    r.f_off = 0;
    p += red_bytes<T>(B, n);
    return r;
}
template <typename T> size_t levels_ws(long B, long P) {
    size_t total = red_bytes<T>(B, P);
    long n = P;
    while (n > ROW_RED_FINAL) {
        n = cdiv(n, ROW_RED_CHUNK);
        total += red_bytes<T>(B, n);
    }
    return total;
}
// rows that fill the chip: four per wavefront, as many wavefronts per SIMD as the kernel's registers allow (mf_row.hpp)
inline long target_rows() { return 256L * 4 * row::row_waves_per_simd(D) * 4; }
inline long plan_chunks(long B, long Tn, long chunks) {
    long P = chunks > 0 ? (chunks > Tn ? Tn : chunks) : cdiv(target_rows(), B);
    if (chunks <= 0) {
        const long maxP = Tn / 4 > 0 ? Tn / 4 : 1;
        if (P > maxP) P = maxP;
    }
    return P < 1 ? 1 : P;
}

template <typename T> bool usable(long B, long Tn, int m, long chunks) {
    return Tn >= 2 && m >= 1 && m <= MF_MAXM && row::row_offsets_fit(Tn, plan_chunks(B, Tn, chunks), D, m, (int)sizeof(T));
}
template <typename T> size_t kf_loglik_ws(long B, long Tn, long chunks) { return levels_ws<T>(B, plan_chunks(B, Tn, chunks)); }

template <typename T>
int kf_loglik(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
              const T* y, const T* Rinv, int rinv_per_step, T add_const, T* out, void* ws, size_t ws_bytes, int* info, long chunks,
              hipEvent_t ev0, hipEvent_t ev1, hipStream_t st) {
    if (!usable<T>(B, Tn, m, chunks)) return -100;
    const long P = plan_chunks(B, Tn, chunks);
    if (ws == nullptr || ws_bytes < levels_ws<T>(B, P)) return -15;
    KfArgs<T> a{B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, info, 0, nullptr};
    char* p = static_cast<char*>(ws);
    RedSys<T> cur = carve<T>(p, B, P);
    const dim3 rgrid((unsigned)cdiv(B * P, 4)), block(64);
    if (ev0) (void)hipEventRecord(ev0, st);
    auto launch = [&](auto mtag) {
        constexpr int M = decltype(mtag)::value;
        if (rinv_per_step) hipLaunchKernelGGL((row::kf_row_kernel<T, D, M, true>), rgrid, block, 0, st, a, cur);
        else hipLaunchKernelGGL((row::kf_row_kernel<T, D, M, false>), rgrid, block, 0, st, a, cur);
    };
    using std::integral_constant;
    if (m == 1) launch(integral_constant<int, 1>{});
    else if (m == 2) launch(integral_constant<int, 2>{});
    else if (m == 3) launch(integral_constant<int, 3>{});
    else launch(integral_constant<int, 4>{});
    if (ev1) (void)hipEventRecord(ev1, st);
    while (cur.n > ROW_RED_FINAL) {
        const long Pn = cdiv(cur.n, ROW_RED_CHUNK);
        RedSys<T> nxt = carve<T>(p, B, Pn);
        hipLaunchKernelGGL((row::red_row_kernel<T, D>), dim3((unsigned)cdiv(B * Pn, 4)), block, 0, st, cur, nxt, B, Pn, info);
        cur = nxt;
    }
    hipLaunchKernelGGL((row::red_row_final_kernel<T, D>), dim3((unsigned)cdiv(B, 4)), block, 0, st, cur, B, add_const, out, info);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T> const RowWideTable<T>* table() {
    static const RowWideTable<T> t = {&usable<T>, &kf_loglik_ws<T>, &kf_loglik<T>};
    return &t;
}

}  // namespace

#define MF_CAT2(a, b) a##b
#define MF_CAT(a, b) MF_CAT2(a, b)
const RowWideTable<float>* MF_CAT(rowwide_f32_d, MF_D)() { return table<float>(); }
const RowWideTable<double>* MF_CAT(rowwide_f64_d, MF_D)() { return table<double>(); }

}  // namespace mf
