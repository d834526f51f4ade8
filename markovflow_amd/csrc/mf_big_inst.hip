// Large-state-dimension (10 <= d <= 64, fp32) instantiations of the Kalman log-likelihood (mf_big.hpp), one padded
// tile size per template instance, and the host-side launcher the C ABI dispatches to.
#include "mf_big.hpp"
#include "mf_launch.hpp"

#include <cstdlib>

namespace mf {
namespace {

constexpr long BIG_RED_CHUNK = 8, BIG_RED_FINAL = 8;
inline long cdivl(long a, long b) { return (a + b - 1) / b; }
inline size_t align_up_big(size_t x) { return (x + 255) & ~size_t(255); }
inline long red_elems(int d) { return 3L * d * d + 2L * d + 1; }
inline size_t red_bytes_big(long B, long n, int d) { return align_up_big(size_t(B) * n * red_elems(d) * sizeof(float)); }

RedSys<float> carve_big(char*& p, long B, long n, int d) {
    RedSys<float> r;
    float* base = reinterpret_cast<float*>(p);
    const long nb = B * n, dd = long(d) * d;
    r.Dv = base;
    r.GU = r.Dv + nb * dd;
    r.F = r.GU + nb * dd;
    r.tv = r.F + nb * dd;
    r.gU = r.tv + nb * d;
    r.sc = r.gU + nb * d;
    r.n = n;
    r.f_stride = n;
    r.f_off = 0;
    p += red_bytes_big(B, n, d);
    return r;
}

// chunks per series: enough workgroups for two rounds over the 256 CUs, chunks of at least 4 transitions
inline void big_partition(long B, long Tn, long chunks, long& P, long& L) {
    static const long target = [] { const char* e = std::getenv("MF_BIG_TARGET_WGS"); return e ? std::atol(e) : 512L; }();
    const long nt = Tn - 1;
    if (nt < 1) { P = 1; L = 1; return; }
    long want = chunks > 0 ? chunks : cdivl(target, B);
    const long maxP = nt / 4 > 0 ? nt / 4 : 1;
    if (want > maxP) want = maxP;
    if (want < 1) want = 1;
    L = cdivl(nt, want);
    P = cdivl(nt, L);
}

template <int DP> int launch_big(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A,
                                 const float* b, const float* cholQ, const float* H, const float* y, const float* Rinv,
                                 int rinv_per_step, float add_const, float* out, void* ws, int* info, long P, long L,
                                 hipEvent_t ev0, hipEvent_t ev1, hipStream_t st) {
    using SM = big::Smem<DP>;
    static const bool attr_ok = [] {
        bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&big::big_kf_chunk_kernel<DP>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SM::BYTES) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&big::big_red_kernel<DP, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, SM::BYTES) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&big::big_red_kernel<DP, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, SM::BYTES) == hipSuccess;
        return ok;
    }();
    if (!attr_ok) return -1000;
    char* p = static_cast<char*>(ws);
    RedSys<float> cur = carve_big(p, B, P, d);
    big::BigArgs a{B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, info};
    if (ev0) (void)hipEventRecord(ev0, st);
    hipLaunchKernelGGL((big::big_kf_chunk_kernel<DP>), dim3((unsigned)(B * P)), dim3(big::NTHR), SM::BYTES, st, a, cur);
    if (ev1) (void)hipEventRecord(ev1, st);
    while (cur.n > BIG_RED_FINAL) {
        const long Pn = cdivl(cur.n, BIG_RED_CHUNK);
        RedSys<float> nxt = carve_big(p, B, Pn, d);
        hipLaunchKernelGGL((big::big_red_kernel<DP, false>), dim3((unsigned)(B * Pn)), dim3(big::NTHR), SM::BYTES, st, cur,
                           nxt, B, Pn, d, 0.f, static_cast<float*>(nullptr), info);
        cur = nxt;
    }
    hipLaunchKernelGGL((big::big_red_kernel<DP, true>), dim3((unsigned)B), dim3(big::NTHR), SM::BYTES, st, cur, cur, B, 1L,
                       d, add_const, out, info);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

}  // namespace

size_t big_kf_loglik_ws(long B, long Tn, int d, long chunks) {
    long P, L;
    big_partition(B, Tn, chunks, P, L);
    size_t total = red_bytes_big(B, P, d);
    long n = P;
    while (n > BIG_RED_FINAL) {
        n = cdivl(n, BIG_RED_CHUNK);
        total += red_bytes_big(B, n, d);
    }
    return total;
}

int big_kf_loglik_f32(long B, long Tn, int d, int m, const float* mu0, const float* cholP0, const float* A, const float* b,
                      const float* cholQ, const float* H, const float* y, const float* Rinv, int rinv_per_step,
                      float add_const, float* out, void* ws, size_t ws_bytes, int* info, long chunks, hipEvent_t ev0,
                      hipEvent_t ev1, hipStream_t st) {
    if (m < 1 || m > big::MAXM_BIG) return -4;
    if (ws == nullptr || ws_bytes < big_kf_loglik_ws(B, Tn, d, chunks)) return -15;
    long P, L;
    big_partition(B, Tn, chunks, P, L);
#define MF_BIG_CASE(DP)                                                                                               \
    return launch_big<DP>(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, add_const, out, ws, info, \
                          P, L, ev0, ev1, st);
    if (d <= 16) { MF_BIG_CASE(16) }
    if (d <= 32) { MF_BIG_CASE(32) }
    if (d <= 48) { MF_BIG_CASE(48) }
    if (d <= 64) { MF_BIG_CASE(64) }
#undef MF_BIG_CASE
    return -100;
}

}  // namespace mf
