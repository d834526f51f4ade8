// Instantiates the streamed, time-partitioned posterior-chain kernels (mf_post_lds.hpp) for ONE state dimension (compile with
// -DMF_D=<d>) and both scalar types, and exports their launch table (mf_launch.hpp: PostOps).  A translation unit of its own:
// the kernels are large (two ~2 500-instruction sweeps per output count) and share nothing with mf_inst.hip but headers.
#ifndef MF_D
#error "compile with -DMF_D=<state dimension>"
#endif
#include "mf_post_lds.hpp"
#include "mf_launch.hpp"

#include <type_traits>

namespace mf {
namespace {

constexpr int D = MF_D;
inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// outputs / observation-precision forms with a streaming instantiation (the LDS image and the per-step DMA count of
// KfLdsCfg must fit, as for the log-likelihood kernel)
template <typename T, int M, bool RS> constexpr bool post_supported() { return KfLdsCfg<T, D, M, RS>::SUPPORTED; }
template <typename T> bool post_covers(int m, int per_step) {
    if (per_step) return m == 1 && post_supported<T, 1, true>();
    switch (m) {
        case 1: return post_supported<T, 1, false>();
        case 2: return post_supported<T, 2, false>();
        case 3: return post_supported<T, 3, false>();
        default: return false;
    }
}
// LDS per wavefront of the emit pass (input image + staging of the output rows); both passes share one time partition
template <typename T> int post_lds_bytes(int m, int per_step) {
    if (per_step) return PostLds<T, D, 1, true>::TOTAL;
    return m == 1 ? PostLds<T, D, 1, false>::TOTAL : (m == 2 ? PostLds<T, D, 2, false>::TOTAL : PostLds<T, D, 3, false>::TOTAL);
}

// chunks per series (P) and transitions per chunk (L): one wavefront on every SIMD the LDS image leaves room for
struct PostPlan { long P, L; };
template <typename T> PostPlan post_plan(long B, long Tn, int m, int per_step, long chunks) {
    const long nt = Tn - 1;
    int w = (160 * 1024) / post_lds_bytes<T>(m, per_step);
    w = w > 4 ? 4 : (w < 1 ? 1 : w);
    long want = chunks > 0 ? chunks : cdiv(256L * 64 * w, B);
    if (chunks <= 0) {
        const long maxP = nt / 4 > 0 ? nt / 4 : 1;      // never chunks shorter than four transitions
        if (want > maxP) want = maxP;
    }
    if (want > nt) want = nt;
    if (want < 1) want = 1;
    PostPlan pl;
    pl.L = cdiv(nt, want);
    pl.P = cdiv(nt, pl.L);
    return pl;
}

template <typename T> size_t post_ws_for(long B, long P) { return PostWs<T, D>::bytes(B, P); }
// 0 = this call is not the streamed kernels' (the caller keeps its other routes)
template <typename T> size_t post_ws(long B, long Tn, int m, int per_step, long chunks) {
    if (B < 1 || Tn < 2 || !post_covers<T>(m, per_step)) return 0;
    const PostPlan pl = post_plan<T>(B, Tn, m, per_step, chunks);
    return post_ws_for<T>(B, pl.P) + 256;
}

// passes 1-3 on the partition (P, L); from_bounds: the caller has filled the boundary states, pass 3 alone
template <typename T>
int post_launch(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
                const T* y, const T* Rinv, int rinv_per_step, T* a_post, T* mu0_post, T* b_post, T* cp0_post, T* cq_post,
                void* ws, size_t ws_bytes, int* info, long P, long L, bool from_bounds, hipEvent_t ev0, hipEvent_t ev1,
                hipStream_t st) {
    if (ws == nullptr || ws_bytes < post_ws_for<T>(B, P)) return -21;
    const PostWs<T, D> w = PostWs<T, D>::carve(ws, B, P);
    const RedSys<T> sum = w.sum;
    T* bPsi = w.bPsi;
    T* bpsi = w.bpsi;
    KfArgs<T> a{B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, info, 0, nullptr};
    const PostOut<T> po{a_post, mu0_post, b_post, cp0_post, cq_post, bPsi, bpsi};
    const dim3 grid((unsigned)cdiv(B * P, 64)), block(64);
    auto launch = [&](auto mtag, auto rtag) {
        constexpr int M = decltype(mtag)::value;
        constexpr bool RS = decltype(rtag)::value;
        if constexpr (KfLdsCfg<T, D, M, RS>::SUPPORTED) {
            constexpr int lds = KfLdsCfg<T, D, M, RS>::LDS_TOTAL, lds_emit = PostLds<T, D, M, RS>::TOTAL;
            static_assert(lds_emit <= 64 * 1024, "emit pass: LDS image + staging beyond the default dynamic-LDS limit");
            constexpr int scan_lds = PostScanLds<T, D>::BYTES;
            if (ev0) (void)hipEventRecord(ev0, st);
            if (P > 1 && !from_bounds) {
                hipLaunchKernelGGL((post_lds_kernel<T, D, M, RS, 0>), grid, block, lds, st, a, L, sum, po);
                hipLaunchKernelGGL((post_scan_kernel<T, D>), dim3((unsigned)B), block, scan_lds, st, sum, B, bPsi, bpsi,
                                   info);
            }
            if (a_post) hipLaunchKernelGGL((post_lds_kernel<T, D, M, RS, 1>), grid, block, lds_emit, st, a, L, sum, po);
            else {
                constexpr int lds2 = PostLds<T, D, M, RS, backward_row_group(M)>::TOTAL;
                if constexpr (lds2 > 64 * 1024) {
                    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&post_lds_kernel<T, D, M, RS, 2>),
                                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
                    if (attr != hipSuccess) return;
                }
                hipLaunchKernelGGL((post_lds_kernel<T, D, M, RS, 2>), grid, block, lds2, st, a, L, sum, po);
            }
            if (ev1) (void)hipEventRecord(ev1, st);
        }
    };
    using std::integral_constant;
    if (rinv_per_step) launch(integral_constant<int, 1>{}, integral_constant<bool, true>{});
    else if (m == 1) launch(integral_constant<int, 1>{}, integral_constant<bool, false>{});
    else if (m == 2) launch(integral_constant<int, 2>{}, integral_constant<bool, false>{});
    else launch(integral_constant<int, 3>{}, integral_constant<bool, false>{});
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T>
int post_chain(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
               const T* y, const T* Rinv, int rinv_per_step, T* a_post, T* mu0_post, T* b_post, T* cp0_post, T* cq_post,
               void* ws, size_t ws_bytes, int* info, long chunks, hipEvent_t ev0, hipEvent_t ev1, hipStream_t st) {
    if (B < 1 || Tn < 2 || !post_covers<T>(m, rinv_per_step)) return -101;
    // LDS-DMA moves 16-byte units: the streamed tensors must be 16-byte aligned (torch allocations are)
    if (((reinterpret_cast<size_t>(A) | reinterpret_cast<size_t>(cholQ)) & 15) != 0) return -101;
    const PostPlan pl = post_plan<T>(B, Tn, m, rinv_per_step, chunks);
    return post_launch<T>(B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, a_post, mu0_post, b_post, cp0_post,
                          cq_post, ws, ws_bytes, info, pl.P, pl.L, false, ev0, ev1, st);
}

template <typename T>
int post_emit(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
              const T* y, const T* Rinv, int rinv_per_step, T* a_post, T* mu0_post, T* b_post, T* cp0_post, T* cq_post, void* ws,
              size_t ws_bytes, int* info, long P, long L, hipStream_t st) {
    if (B < 1 || Tn < 2 || P < 1 || L < 1 || (P - 1) * L >= Tn - 1 || P * L < Tn - 1 || !post_covers<T>(m, rinv_per_step)) return -101;
    if (((reinterpret_cast<size_t>(A) | reinterpret_cast<size_t>(cholQ)) & 15) != 0) return -101;
    return post_launch<T>(B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, a_post, mu0_post, b_post, cp0_post,
                          cq_post, ws, ws_bytes, info, P, L, true, nullptr, nullptr, st);
}

template <typename T> int post_plan_of(long B, long Tn, int m, int per_step, long chunks, long* P, long* L) {
    if (B < 1 || Tn < 2 || !post_covers<T>(m, per_step)) return -101;
    const PostPlan pl = post_plan<T>(B, Tn, m, per_step, chunks);
    *P = pl.P;
    *L = pl.L;
    return 0;
}

template <typename T> const PostOps<T>* table() {
    static const PostOps<T> t = {&post_ws<T>, &post_chain<T>, &post_plan_of<T>, &post_emit<T>};
    return &t;
}

}  // namespace

#define MF_CAT2(a, b) a##b
#define MF_CAT(a, b) MF_CAT2(a, b)
const PostOps<float>* MF_CAT(post_ops_f32_d, MF_D)() { return table<float>(); }
const PostOps<double>* MF_CAT(post_ops_f64_d, MF_D)() { return table<double>(); }

}  // namespace mf
