// Instantiates the streamed backward of KalmanFilter.log_likelihood (mf_grad_lds.hpp) for ONE state dimension (compile with
// -DMF_D=<d>) and both scalar types, and exports its launch table (mf_launch.hpp: GradOps).  The posterior chain it starts from
// is the one of mf_post_inst.hip (PostOps of the same state dimension), run on the partition chosen here.
#ifndef MF_D
#error "compile with -DMF_D=<state dimension>"
#endif
#include "mf_gpr_grad.hpp"
#include "mf_launch.hpp"

#include <type_traits>

namespace mf {

#define MF_CAT2(a, b) a##b
#define MF_CAT(a, b) MF_CAT2(a, b)

namespace {

constexpr int D = MF_D;
inline long cdiv(long a, long b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

template <typename T> const PostOps<T>* post_ops();
template <> const PostOps<float>* post_ops<float>() { return MF_CAT(post_ops_f32_d, MF_D)(); }
template <> const PostOps<double>* post_ops<double>() { return MF_CAT(post_ops_f64_d, MF_D)(); }

template <typename T, int M, bool RS> constexpr bool grad_supported() { return GradLds<T, D, M, RS>::SUPPORTED; }
template <typename T> bool grad_covers(int m, int per_step) {
    if (per_step) return m == 1 && grad_supported<T, 1, true>();
    switch (m) {
        case 1: return grad_supported<T, 1, false>();
        case 2: return grad_supported<T, 2, false>();
        case 3: return grad_supported<T, 3, false>();
        default: return false;
    }
}
template <typename T> int grad_lds_bytes(int m, int per_step) {
    if (per_step) return GradLds<T, D, 1, true>::TOTAL;
    return m == 1 ? GradLds<T, D, 1, false>::TOTAL : (m == 2 ? GradLds<T, D, 2, false>::TOTAL : GradLds<T, D, 3, false>::TOTAL);
}

// chunks per series: one wavefront on every SIMD the LDS image of pass 5 leaves room for; all five passes share the partition
// (the posterior chain is asked for exactly this many chunks and reports the partition it made of them)
template <typename T> bool grad_plan(long B, long Tn, int m, int per_step, long chunks, long& P, long& L) {
    int w = (160 * 1024) / grad_lds_bytes<T>(m, per_step);
    w = w > 4 ? 4 : (w < 1 ? 1 : w);
    const long nt = Tn - 1;
    long want = chunks > 0 ? chunks : cdiv(256L * 64 * w, B);
    if (chunks <= 0) {
        const long maxP = nt / 4 > 0 ? nt / 4 : 1;
        if (want > maxP) want = maxP;
    }
    if (want > nt) want = nt;
    if (want < 2) want = 2;            // pass 4 needs the summaries of passes 1-2, which a single chunk skips
    return post_ops<T>()->plan(B, Tn, m, per_step, want, &P, &L) == 0 && P >= 2;
}

// the partition when the forward evaluation's summaries are given: groups of k of its chunks
inline bool grad_plan_from(long B, long Tn, long chunks, int w, long fwd_P, long fwd_L, long& P, long& L, long& k) {
    const long nt = Tn - 1;
    if (fwd_P < 2 || fwd_L < 1 || (fwd_P - 1) * fwd_L >= nt || fwd_P * fwd_L < nt) return false;
    long want = chunks > 0 ? chunks : cdiv(256L * 64 * w, B);
    if (want < 2) want = 2;
    k = cdiv(fwd_P, want);
    if (k > fwd_P / 2) k = fwd_P / 2;
    if (k < 1) k = 1;
    L = k * fwd_L;
    P = cdiv(nt, L);
    return P >= 2;
}

template <typename T> struct GradWs {
    size_t post, chainA, chainb, start_m, start_S, total;
    GradWs(long B, long Tn, long P) {
        const size_t nt = size_t(Tn - 1);
        post = align_up(PostWs<T, D>::bytes(B, P) + 256);
        chainA = align_up(size_t(B) * nt * PostLds<T, D, 1, false>::REC);      // the chain as packed records
        chainb = 0;
        start_m = align_up(size_t(B) * P * D * sizeof(T));
        start_S = align_up(size_t(B) * P * D * D * sizeof(T));
        total = post + chainA + chainb + align_up(size_t(B) * D * sizeof(T)) + align_up(size_t(B) * D * D * sizeof(T)) + start_m + start_S;
    }
};

// (sized for the finest partition either route makes: at most the chunks of grad_plan, or of the forward evaluation's groups)
template <typename T> size_t grad_ws(long B, long Tn, int m, int per_step, long chunks) {
    if (B < 1 || Tn < 2 || !grad_covers<T>(m, per_step)) return 0;
    long P, L;
    if (!grad_plan<T>(B, Tn, m, per_step, chunks, P, L)) return 0;
    if (post_ops<T>()->ws(B, Tn, m, per_step, P) == 0) return 0;
    return GradWs<T>(B, Tn, 2 * P + 2).total + 256;
}

template <typename T>
int grad_run(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H, const T* y,
             const T* Rinv, int rinv_per_step, const T* weights, T* g_mu0, T* g_cholP0, T* g_A, T* g_b, T* g_cholQ, T* g_H, T* g_y,
             T* g_Om, void* ws, size_t ws_bytes, int* info, long chunks, const void* fwd_ws, long fwd_P, long fwd_L,
             hipEvent_t ev0, hipEvent_t ev1, hipStream_t st) {
    if (B < 1 || Tn < 2 || !grad_covers<T>(m, rinv_per_step)) return -101;
    if (((reinterpret_cast<size_t>(A) | reinterpret_cast<size_t>(cholQ) | reinterpret_cast<size_t>(g_A) |
          reinterpret_cast<size_t>(g_cholQ)) & 15) != 0) return -101;
    long P, L, k = 0;
    if (!grad_plan<T>(B, Tn, m, rinv_per_step, chunks, P, L)) return -101;
    const long Pmax = 2 * P + 2;                     // what the workspace query promised
    bool from_fwd = false;
    if (fwd_ws != nullptr) {
        int w = (160 * 1024) / grad_lds_bytes<T>(m, rinv_per_step);
        w = w > 4 ? 4 : (w < 1 ? 1 : w);
        long P2, L2;
        if (grad_plan_from(B, Tn, chunks, w, fwd_P, fwd_L, P2, L2, k) && P2 <= Pmax) { from_fwd = true; P = P2; L = L2; }
    }
    const GradWs<T> lay(B, Tn, P);
    if (ws == nullptr || ws_bytes < lay.total) return -21;
    char* p = static_cast<char*>(ws);
    void* post_ws = p; p += lay.post;
    T* rec_post = reinterpret_cast<T*>(p); p += lay.chainA;
    T* mu0_post = reinterpret_cast<T*>(p); p += align_up(size_t(B) * D * sizeof(T));
    T* cp0_post = reinterpret_cast<T*>(p); p += align_up(size_t(B) * D * D * sizeof(T));
    T* start_m = reinterpret_cast<T*>(p); p += lay.start_m;
    T* start_S = reinterpret_cast<T*>(p);
    const PostWs<T, D> w = PostWs<T, D>::carve(post_ws, B, P);
    KfArgs<T> a{B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, info, 0, weights};
    const GradIo<T> io{rec_post, w.bPsi, w.bpsi, start_m, start_S, from_fwd ? mu0_post : nullptr,
                       from_fwd ? cp0_post : nullptr, g_mu0, g_cholP0, g_A, g_b, g_cholQ, g_H, g_y, g_Om};
    const dim3 grid((unsigned)cdiv(B * P, 64)), block(64);
    constexpr int scan_lds = PostScanLds<T, D>::BYTES;
    if (ev0) (void)hipEventRecord(ev0, st);
    if (from_fwd) {
        // the summaries of the forward evaluation (level 0 of mf_kf_loglik: the first fwd_P blocks per series of its workspace)
        // stand in for passes 1, 2 and 4: boundary states and start moments at every k-th of its chunk boundaries
        char* fp = const_cast<char*>(static_cast<const char*>(fwd_ws));
        RedSys<T> k0;
        {
            T* base = reinterpret_cast<T*>(fp);
            const long nb = B * fwd_P;
            k0.Dv = base; k0.GU = k0.Dv + nb * D * D; k0.F = k0.GU + nb * D * D; k0.tv = k0.F + nb * D * D;
            k0.gU = k0.tv + nb * D; k0.sc = k0.gU + nb * D;
            k0.n = fwd_P; k0.f_stride = fwd_P; k0.f_off = 0;
        }
        int G = 64;                                    // lanes per series: a power of two that holds the forward's chunks
        if (fwd_P <= 32) { G = 1; while (G < fwd_P) G <<= 1; }
        const dim3 sgrid((unsigned)cdiv(B, 64 / G));
        hipLaunchKernelGGL((k0_scan_kernel<T, D, false>), sgrid, block, scan_lds, st, k0, B, G, k, P, io, info);
        hipLaunchKernelGGL((k0_scan_kernel<T, D, true>), sgrid, block, scan_lds, st, k0, B, G, k, P, io, info);
        const int rc = post_ops<T>()->emit(B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, nullptr, mu0_post, nullptr,
                                           cp0_post, rec_post, post_ws, lay.post, info, P, L, st);
        if (rc != 0) return rc;
    } else {
        // passes 1-3: the posterior chain (without its transitions) on P chunks
        const int rc = post_ops<T>()->chain(B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, nullptr, mu0_post, nullptr,
                                            cp0_post, rec_post, post_ws, lay.post, info, P, nullptr, nullptr, st);
        if (rc != 0) return rc;
    }
    bool launched = false;                  // (a call that launched nothing must not hand back uninitialised gradients: ADVICE r04)
    auto launch = [&](auto mtag, auto rtag) {
        constexpr int M = decltype(mtag)::value;
        constexpr bool RS = decltype(rtag)::value;
        if constexpr (GradLds<T, D, M, RS>::SUPPORTED) {
            constexpr int lds = GradLds<T, D, M, RS>::TOTAL;
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&grad_lds_kernel<T, D, M, RS>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (attr != hipSuccess) return;
            if (!from_fwd) hipLaunchKernelGGL((grad_start_kernel<T, D, M>), dim3((unsigned)B), block, scan_lds, st, a, w.sum, io);
            hipLaunchKernelGGL((grad_lds_kernel<T, D, M, RS>), grid, block, lds, st, a, L, io);
            launched = true;
        }
    };
    using std::integral_constant;
    if (rinv_per_step) launch(integral_constant<int, 1>{}, integral_constant<bool, true>{});
    else if (m == 1) launch(integral_constant<int, 1>{}, integral_constant<bool, false>{});
    else if (m == 2) launch(integral_constant<int, 2>{}, integral_constant<bool, false>{});
    else launch(integral_constant<int, 3>{}, integral_constant<bool, false>{});
    if (ev1) (void)hipEventRecord(ev1, st);
    if (!launched) return -1000;
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// ---- posterior_state_space_model from the filter's summaries ------------------------------------------------------------------------
// groups of k forward chunks, k chosen so that the emit pass runs on about the partition it would choose itself
template <typename T> bool post_from_fwd_plan(long B, long Tn, int m, int per_step, long fwd_P, long fwd_L, long& P, long& L, long& k) {
    const long nt = Tn - 1;
    if (fwd_P < 2 || fwd_L < 1 || (fwd_P - 1) * fwd_L >= nt || fwd_P * fwd_L < nt) return false;
    long Pw = 0, Lw = 0;
    if (post_ops<T>()->plan(B, Tn, m, per_step, 0, &Pw, &Lw) != 0 || Pw < 1) return false;
    k = cdiv(fwd_P, Pw);
    if (k < 1) k = 1;
    L = k * fwd_L;
    P = cdiv(nt, L);
    return P >= 1;
}
template <typename T> size_t post_from_fwd_ws(long B, long Tn, int m, int per_step, long fwd_P) {
    if (B < 1 || Tn < 2 || fwd_P < 2) return 0;
    if (post_ops<T>()->ws(B, Tn, m, per_step, 0) == 0) return 0;
    return align_up(PostWs<T, D>::bytes(B, fwd_P) + 256);           // (no group has more chunks than the forward evaluation)
}
template <typename T>
int post_from_fwd(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H, const T* y,
                  const T* Rinv, int rinv_per_step, T* a_post, T* mu0_post, T* b_post, T* cp0_post, T* cq_post, void* ws,
                  size_t ws_bytes, int* info, const void* fwd_ws, long fwd_P, long fwd_L, hipEvent_t ev0, hipEvent_t ev1,
                  hipStream_t st) {
    if (B < 1 || Tn < 2 || fwd_ws == nullptr || a_post == nullptr) return -101;
    long P, L, k;
    if (!post_from_fwd_plan<T>(B, Tn, m, rinv_per_step, fwd_P, fwd_L, P, L, k)) return -101;
    if (ws == nullptr || ws_bytes < PostWs<T, D>::bytes(B, P)) return -21;
    const PostWs<T, D> w = PostWs<T, D>::carve(ws, B, P);
    if (ev0) (void)hipEventRecord(ev0, st);
    if (P > 1) {
        RedSys<T> k0;
        T* base = reinterpret_cast<T*>(const_cast<char*>(static_cast<const char*>(fwd_ws)));
        const long nb = B * fwd_P;
        k0.Dv = base; k0.GU = k0.Dv + nb * D * D; k0.F = k0.GU + nb * D * D; k0.tv = k0.F + nb * D * D;
        k0.gU = k0.tv + nb * D; k0.sc = k0.gU + nb * D;
        k0.n = fwd_P; k0.f_stride = fwd_P; k0.f_off = 0;
        const GradIo<T> io{nullptr, w.bPsi, w.bpsi, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr};
        int G = 64;
        if (fwd_P <= 32) { G = 1; while (G < fwd_P) G <<= 1; }
        constexpr int scan_lds = PostScanLds<T, D>::BYTES;
        hipLaunchKernelGGL((k0_scan_kernel<T, D, true>), dim3((unsigned)cdiv(B, 64 / G)), dim3(64), scan_lds, st, k0, B, G, k, P, io,
                           info);
    }
    const int rc = post_ops<T>()->emit(B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, a_post, mu0_post, b_post, cp0_post,
                                       cq_post, ws, ws_bytes, info, P, L, st);
    if (ev1) (void)hipEventRecord(ev1, st);
    return rc;
}

// ---- GaussianProcessRegression.log_likelihood: the backward with the kernel -> state-space-model step fused (mf_gpr_grad.hpp) ------
// Needs the summaries of the fused forward (mf_gpr_matern_loglik on fwd_P chunks of fwd_L transitions): the backward runs on
// the same partition.  Workspace: boundary states, start moments, the chain's packed records, block 0's marginal.
template <typename T> struct GprGradWs {
    size_t post, rec, start_m, start_S, m0, c0, total;
    GprGradWs(long B, long Tn, long P) {
        const size_t nt = size_t(Tn - 1);
        post = align_up(PostWs<T, D>::bytes(B, P) + 256);
        rec = align_up(size_t(B) * nt * PostLds<T, D, 1, false>::REC);
        start_m = align_up(size_t(B) * P * D * sizeof(T));
        start_S = align_up(size_t(B) * P * D * D * sizeof(T));
        m0 = align_up(size_t(B) * D * sizeof(T));
        c0 = align_up(size_t(B) * D * D * sizeof(T));
        total = post + rec + start_m + start_S + m0 + c0;
    }
};
template <typename T> size_t gpr_grad_ws(long B, long Tn, long fwd_P) {
    if (B < 1 || Tn < 2 || fwd_P < 2) return 0;
    return GprGradWs<T>(B, Tn, fwd_P).total + 256;
}

template <typename T, int O0, int O1>
int gpr_grad_launch(const GprArgs<T>& a, const GprBwdIo<T>& eio, const GradIo<T>& gio, const T* weights, hipStream_t st) {
    using Gen = GprGen<T, O0, O1>;
    using GB = GprBwdLds<T, Gen::D, Gen::K0 * Gen::K0 + Gen::K1 * Gen::K1>;
    static_assert(Gen::D == D, "signature of another state dimension");
    constexpr int lds = GB::TOTAL;
    static_assert(lds <= 64 * 1024, "GPR backward: LDS beyond the default dynamic-LDS limit");
    const dim3 grid((unsigned)cdiv(a.B * a.P, 64)), block(64);
    hipLaunchKernelGGL((gpr_emit_kernel<T, O0, O1>), grid, block, lds, st, a, eio);
    hipLaunchKernelGGL((gpr_grad_kernel<T, O0, O1>), grid, block, lds, st, a, gio, weights);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T>
int gpr_grad_run(long B, long Tn, int ncomp, const int* orders, const T* lam, const T* var, int per_series, const T* t, const T* y,
                 const T* rinv, T jitter, const T* weights, T* g_packed, T* g_cholP0, T* g_Om, void* ws, size_t ws_bytes,
                 int* info, const void* fwd_ws, long fwd_P, long fwd_L, hipStream_t st) {
    if (B < 1 || Tn < 2 || ncomp < 1 || ncomp > 2) return -101;
    const long nt = Tn - 1;
    if (fwd_ws == nullptr || fwd_P < 2 || fwd_L < 1 || (fwd_P - 1) * fwd_L >= nt || fwd_P * fwd_L < nt) return -101;
    if (reinterpret_cast<size_t>(g_packed) & 15) return -101;
    const long P = fwd_P, L = fwd_L;
    const GprGradWs<T> lay(B, Tn, P);
    if (ws == nullptr || ws_bytes < lay.total) return -21;
    char* p = static_cast<char*>(ws);
    void* post_ws = p; p += lay.post;
    void* rec = p; p += lay.rec;
    T* start_m = reinterpret_cast<T*>(p); p += lay.start_m;
    T* start_S = reinterpret_cast<T*>(p); p += lay.start_S;
    T* mu0_post = reinterpret_cast<T*>(p); p += lay.m0;
    T* cp0_post = reinterpret_cast<T*>(p);
    const PostWs<T, D> w = PostWs<T, D>::carve(post_ws, B, P);
    const GradIo<T> gio{rec, w.bPsi, w.bpsi, start_m, start_S, mu0_post, cp0_post, nullptr, g_cholP0, g_packed, nullptr, nullptr,
                        nullptr, nullptr, g_Om};
    const GprBwdIo<T> eio{rec, w.bPsi, w.bpsi, mu0_post, cp0_post, nullptr, nullptr, nullptr};
    RedSys<T> k0;
    {
        T* base = reinterpret_cast<T*>(const_cast<char*>(static_cast<const char*>(fwd_ws)));
        const long nb = B * P;
        k0.Dv = base; k0.GU = k0.Dv + nb * D * D; k0.F = k0.GU + nb * D * D; k0.tv = k0.F + nb * D * D;
        k0.gU = k0.tv + nb * D; k0.sc = k0.gU + nb * D;
        k0.n = P; k0.f_stride = P; k0.f_off = 0;
    }
    int G = 64;
    if (P <= 32) { G = 1; while (G < P) G <<= 1; }
    const dim3 sgrid((unsigned)cdiv(B, 64 / G)), block(64);
    constexpr int scan_lds = PostScanLds<T, D>::BYTES;
    hipLaunchKernelGGL((k0_scan_kernel<T, D, false>), sgrid, block, scan_lds, st, k0, B, G, 1L, P, gio, info);
    hipLaunchKernelGGL((k0_scan_kernel<T, D, true>), sgrid, block, scan_lds, st, k0, B, G, 1L, P, gio, info);
    const GprArgs<T> a{B, Tn, lam, var, per_series ? (long)ncomp : 0L, t, y, rinv, jitter, P, L, info};
    const int o0 = orders[0], o1 = ncomp > 1 ? orders[1] : 0;
    int rc = -101;
    if constexpr (D == 1) { if (o0 == 1 && o1 == 0) rc = gpr_grad_launch<T, 1, 0>(a, eio, gio, weights, st); }
    if constexpr (D == 2) { if (o0 == 3 && o1 == 0) rc = gpr_grad_launch<T, 3, 0>(a, eio, gio, weights, st); }
    if constexpr (D == 3) { if (o0 == 5 && o1 == 0) rc = gpr_grad_launch<T, 5, 0>(a, eio, gio, weights, st); }
    if constexpr (D == 4) { if (o0 == 3 && o1 == 3) rc = gpr_grad_launch<T, 3, 3>(a, eio, gio, weights, st); }
    if constexpr (D == 5) {
        if (o0 == 5 && o1 == 3) rc = gpr_grad_launch<T, 5, 3>(a, eio, gio, weights, st);
        else if (o0 == 3 && o1 == 5) rc = gpr_grad_launch<T, 3, 5>(a, eio, gio, weights, st);
    }
    if constexpr (D == 6) { if (o0 == 5 && o1 == 5) rc = gpr_grad_launch<T, 5, 5>(a, eio, gio, weights, st); }
    return rc;
}

// ---- GaussianProcessRegression: posterior_state_space_model with the kernel -> state-space-model step fused --------------------------
template <typename T, int O0, int O1>
int gpr_post_launch(const GprArgs<T>& a, const GprBwdIo<T>& eio, hipStream_t st) {
    static_assert(GprGen<T, O0, O1>::D == D, "signature of another state dimension");
    constexpr int lds = GprPostLds<T, D>::TOTAL;
    static_assert(lds <= 64 * 1024, "GPR posterior: LDS beyond the default dynamic-LDS limit");
    hipLaunchKernelGGL((gpr_emit_kernel<T, O0, O1, true>), dim3((unsigned)cdiv(a.B * a.P, 64)), dim3(64), lds, st, a, eio);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
template <typename T> size_t gpr_post_ws(long B, long Tn, long fwd_P) {
    if (B < 1 || Tn < 2 || fwd_P < 2) return 0;
    return align_up(PostWs<T, D>::bytes(B, fwd_P) + 256);
}
template <typename T>
int gpr_post_run(long B, long Tn, int ncomp, const int* orders, const T* lam, const T* var, int per_series, const T* t, const T* y,
                 const T* rinv, T jitter, T* a_post, T* mu0_post, T* b_post, T* cp0_post, T* cq_post, void* ws, size_t ws_bytes,
                 int* info, const void* fwd_ws, long fwd_P, long fwd_L, hipStream_t st) {
    if (B < 1 || Tn < 2 || ncomp < 1 || ncomp > 2) return -101;
    const long nt = Tn - 1;
    if (fwd_ws == nullptr || fwd_P < 2 || fwd_L < 1 || (fwd_P - 1) * fwd_L >= nt || fwd_P * fwd_L < nt) return -101;
    if ((reinterpret_cast<size_t>(a_post) | reinterpret_cast<size_t>(cq_post)) & 15) return -101;
    const long P = fwd_P, L = fwd_L;
    if (ws == nullptr || ws_bytes < PostWs<T, D>::bytes(B, P)) return -21;
    const PostWs<T, D> w = PostWs<T, D>::carve(ws, B, P);
    RedSys<T> k0;
    {
        T* base = reinterpret_cast<T*>(const_cast<char*>(static_cast<const char*>(fwd_ws)));
        const long nb = B * P;
        k0.Dv = base; k0.GU = k0.Dv + nb * D * D; k0.F = k0.GU + nb * D * D; k0.tv = k0.F + nb * D * D;
        k0.gU = k0.tv + nb * D; k0.sc = k0.gU + nb * D;
        k0.n = P; k0.f_stride = P; k0.f_off = 0;
    }
    const GradIo<T> io{nullptr, w.bPsi, w.bpsi, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr};
    int G = 64;
    if (P <= 32) { G = 1; while (G < P) G <<= 1; }
    constexpr int scan_lds = PostScanLds<T, D>::BYTES;
    hipLaunchKernelGGL((k0_scan_kernel<T, D, true>), dim3((unsigned)cdiv(B, 64 / G)), dim3(64), scan_lds, st, k0, B, G, 1L, P, io, info);
    const GprArgs<T> a{B, Tn, lam, var, per_series ? (long)ncomp : 0L, t, y, rinv, jitter, P, L, info};
    const GprBwdIo<T> eio{nullptr, w.bPsi, w.bpsi, mu0_post, cp0_post, a_post, b_post, cq_post};
    const int o0 = orders[0], o1 = ncomp > 1 ? orders[1] : 0;
    int rc = -101;
    if constexpr (D == 1) { if (o0 == 1 && o1 == 0) rc = gpr_post_launch<T, 1, 0>(a, eio, st); }
    if constexpr (D == 2) { if (o0 == 3 && o1 == 0) rc = gpr_post_launch<T, 3, 0>(a, eio, st); }
    if constexpr (D == 3) { if (o0 == 5 && o1 == 0) rc = gpr_post_launch<T, 5, 0>(a, eio, st); }
    if constexpr (D == 4) { if (o0 == 3 && o1 == 3) rc = gpr_post_launch<T, 3, 3>(a, eio, st); }
    if constexpr (D == 5) {
        if (o0 == 5 && o1 == 3) rc = gpr_post_launch<T, 5, 3>(a, eio, st);
        else if (o0 == 3 && o1 == 5) rc = gpr_post_launch<T, 3, 5>(a, eio, st);
    }
    if constexpr (D == 6) { if (o0 == 5 && o1 == 5) rc = gpr_post_launch<T, 5, 5>(a, eio, st); }
    return rc;
}

template <typename T> const GradOps<T>* table() {
    static const GradOps<T> t = {&grad_ws<T>, &grad_run<T>, &post_from_fwd_ws<T>, &post_from_fwd<T>, &gpr_grad_ws<T>, &gpr_grad_run<T>,
                                 &gpr_post_ws<T>, &gpr_post_run<T>};
    return &t;
}

}  // namespace

const GradOps<float>* MF_CAT(grad_ops_f32_d, MF_D)() { return table<float>(); }
const GradOps<double>* MF_CAT(grad_ops_f64_d, MF_D)() { return table<double>(); }

}  // namespace mf
