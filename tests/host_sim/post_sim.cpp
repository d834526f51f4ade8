// CPU build of the block steps of the streamed posterior-chain kernels (markovflow_amd/csrc/mf_post_math.hpp): the three passes
// of mf_post_lds.hpp - reversed up-sweep per chunk, scan over the chunk summaries, emit - run lane by lane on the host, with the
// kernels' own step functions, chunk convention and scan order.  Test infrastructure (tests/test_post_host_sim.py compares it
// with the numpy oracle); built with `hipcc -x hip --offload-device-only`-free host compilation:  hipcc -O2 -shared -fPIC.
#include "../../markovflow_amd/csrc/mf_post_math.hpp"

#include <cstdint>
#include <vector>

namespace {
using namespace mf;

// where the emit step hands over its outputs: plain stores at index k of the posterior chain (tick sites: nothing to do)
template <typename T, int D> struct HostSink {
    static constexpr int H0 = (D + 1) / 2;
    T* a_post; T* b_post; T* cq_post; long k;
    template <int SITE> void tick(bool) { static_assert(SITE >= 0 && SITE < EMIT_SITES, "tick site out of range"); }
    void stage_factor(const T (&Gi)[D][D], const T (&mean)[D], bool) {
        for (int i = 0; i < H0; ++i) for (int j = 0; j < D; ++j) cq_post[k * D * D + i * D + j] = j <= i ? Gi[i][j] : T(0);
        store_vec<T, D>(b_post + k * D, mean);
    }
    void stage_factor_rest(const T (&Gi)[D][D], bool) {
        for (int i = H0; i < D; ++i) for (int j = 0; j < D; ++j) cq_post[k * D * D + i * D + j] = j <= i ? Gi[i][j] : T(0);
    }
    template <int HALF, int R> void stage_transition(const T (&Ap)[R][D], bool) {
        const int r0 = HALF * H0, r1 = HALF ? D : H0;
        for (int i = r0; i < r1; ++i) for (int j = 0; j < D; ++j) a_post[k * D * D + i * D + j] = Ap[i - r0][j];
    }
};

template <typename T, int D, int M>
int run(long B, long Tn, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H, const T* y,
        const T* Rinv, int per_step, long L, T* a_post, T* mu0_post, T* b_post, T* cp0_post, T* cq_post) {
    const long nt = Tn - 1;
    if (nt < 1 || L < 1) return -1;
    const long P = (nt + L - 1) / L;
    bool bad = false;
    std::vector<PostSummary<T, D>> sum(P), tmp(P);
    for (long s = 0; s < B; ++s) {
        auto load_step = [&](long t, T (&C)[D][D], T (&mv)[D], T (&hk)[M * D], T (&yk)[M], T (&Rsh)[M * M], T (&Bm)[D][D]) {
            load_lower<T, D>(cholQ + (s * nt + t) * D * D, C);
            load_vec<T, D>(b + (s * nt + t) * D, mv);
            load_mat<T, D, D>(A + (s * nt + t) * D * D, Bm);
            for (int e = 0; e < M * D; ++e) hk[e] = H[(s * Tn + t + 1) * M * D + e];
            for (int e = 0; e < M; ++e) yk[e] = y[(s * Tn + t + 1) * M + e];
            for (int e = 0; e < M * M; ++e) Rsh[e] = per_step ? Rinv[(s * Tn + t + 1) * M * M + e] : Rinv[e];
        };
        // ---- pass 1: summaries of chunks 1 .. P-1 (mirrored index j = P-1-c) ----
        for (long c = 0; c < P; ++c) {
            const long tau0 = c * L;
            long len = nt - tau0;
            if (len > L) len = L;
            Elim<T, D, true> E;
            E.init();
            for (long e = len - 1; e >= 0; --e) {
                T C[D][D] = {}, mv[D], hk[M * D], yk[M], Rsh[M * M], Bm[D][D];
                load_step(tau0 + e, C, mv, hk, yk, Rsh, Bm);
                if (e == len - 1) post_up_step<T, D, M, true>(E, C, mv, hk, yk, Rsh, Bm, NoPump{}, true, c < P - 1);
                else post_up_step<T, D, M, false>(E, C, mv, hk, yk, Rsh, Bm, NoPump{}, true, c < P - 1);
            }
            bad |= E.bad;
            PostSummary<T, D>& o = sum[P - 1 - c];
            for (int i = 0; i < D; ++i) {
                o.tv[i] = E.t[i]; o.gU[i] = E.gU[i];
                for (int j = 0; j < D; ++j) {
                    o.F[i][j] = E.X[i][j];
                    o.Dv[i][j] = j <= i ? E.Phi[i][j] : T(0);
                    o.GU[i][j] = j <= i ? E.GU[i][j] : T(0);
                }
            }
        }
        // ---- pass 2: inclusive Kogge-Stone scan over the mirrored summaries ----
        for (long off = 1; off < P; off *= 2) {
            tmp = sum;
            for (long j = off; j < P; ++j) post_combine<T, D>(tmp[j - off], sum[j], bad);
        }
        // ---- pass 3: emit ----
        for (long c = 0; c < P; ++c) {
            const long tau0 = c * L;
            long len = nt - tau0;
            if (len > L) len = L;
            T Phi[D][D] = {}, t[D] = {};
            if (c < P - 1) {
                const PostSummary<T, D>& o = sum[P - 2 - c];
                for (int i = 0; i < D; ++i) {
                    t[i] = o.tv[i];
                    for (int j = 0; j <= i; ++j) Phi[i][j] = o.Dv[i][j];
                }
            }
            for (long e = len - 1; e >= 0; --e) {
                T C[D][D] = {}, mv[D], hk[M * D], yk[M], Rsh[M * M], Bm[D][D];
                load_step(tau0 + e, C, mv, hk, yk, Rsh, Bm);
                HostSink<T, D> sink{a_post, b_post, cq_post, s * nt + tau0 + e};
                post_emit_step<T, D, M>(Phi, t, bad, C, mv, hk, yk, Rsh, Bm, NoPump{}, sink, true);
            }
            if (c == 0) {
                T C[D][D] = {}, mv[D], hk[M * D], yk[M], Rsh[M * M], mean[D], Gi[D][D] = {};
                load_lower<T, D>(cholP0 + s * D * D, C);
                load_vec<T, D>(mu0 + s * D, mv);
                for (int e = 0; e < M * D; ++e) hk[e] = H[(s * Tn) * M * D + e];
                for (int e = 0; e < M; ++e) yk[e] = y[(s * Tn) * M + e];
                for (int e = 0; e < M * M; ++e) Rsh[e] = per_step ? Rinv[(s * Tn) * M * M + e] : Rinv[e];
                post_emit_prior<T, D, M>(Phi, t, bad, C, mv, hk, yk, Rsh, mean, Gi);
                store_vec<T, D>(mu0_post + s * D, mean);
                store_lower<T, D>(cp0_post + s * D * D, Gi);
            }
        }
    }
    return bad ? 1 : 0;
}

template <typename T, int D>
int run_m(int m, long B, long Tn, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H, const T* y,
          const T* Rinv, int per_step, long L, T* a_post, T* mu0_post, T* b_post, T* cp0_post, T* cq_post) {
    switch (m) {
        case 1: return run<T, D, 1>(B, Tn, mu0, cholP0, A, b, cholQ, H, y, Rinv, per_step, L, a_post, mu0_post, b_post, cp0_post, cq_post);
        case 2: return run<T, D, 2>(B, Tn, mu0, cholP0, A, b, cholQ, H, y, Rinv, per_step, L, a_post, mu0_post, b_post, cp0_post, cq_post);
        case 3: return run<T, D, 3>(B, Tn, mu0, cholP0, A, b, cholQ, H, y, Rinv, per_step, L, a_post, mu0_post, b_post, cp0_post, cq_post);
        default: return -2;
    }
}
}  // namespace

extern "C" int mf_post_host_sim_f64(int64_t B, int64_t Tn, int d, int m, const double* mu0, const double* cholP0, const double* A,
                                    const double* b, const double* cholQ, const double* H, const double* y, const double* Rinv,
                                    int per_step, int64_t L, double* a_post, double* mu0_post, double* b_post, double* cp0_post,
                                    double* cq_post) {
#define MF_CASE(DD) case DD: return run_m<double, DD>(m, B, Tn, mu0, cholP0, A, b, cholQ, H, y, Rinv, per_step, L, a_post, mu0_post, b_post, cp0_post, cq_post);
    switch (d) {
        MF_CASE(1) MF_CASE(2) MF_CASE(3) MF_CASE(4) MF_CASE(5) MF_CASE(6)
        default: return -3;
    }
#undef MF_CASE
}
