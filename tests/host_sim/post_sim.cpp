// CPU build of the block steps of the streamed posterior-chain kernels (markovflow_amd/csrc/mf_post_math.hpp): the three passes
// of mf_post_lds.hpp - reversed up-sweep per chunk, scan over the chunk summaries, emit - run lane by lane on the host, with the
// kernels' own step functions, chunk convention and scan order.  Test infrastructure (tests/test_post_host_sim.py compares it
// with the numpy oracle); built with `hipcc -x hip --offload-device-only`-free host compilation:  hipcc -O2 -shared -fPIC.
#include "../../markovflow_amd/csrc/mf_grad_math.hpp"

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
    void stage_factor_rest(const T (&Gi)[D][D], const T (&)[D], bool) {
        for (int i = H0; i < D; ++i) for (int j = 0; j < D; ++j) cq_post[k * D * D + i * D + j] = j <= i ? Gi[i][j] : T(0);
    }
    template <int HALF, int R> void stage_transition(const T (&Ap)[R][D], bool) {
        const int r0 = HALF * H0, r1 = HALF ? D : H0;
        for (int i = r0; i < r1; ++i) for (int j = 0; j < D; ++j) a_post[k * D * D + i * D + j] = Ap[i - r0][j];
    }
};

// the streamed backward of log_likelihood (mf_grad_math.hpp): where the forward-in-time pass puts its gradients
template <typename T> struct GradOut {
    const T* weights; T* gmu0; T* gC0; T* gA; T* gb; T* gC; T* gH; T* gy; T* gOm;
};
template <typename T, int D, int M> struct HostGradSink {
    static constexpr int H0 = (D + 1) / 2;
    const GradOut<T>& o; long t, k1;      // transition index in [B, T-1] tensors, time point index in [B, T] tensors
    template <int HALF, int R> void put_gA(const T (&rows)[R][D], bool) {
        for (int i = 0; i < R; ++i) for (int j = 0; j < D; ++j) o.gA[t * D * D + (HALF * H0 + i) * D + j] = rows[i][j];
    }
    template <int HALF, int R> void put_gC(const T (&rows)[R][D], bool) {
        for (int i = 0; i < R; ++i) for (int j = 0; j < D; ++j) o.gC[t * D * D + (HALF * H0 + i) * D + j] = rows[i][j];
    }
    void put_gb(const T (&v)[D], bool) { store_vec<T, D>(o.gb + t * D, v); }
    void put_obs(const T (&gH)[M * D], const T (&gy)[M], const T (&gOm)[M * M], bool) {
        for (int e = 0; e < M * D; ++e) o.gH[k1 * M * D + e] = gH[e];
        for (int e = 0; e < M; ++e) o.gy[k1 * M + e] = gy[e];
        for (int e = 0; e < M * M; ++e) o.gOm[k1 * M * M + e] = gOm[e];
    }
};

template <typename T, int D, int M>
int run(long B, long Tn, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H, const T* y,
        const T* Rinv, int per_step, long L, T* a_post, T* mu0_post, T* b_post, T* cp0_post, T* cq_post,
        const GradOut<T>* go = nullptr) {
    const long nt = Tn - 1;
    if (nt < 1 || L < 1) return -1;
    const long P = (nt + L - 1) / L;
    bool bad = false;
    std::vector<PostSummary<T, D>> sum(P), tmp(P), raw(P);
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
        raw = sum;
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
        if (!go) continue;
        // ---- the backward of log_likelihood: start moments of every chunk (prefix compositions in time order, closed by the
        // prior), then a forward pass per chunk ----
        const T wgt = go->weights ? go->weights[s] : T(1);
        T C0[D][D] = {}, m0v[D], h0[M * D], y0[M], R0[M * M], Lam0[D][D] = {}, lam0[D];
        load_lower<T, D>(cholP0 + s * D * D, C0);
        load_vec<T, D>(mu0 + s * D, m0v);
        for (int e = 0; e < M * D; ++e) h0[e] = H[(s * Tn) * M * D + e];
        for (int e = 0; e < M; ++e) y0[e] = y[(s * Tn) * M + e];
        for (int e = 0; e < M * M; ++e) R0[e] = per_step ? Rinv[(s * Tn) * M * M + e] : Rinv[e];
        grad_prior_terms<T, D, M>(C0, m0v, h0, y0, R0, Lam0, lam0, bad);
        PostSummary<T, D> pre = raw[P - 1];
        for (long c = 0; c < P; ++c) {
            const long tau0 = c * L;
            long len = nt - tau0;
            if (len > L) len = L;
            T Lam[D][D] = {}, lam[D], Psi[D][D] = {}, psi[D], mk[D], Sk[D][D] = {};
            if (c == 0) {
                for (int i = 0; i < D; ++i) { lam[i] = lam0[i]; for (int j = 0; j <= i; ++j) Lam[i][j] = Lam0[i][j]; }
                for (int i = 0; i < D; ++i) { psi[i] = sum[P - 1].tv[i]; for (int j = 0; j <= i; ++j) Psi[i][j] = sum[P - 1].Dv[i][j]; }
            } else {
                grad_close_prefix<T, D>(pre, Lam0, lam0, Lam, lam, bad);      // pre = chunks 0 .. c-1
                for (int i = 0; i < D; ++i) { psi[i] = T(0); for (int j = 0; j <= i; ++j) Psi[i][j] = T(0); }
                if (c < P) {
                    // the state the emit pass of chunk c-1 restarts from: everything on the right of block c L
                    const PostSummary<T, D>& o = sum[P - 1 - c];
                    for (int i = 0; i < D; ++i) { psi[i] = o.tv[i]; for (int j = 0; j <= i; ++j) Psi[i][j] = o.Dv[i][j]; }
                }
                post_combine<T, D>(raw[P - 1 - c], pre, bad);                 // pre = chunks 0 .. c
            }
            grad_marginal<T, D>(Lam, lam, Psi, psi, mk, Sk, bad);
            if (c == 0) {
                T gmu0[D], gC0[D][D];
                grad_prior<T, D>(C0, m0v, mk, Sk, wgt, gmu0, gC0, bad);
                store_vec<T, D>(go->gmu0 + s * D, gmu0);
                store_mat<T, D, D>(go->gC0 + s * D * D, gC0);
            }
            for (long e = 0; e < len; ++e) {
                const long t = tau0 + e;
                T C[D][D] = {}, mv[D], hk1[M * D], yk1[M], Rsh[M * M], hk[M * D], yk[M], Bm[D][D], G[D][D] = {}, bp[D];
                load_step(t, C, mv, hk1, yk1, Rsh, Bm);
                // the observation of the block the transition leaves
                for (int q = 0; q < M * D; ++q) hk[q] = H[(s * Tn + t) * M * D + q];
                for (int q = 0; q < M; ++q) yk[q] = y[(s * Tn + t) * M + q];
                for (int q = 0; q < M * M; ++q) Rsh[q] = per_step ? Rinv[(s * Tn + t) * M * M + q] : Rinv[q];
                load_lower<T, D>(cq_post + (s * nt + t) * D * D, G);
                load_vec<T, D>(b_post + (s * nt + t) * D, bp);
                HostGradSink<T, D, M> sink{*go, s * nt + t, s * Tn + t};
                auto Aat = [&](int i, int j) { return Bm[i][j]; };
                auto Gat = [&](int i, int j) { return G[i][j]; };
                auto bqat = [&](int i) { return mv[i]; };
                auto bpat = [&](int i) { return bp[i]; };
                grad_step<T, D, M>(mk, Sk, bad, C, hk, yk, Rsh, wgt, Aat, Gat, bqat, bpat, NoGradPump{}, sink, true);
            }
            if (c == P - 1) {                                                  // the last block of the series: its observation
                T hk[M * D], yk[M], Rsh[M * M], gH[M * D], gy[M], gOm[M * M];
                for (int q = 0; q < M * D; ++q) hk[q] = H[(s * Tn + nt) * M * D + q];
                for (int q = 0; q < M; ++q) yk[q] = y[(s * Tn + nt) * M + q];
                for (int q = 0; q < M * M; ++q) Rsh[q] = per_step ? Rinv[(s * Tn + nt) * M * M + q] : Rinv[q];
                grad_obs<T, D, M>(hk, yk, Rsh, mk, Sk, wgt, gH, gy, gOm);
                HostGradSink<T, D, M> sink{*go, 0, s * Tn + nt};
                sink.put_obs(gH, gy, gOm, true);
            }
        }
    }
    return bad ? 1 : 0;
}

template <typename T, int D>
int run_m(int m, long B, long Tn, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H, const T* y,
          const T* Rinv, int per_step, long L, T* a_post, T* mu0_post, T* b_post, T* cp0_post, T* cq_post,
          const GradOut<T>* go = nullptr) {
    switch (m) {
        case 1: return run<T, D, 1>(B, Tn, mu0, cholP0, A, b, cholQ, H, y, Rinv, per_step, L, a_post, mu0_post, b_post, cp0_post, cq_post, go);
        case 2: return run<T, D, 2>(B, Tn, mu0, cholP0, A, b, cholQ, H, y, Rinv, per_step, L, a_post, mu0_post, b_post, cp0_post, cq_post, go);
        case 3: return run<T, D, 3>(B, Tn, mu0, cholP0, A, b, cholQ, H, y, Rinv, per_step, L, a_post, mu0_post, b_post, cp0_post, cq_post, go);
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

// the same three passes, then the streamed backward of log_likelihood on the chain they produced (only cholQ' and b' are read)
extern "C" int mf_grad_host_sim_f64(int64_t B, int64_t Tn, int d, int m, const double* mu0, const double* cholP0, const double* A,
                                    const double* b, const double* cholQ, const double* H, const double* y, const double* Rinv,
                                    int per_step, int64_t L, const double* weights, double* gmu0, double* gC0, double* gA,
                                    double* gb, double* gC, double* gH, double* gy, double* gOm) {
    const long nt = Tn - 1;
    std::vector<double> a_post(B * nt * d * d), b_post(B * nt * d), cq_post(B * nt * d * d), mu0_post(B * d), cp0_post(B * d * d);
    const GradOut<double> go{weights, gmu0, gC0, gA, gb, gC, gH, gy, gOm};
#define MF_CASE(DD) case DD: return run_m<double, DD>(m, B, Tn, mu0, cholP0, A, b, cholQ, H, y, Rinv, per_step, L, a_post.data(), mu0_post.data(), b_post.data(), cp0_post.data(), cq_post.data(), &go);
    switch (d) {
        MF_CASE(1) MF_CASE(2) MF_CASE(3) MF_CASE(4) MF_CASE(5) MF_CASE(6)
        default: return -3;
    }
#undef MF_CASE
}
