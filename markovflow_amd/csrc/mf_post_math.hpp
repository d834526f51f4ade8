// Arithmetic of the time-partitioned, fused posterior chain (posterior_state_space_model, kalman_filter.py:109-182) for one
// lane = one (series, time-chunk): the block steps of the three passes of mf_post_lds.hpp, free of any memory plumbing.
//
//   backward recursion (block_tri_diag.py:438-545, kalman_filter.py:149-174):
//       Delta_{T-1} = D_{T-1},  Delta_k = D_k - S_k^T Delta_{k+1}^-1 S_k,  x_k = eta_k - S_k^T Delta_{k+1}^-1 x_{k+1},
//       A'_{k+1} = -Delta_{k+1}^-1 S_k,  (mu0', b'_k) = Delta_k^-1 x_k,  (cholP0', cholQ'_k) = chol(Delta_k^-1)
//   with the posterior precision assembled on the way (state_space_model.py:431-483, kalman_filter.py:86-101):
//       D_k = Q_{k-1}^-1 + H_k^T R_k^-1 H_k + A_k^T Q_k^-1 A_k,  S_k = -Q_k^-1 A_k,
//       eta_k = Q_{k-1}^-1 b_{k-1} + H_k^T R_k^-1 y_k - A_k^T Q_k^-1 b_k          (Q_{-1} := P0, b_{-1} := mu0).
//
// The state carried from transition t+1 to transition t is the part of (Delta_{t+1}, x_{t+1}) that transitions > t produce,
//       Psi_{t+1} = A_{t+1}^T Q_{t+1}^-1 A_{t+1} - S_{t+1}^T Delta_{t+2}^-1 S_{t+1},   psi likewise,
// and the step of transition t - which reads exactly what the forward log-likelihood step of transition t reads:
// cholQ_t, b_t, A_t, H_{t+1}, y_{t+1} - completes it with Q_t^-1 + H^T R^-1 H.
//
// Everything here is `__host__ __device__`: tests/host_sim builds these very functions for the CPU and runs the three passes
// lane by lane against the numpy oracle (no GPU in the build container).
#pragma once
#include "mf_kernels.hpp"
#include <type_traits>

namespace mf {

// S(lower) += Lo^T Lo
template <typename T, int D> MF_HD void trimulT_self_lower_acc(const T (&Lo)[D][D], T (&S)[D][D]) {
    MF_UNROLL for (int k = D - 1; k >= 0; --k)
        MF_UNROLL for (int i = 0; i <= k; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) S[i][j] += Lo[k][i] * Lo[k][j];
}
// out += Lo^T a
template <typename T, int D> MF_HD void trimulT_lower_vec_acc(const T (&Lo)[D][D], const T (&a)[D], T (&out)[D]) {
    MF_UNROLL for (int k = 0; k < D; ++k)
        MF_UNROLL for (int i = 0; i <= k; ++i) out[i] += Lo[k][i] * a[k];
}
// Y <- Lo^T Y in place (row i of the result only needs rows k >= i of Y: top-down)
template <typename T, int D, int N> MF_HD void trimulT_lower_inplace(const T (&Lo)[D][D], T (&Y)[D][N]) {
    MF_UNROLL for (int i = 0; i < D; ++i) {
        MF_UNROLL for (int j = 0; j < N; ++j) Y[i][j] *= Lo[i][i];
        MF_UNROLL for (int k = i + 1; k < D; ++k)
            MF_UNROLL for (int j = 0; j < N; ++j) Y[i][j] += Lo[k][i] * Y[k][j];
    }
}
// X <- -(V^T X), column by column in place
template <typename T, int D> MF_HD void neg_mulT_inplace(const T (&V)[D][D], T (&X)[D][D]) {
    MF_UNROLL for (int c = 0; c < D; ++c) {
        T col[D];
        MF_UNROLL for (int i = 0; i < D; ++i) col[i] = V[0][i] * X[0][c];
        MF_UNROLL for (int k = 1; k < D; ++k)
            MF_UNROLL for (int i = 0; i < D; ++i) col[i] += V[k][i] * X[k][c];
        MF_UNROLL for (int i = 0; i < D; ++i) X[i][c] = -col[i];
    }
}
// In-place factorisation S = G^T G with G LOWER triangular (positive diagonal), i.e. the Cholesky factorisation taken from
// the last row upwards.  Why: S^-1 = G^-1 G^-T, so the (unique) lower Cholesky factor of S^-1 is simply G^-1 - the posterior
// chain wants chol(Delta^-1), and the usual route  L = chol(Delta), L^-1, L^-T L^-1, chol(.)  pays a second Cholesky with its
// dependent rsqrt chain.  On exit the lower triangle of S holds G and Gd[j] = 1 / G[j][j].
template <typename T, int D> MF_HD void chol_lower_rev(T (&S)[D][D], T (&Gd)[D], bool& bad) {
    MF_UNROLL for (int j = D - 1; j >= 0; --j) {
        const T s = S[j][j];
        bad |= !(s > T(0));
        const T inv = t_rsqrt<T>(s);
        S[j][j] = s * inv;
        Gd[j] = inv;
        MF_UNROLL for (int i = 0; i < j; ++i) S[j][i] *= inv;
        MF_UNROLL for (int k = 0; k < j; ++k)
            MF_UNROLL for (int i = 0; i <= k; ++i) S[k][i] -= S[j][k] * S[j][i];
    }
}
// Ci = C^-1 for lower-triangular C whose reciprocal diagonal is already known
template <typename T, int D> MF_HD void tri_inv_lower_d(const T (&C)[D][D], const T (&Cd)[D], T (&Ci)[D][D]) {
    MF_UNROLL for (int i = 0; i < D; ++i) Ci[i][i] = Cd[i];
    MF_UNROLL for (int i = 1; i < D; ++i) {
        T acc[D];
        MF_UNROLL for (int j = 0; j < i; ++j) acc[j] = T(0);
        MF_UNROLL for (int k = 0; k < i; ++k)
            MF_UNROLL for (int j = 0; j <= k; ++j) acc[j] += C[i][k] * Ci[k][j];
        MF_UNROLL for (int j = 0; j < i; ++j) Ci[i][j] = -acc[j] * Ci[i][i];
    }
}

// no-op stand-in for the LDS-DMA pump of the device kernels (host simulation, and kernels that load directly)
#ifndef MF_NOPUMP_DEFINED
#define MF_NOPUMP_DEFINED
struct NoPump {
    template <int K> MF_HD void small() const {}
    template <int K> MF_HD void big() const {}
};
#endif

// ---- pass 1 ("up"): one transition of the REVERSED partitioned elimination ------------------------------------------------
// Chunk c owns transitions [c L, min((c+1) L, T-1)) and walks them from the last to the first.  The block its last
// transition leads to (block (c+1) L) is its separator - owned, as a remaining block, by chunk c+1 - and the block its
// first transition leaves (block c L) is what remains of it.  E holds: (Phi, t) = (Psi, psi) of the current block, X the
// coupling [current block x separator], (GU, gU) what the chunk adds to the separator: for the separator this includes
// its own Q^-1 + H^T R^-1 H (FIRST step), so that the pivot a reduced elimination needs is Psi_sep + GU while the
// state the emit pass restarts from is Psi_sep alone.  The last chunk has no separator: block T-1 is eliminated like an
// interior block with an empty spike.
// Bm: A_t on entry (destroyed).  FIRST: the wave's first step (lanes with a separator take the separator form).
template <typename T, int D, int M, bool FIRST, typename Pump>
MF_HD void post_up_step(Elim<T, D, true>& E, const T (&C)[D][D], const T (&mvec)[D], const T (&hk)[M * D],
                        const T (&yk)[M], const T (&Rsh)[M * M], T (&Bm)[D][D], const Pump& pump, bool active,
                        bool sep_lane) {
    const bool sep = FIRST && sep_lane && active, ord = active && !sep;
    T Ci[D][D], w[D];
    if (active) {
        LogAcc<T> unused;
        unused.init();
        tri_inv_lower<T, D>(C, Ci, unused, E.bad);
    }
    pump.template small<0>();
    if (active) trimul_lower_vec<T, D>(Ci, mvec, w);
    pump.template small<1>();
    pump.template big<0>();
    if (FIRST && sep) {
        trimulT_self_lower<T, D>(Ci, E.GU);
        trimulT_lower_vec<T, D>(Ci, w, E.gU);
        Obs<T, D, M>::apply(hk, yk, Rsh, M, E.GU, E.gU);
    }
    if (ord) {
        trimulT_self_lower_acc<T, D>(Ci, E.Phi);
        trimulT_lower_vec_acc<T, D>(Ci, w, E.t);
        Obs<T, D, M>::apply(hk, yk, Rsh, M, E.Phi, E.t);       // (Phi, t) = (Delta, x) of block t+1 incl. the spike's fill-in
        E.eliminate_main();
    }
    pump.template big<1>();
    if (ord) E.eliminate_spike();
    pump.template big<2>();
    T Pn[D][D], pn[D];
    if (active) {
        trimul_lower_inplace<T, D, D>(Ci, Bm);                  // B = C^-1 A
        gemv_t<T, D, D>(Bm, w, pn);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            pn[i] = -pn[i];                                     // -A^T Q^-1 b
            MF_UNROLL for (int j = 0; j <= i; ++j) Pn[i][j] = T(0);
        }
        syrk_tn_lower<T, D, D>(Bm, Pn, T(1));                   // A^T Q^-1 A
        neg_trimulT_lower_inplace<T, D, D>(Ci, Bm);             // S_t = -Q^-1 A   (rows: block t+1, columns: block t)
    }
    pump.template big<3>();
    if (FIRST && sep) {
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) E.X[i][j] = Bm[j][i];
    }
    if (ord) {
        trsm_left_lower<T, D, D>(E.Phi, E.Li, Bm);              // V = L^-1 S
        syrk_tn_lower<T, D, D>(Bm, Pn, T(-1));
        T vz[D];
        gemv_t<T, D, D>(Bm, E.t, vz);
        MF_UNROLL for (int i = 0; i < D; ++i) pn[i] -= vz[i];
        neg_mulT_inplace<T, D>(Bm, E.X);                        // coupling of block t with the separator: -V^T (L^-1 X)
    }
    if (active) {
        MF_UNROLL for (int i = 0; i < D; ++i) {
            E.t[i] = pn[i];
            MF_UNROLL for (int j = 0; j <= i; ++j) E.Phi[i][j] = Pn[i][j];
        }
    }
}

// ---- pass 2 ("scan"): composition of two consecutive chunk summaries ------------------------------------------------------
// A summary (Dv, tv | GU, gU | F) of a run of chunks says: the run's remaining block has (Psi, psi) = (Dv, tv) once the
// run's separator is eliminated with pivot  Psi_sep + GU, and F couples the two.  `a` is the run on the RIGHT in time (its
// remaining block is b's separator).  The result replaces b: eliminate a's remaining block.
template <typename T, int D> struct PostSummary {
    T Dv[D][D], tv[D], GU[D][D], gU[D], F[D][D];
};
template <typename T, int D> MF_HD void post_combine(const PostSummary<T, D>& a, PostSummary<T, D>& b, bool& bad) {
    Elim<T, D, true> E;
    E.init();
    MF_UNROLL for (int i = 0; i < D; ++i) {
        E.t[i] = a.tv[i] + b.gU[i];
        E.gU[i] = a.gU[i];
        MF_UNROLL for (int j = 0; j < D; ++j) E.X[i][j] = a.F[i][j];
        MF_UNROLL for (int j = 0; j <= i; ++j) {
            E.Phi[i][j] = a.Dv[i][j] + b.GU[i][j];
            E.GU[i][j] = a.GU[i][j];
        }
    }
    E.eliminate();
    trsm_right_lower_t<T, D, D>(E.Phi, E.Li, b.F);              // W = F_b L^-T
    E.advance(b.F, b.Dv, b.tv);
    bad |= E.bad;
    MF_UNROLL for (int i = 0; i < D; ++i) {
        b.tv[i] = E.t[i];
        b.gU[i] = E.gU[i];
        MF_UNROLL for (int j = 0; j < D; ++j) b.F[i][j] = E.X[i][j];
        MF_UNROLL for (int j = 0; j <= i; ++j) {
            b.Dv[i][j] = E.Phi[i][j];
            b.GU[i][j] = E.GU[i][j];
        }
    }
}

// compile-time loop: f(std::integral_constant<int, I>) for I = I0 ... I1-1 (a tick site needs its index as a constant)
template <int I0, int I1, typename F> MF_HD void static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, I1>(f);
    }
}

// ---- pass 3 ("emit"): one transition of the textbook backward recursion, restarted from a chunk boundary ----------------
// (Phi, t) = (Psi, psi) of block t+1 on entry, of block t on exit.  Bm: A_t on entry (destroyed).  The outputs of the posterior
// chain at index t are handed to `sink` in pieces, as soon as they exist:
//   sink.stage_factor(Gi, mean)      Gi = chol(Delta_{t+1}^-1) (-> cholQ'_t; the sink takes rows 0 .. D/2-1 now),
//                                    mean = Delta_{t+1}^-1 x_{t+1} (-> b'_t)
//   sink.stage_factor_rest(Gi, mean) rows D/2 .. D-1 of the same factor (a sink that packs factor and mean takes its second part)
//   sink.stage_transition<HALF>(Ap)  rows [HALF D/2, ...) of A'_{t+1} = -Delta_{t+1}^-1 S_t (Ap holds these rows only)
// Between the arithmetic the step calls sink.tick<SITE>() at EMIT_SITES places spaced ~25-35 multiply-adds apart: a device
// sink issues ONE store instruction per tick (mf_post_lds.hpp: a SIMD's store path takes a 1-KB store per ~350 cycles under
// load and only the NEXT store waits for it, so stores spread through the arithmetic cost nothing, stores issued back to
// back cost ~13 k cycles per step).  Site map: [0, 9) while (Delta, x) is completed and factored - the previous step's last
// piece drains there -, [9, 34) the products up to A' (the factor's pieces drain), [34, 44) the rest.
// Sinks are called by ALL lanes (a device sink moves other lanes' rows); `active` is theirs to use.
// TRANS = false: the transitions A' are neither formed nor handed over (the streamed backward of log_likelihood rebuilds them from
// chol(Q'), mf_grad_math.hpp) - two triangular products and a third of the stores less.
constexpr int EMIT_SITES = 44;
template <typename T, int D, int M, bool TRANS = true, typename Pump, typename Sink>
MF_HD void post_emit_step(T (&Phi)[D][D], T (&t)[D], bool& bad, const T (&C)[D][D], const T (&mvec)[D],
                          const T (&hk)[M * D], const T (&yk)[M], const T (&Rsh)[M * M], T (&Bm)[D][D], const Pump& pump,
                          Sink& sink, bool active) {
    constexpr int H0 = (D + 1) / 2;                             // rows in the first half of a matrix output
    T Ci[D][D], w[D], z[D];
    if (active) {
        LogAcc<T> unused;
        unused.init();
        tri_inv_lower<T, D>(C, Ci, unused, bad);
    }
    sink.template tick<0>(active);
    pump.template small<0>();
    if (active) {
        trimul_lower_vec<T, D>(Ci, mvec, w);
        trimulT_self_lower_acc<T, D>(Ci, Phi);
    }
    sink.template tick<1>(active);
    if (active) {
        trimulT_lower_vec_acc<T, D>(Ci, w, t);
        Obs<T, D, M>::apply(hk, yk, Rsh, M, Phi, t);            // (Phi, t) = (Delta_{t+1}, x_{t+1})
    }
    sink.template tick<2>(active);
    pump.template small<1>();
    pump.template big<0>();
    T Gi[D][D], mean[D];
    {
        // Delta = G^T G from the last row upwards (chol_lower_rev), two rows per tick site
        T Gd[D];
        static_for<0, 3>([&](auto ic) {
            constexpr int K = decltype(ic)::value;
            if (active) {
                MF_UNROLL for (int j = D - 1 - (K * D) / 3; j > D - 1 - ((K + 1) * D) / 3; --j) {
                    const T s = Phi[j][j];
                    bad |= !(s > T(0));
                    const T inv = t_rsqrt<T>(s);
                    Phi[j][j] = s * inv;
                    Gd[j] = inv;
                    MF_UNROLL for (int i = 0; i < j; ++i) Phi[j][i] *= inv;
                    MF_UNROLL for (int k = 0; k < j; ++k)
                        MF_UNROLL for (int i = 0; i <= k; ++i) Phi[k][i] -= Phi[j][k] * Phi[j][i];
                }
            }
            sink.template tick<3 + K>(active);
        });
        if (active) tri_inv_lower_d<T, D>(Phi, Gd, Gi);         // chol(Delta^-1) = G^-1
        sink.template tick<6>(active);
        if (active) {
            trimulT_lower_vec<T, D>(Gi, t, z);                  // z = G^-T x
            trimul_lower_vec<T, D>(Gi, z, mean);                // Delta^-1 x
        }
        sink.template tick<7>(active);
        sink.template tick<8>(active);
    }
    sink.stage_factor(Gi, mean, active);
    pump.template big<1>();
    T Pn[D][D], pn[D];
    // B = C^-1 A in place, bottom-up, a row per site
    static_for<0, D>([&](auto ic) {
        constexpr int i = D - 1 - decltype(ic)::value;
        if (active) {
            MF_UNROLL for (int j = 0; j < D; ++j) Bm[i][j] *= Ci[i][i];
            MF_UNROLL for (int k = 0; k < i; ++k)
                MF_UNROLL for (int j = 0; j < D; ++j) Bm[i][j] += Ci[i][k] * Bm[k][j];
        }
        sink.template tick<9 + (D - 1 - i)>(active);
    });
    if (active) {
        gemv_t<T, D, D>(Bm, w, pn);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            pn[i] = -pn[i];
            MF_UNROLL for (int j = 0; j <= i; ++j) Pn[i][j] = T(0);
        }
    }
    sink.template tick<9 + D>(active);
    pump.template big<2>();
    // A^T Q^-1 A = B^T B, a row of B per site
    static_for<0, D>([&](auto ic) {
        constexpr int k = decltype(ic)::value;
        if (active) {
            MF_UNROLL for (int i = 0; i < D; ++i)
                MF_UNROLL for (int j = 0; j <= i; ++j) Pn[i][j] += Bm[k][i] * Bm[k][j];
        }
        sink.template tick<10 + D + k>(active);
    });
    sink.stage_factor_rest(Gi, mean, active);
    // -S_t = Q^-1 A = C^-T B in place, top-down
    static_for<0, D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (active) {
            MF_UNROLL for (int j = 0; j < D; ++j) Bm[i][j] *= Ci[i][i];
            MF_UNROLL for (int k = i + 1; k < D; ++k)
                MF_UNROLL for (int j = 0; j < D; ++j) Bm[i][j] += Ci[k][i] * Bm[k][j];
        }
        sink.template tick<10 + 2 * D + i>(active);
    });
    pump.template big<3>();
    // Vn = G^-T (-S) in place, top-down
    static_for<0, D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (active) {
            MF_UNROLL for (int j = 0; j < D; ++j) Bm[i][j] *= Gi[i][i];
            MF_UNROLL for (int k = i + 1; k < D; ++k)
                MF_UNROLL for (int j = 0; j < D; ++j) Bm[i][j] += Gi[k][i] * Bm[k][j];
        }
        sink.template tick<10 + 3 * D + i>(active);
    });
    static_assert(10 + 4 * D <= 34 && D <= 6, "site map of post_emit_step: state dimensions up to 6");
    // A'_{t+1} = -Delta^-1 S = G^-1 Vn, the first rows
    {
        T Ap[H0][D];
        if (TRANS && active) {
            MF_UNROLL for (int i = 0; i < H0; ++i) {
                MF_UNROLL for (int j = 0; j < D; ++j) Ap[i][j] = Gi[i][0] * Bm[0][j];
                MF_UNROLL for (int k = 1; k <= i; ++k)
                    MF_UNROLL for (int j = 0; j < D; ++j) Ap[i][j] += Gi[i][k] * Bm[k][j];
            }
        }
        if constexpr (TRANS) sink.template stage_transition<0>(Ap, active);
    }
    {
        T Ap[D - H0 > 0 ? D - H0 : 1][D];
        static_for<H0, D>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (TRANS && active) {
                MF_UNROLL for (int j = 0; j < D; ++j) Ap[i - H0][j] = Gi[i][0] * Bm[0][j];
                MF_UNROLL for (int k = 1; k <= i; ++k)
                    MF_UNROLL for (int j = 0; j < D; ++j) Ap[i - H0][j] += Gi[i][k] * Bm[k][j];
            }
            sink.template tick<34 + (i - H0)>(active);
        });
        // Psi_t = A^T Q^-1 A - S^T Delta^-1 S, a row of Vn per site
        static_for<0, D>([&](auto ic) {
            constexpr int k = decltype(ic)::value;
            if (active) {
                MF_UNROLL for (int i = 0; i < D; ++i)
                    MF_UNROLL for (int j = 0; j <= i; ++j) Pn[i][j] -= Bm[k][i] * Bm[k][j];
            }
            sink.template tick<34 + (D - H0) + k>(active);
        });
        if (active) {
            T vz[D];
            gemv_t<T, D, D>(Bm, z, vz);
            MF_UNROLL for (int i = 0; i < D; ++i) {
                t[i] = pn[i] + vz[i];                           // psi_t = -A^T Q^-1 b - S^T Delta^-1 x
                MF_UNROLL for (int j = 0; j <= i; ++j) Phi[i][j] = Pn[i][j];
            }
        }
        sink.template tick<34 + (D - H0) + D>(active);
        static_assert(34 + (D - H0) + D < EMIT_SITES, "site map of post_emit_step");
        if constexpr (TRANS && D - H0 > 0) sink.template stage_transition<1>(Ap, active);
    }
}

// Block 0 (the prior: cholP0, mu0, H_0, y_0) closes the chain: (mu0', cholP0') from (Psi_0, psi_0).
template <typename T, int D, int M>
MF_HD void post_emit_prior(T (&Phi)[D][D], T (&t)[D], bool& bad, const T (&C)[D][D], const T (&mvec)[D],
                           const T (&hk)[M * D], const T (&yk)[M], const T (&Rsh)[M * M], T (&mean)[D], T (&Gi)[D][D]) {
    T Ci[D][D], w[D], z[D], Gd[D];
    LogAcc<T> unused;
    unused.init();
    tri_inv_lower<T, D>(C, Ci, unused, bad);
    trimul_lower_vec<T, D>(Ci, mvec, w);
    trimulT_self_lower_acc<T, D>(Ci, Phi);
    trimulT_lower_vec_acc<T, D>(Ci, w, t);
    Obs<T, D, M>::apply(hk, yk, Rsh, M, Phi, t);
    chol_lower_rev<T, D>(Phi, Gd, bad);
    tri_inv_lower_d<T, D>(Phi, Gd, Gi);
    trimulT_lower_vec<T, D>(Gi, t, z);
    trimul_lower_vec<T, D>(Gi, z, mean);
}

}  // namespace mf
