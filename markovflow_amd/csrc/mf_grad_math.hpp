// Block arithmetic of the STREAMED backward of KalmanFilter.log_likelihood (mf_grad_lds.hpp): Fisher's identity,
// grad log p(y) = E_{x|y}[grad log p(x, y)], evaluated by a lane per (series, time chunk) that walks its transitions FORWARD in
// time with the smoothed pairwise marginals in registers - they never reach HBM:
//
//   start          (m, S) of the chunk's first block from the two sides of the separator: everything on its left (Lam, lam) -
//                  the prefix composition of the chunk summaries closed by the prior - and on its right (Psi, psi), the state
//                  the emit pass of the posterior chain restarts from:  S = (Lam + Psi)^-1,  m = S (lam + psi)
//   per transition the posterior chain gives x_{k+1} | x_k ~ N(A' x_k + b', Q'), of which only chol(Q') and b' are read:
//                  A' = Q' Q^-1 A.  Then  X = Cov(x_{k+1}, x_k) = A' S_k,  m_{k+1} = A' m_k + b',  S_{k+1} = X A'^T + Q',
//                  e = x_{k+1} - A x_k - b = (A' - A) x_k + (b' - b) + eps':  E[e] = m_{k+1} - A m_k - b,
//                  E[e x_k^T] = (A' - A) S_k + E[e] m_k^T,  Psi = (A' - A) S_k (A' - A)^T + Q' + E[e] E[e]^T  and
//                  d/dA = Q^-1 E[e x^T],  d/db = Q^-1 E[e],  d/dC = tril(C^-T (C^-1 Psi C^-T - I)),
//                  r = y - H x_k:  d/dH = R^-1 (E[r] m^T - H S),  d/dy = -R^-1 E[r],  Omega = E[r] E[r]^T + H S H^T
// (the same closed forms as kf_grad_kernel, mf_kernels.hpp, which reads the moments from HBM).  Everything is
// `__host__ __device__`: tests/host_sim runs these steps on the CPU against the oracle.
// Reference: TensorFlow reverse mode through kalman_filter.py:184-255 (banded_matrices' registered gradients); pinned by
// tests/integration/models/test_gaussian_process_regression.py:117-130 and test_variational.py:123-132 there.
#pragma once
#include "mf_post_math.hpp"

namespace mf {

// P0^-1 + H_0^T R^-1 H_0 (lower) and P0^-1 mu0 + H_0^T R^-1 y_0: what block 0 owns
template <typename T, int D, int M>
MF_HD void grad_prior_terms(const T (&C0)[D][D], const T (&mu0)[D], const T* hk, const T* yk, const T* Rsh, T (&Lam)[D][D],
                            T (&lam)[D], bool& bad) {
    T Ci[D][D], w[D];
    LogAcc<T> unused;
    unused.init();
    tri_inv_lower<T, D>(C0, Ci, unused, bad);
    trimul_lower_vec<T, D>(Ci, mu0, w);
    trimulT_self_lower<T, D>(Ci, Lam);
    trimulT_lower_vec<T, D>(Ci, w, lam);
    Obs<T, D, M>::apply(hk, yk, Rsh, M, Lam, lam);
}

// Marginal of a block from its two sides: (Lam, lam) everything on the left INCLUDING the block's own terms (destroyed),
// (Psi, psi) everything on the right.  S comes back in the lower triangle.
template <typename T, int D>
MF_HD void grad_marginal(T (&Lam)[D][D], T (&lam)[D], const T (&Psi)[D][D], const T (&psi)[D], T (&m)[D], T (&S)[D][D], bool& bad) {
    MF_UNROLL for (int i = 0; i < D; ++i) {
        lam[i] += psi[i];
        MF_UNROLL for (int j = 0; j <= i; ++j) Lam[i][j] += Psi[i][j];
    }
    T Li[D], Linv[D][D], z[D];
    LogAcc<T> unused;
    unused.init();
    chol_lower<T, D>(Lam, Li, unused, bad);
    tri_inv_lower_d<T, D>(Lam, Li, Linv);
    trimulT_self_lower<T, D>(Linv, S);                              // (L L^T)^-1 = L^-T L^-1
    trimul_lower_vec<T, D>(Linv, lam, z);
    trimulT_lower_vec<T, D>(Linv, z, m);
}

// (Lam, lam) of the block a prefix of chunks [0, c] leaves on its right: the composition `pre` (remaining block = block 0) closed
// by the prior's terms.  post_combine with the prior as a pseudo-run on the left whose separator is block 0.
template <typename T, int D>
MF_HD void grad_close_prefix(const PostSummary<T, D>& pre, const T (&Lam0)[D][D], const T (&lam0)[D], T (&Lam)[D][D], T (&lam)[D],
                             bool& bad) {
    PostSummary<T, D> pr;
    MF_UNROLL for (int i = 0; i < D; ++i) {
        pr.tv[i] = T(0);
        pr.gU[i] = lam0[i];
        MF_UNROLL for (int j = 0; j < D; ++j) {
            pr.F[i][j] = T(0);
            pr.Dv[i][j] = T(0);
            pr.GU[i][j] = j <= i ? Lam0[i][j] : T(0);
        }
    }
    post_combine<T, D>(pre, pr, bad);
    MF_UNROLL for (int i = 0; i < D; ++i) {
        lam[i] = pr.gU[i];
        MF_UNROLL for (int j = 0; j <= i; ++j) Lam[i][j] = pr.GU[i][j];
    }
}

// Gradients of the terms block 0 owns: the prior (mu0, cholP0) and the observation of time point 0.
template <typename T, int D>
MF_HD void grad_prior(const T (&C0)[D][D], const T (&mu0)[D], const T (&m)[D], const T (&S)[D][D], T wgt, T (&gmu0)[D],
                      T (&gC0)[D][D], bool& bad) {
    T Ci[D][D], dv[D], u[D];
    LogAcc<T> unused;
    unused.init();
    tri_inv_lower<T, D>(C0, Ci, unused, bad);
    MF_UNROLL for (int i = 0; i < D; ++i) dv[i] = m[i] - mu0[i];
    trimul_lower_vec<T, D>(Ci, dv, u);
    trimulT_lower_vec<T, D>(Ci, u, gmu0);
    MF_UNROLL for (int i = 0; i < D; ++i) gmu0[i] *= wgt;
    // N = C^-1 (S + dv dv^T) C^-T - I (lower), gC0 = tril(C^-T N)
    T N1[D][D], N[D][D];
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int l = 0; l <= i; ++l) {
            T acc = T(0);
            MF_UNROLL for (int k = 0; k <= i; ++k) acc += Ci[i][k] * ((k >= l ? S[k][l] : S[l][k]) + dv[k] * dv[l]);
            N1[i][l] = acc;
        }
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j <= i; ++j) {
            T acc = (i == j) ? T(-1) : T(0);
            MF_UNROLL for (int l = 0; l <= j; ++l) acc += N1[i][l] * Ci[j][l];
            N[i][j] = acc;
        }
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j < D; ++j) {
            T acc = T(0);
            if (j <= i) { MF_UNROLL for (int k = i; k < D; ++k) acc += Ci[k][i] * N[k][j]; }
            gC0[i][j] = wgt * acc;
        }
}

// Observation of one time point with smoothed moments (m, S): gH [M D], gy [M], gOm [M M], weighted.
template <typename T, int D, int M>
MF_HD void grad_obs(const T* hk, const T* yk, const T* Rsh, const T (&m)[D], const T (&S)[D][D], T wgt, T (&gH)[M * D], T (&gy)[M],
                    T (&gOm)[M * M]) {
    T r[M], Rr[M], HS[M][D];
    MF_UNROLL for (int o = 0; o < M; ++o) {
        T acc = yk[o];
        MF_UNROLL for (int i = 0; i < D; ++i) acc -= hk[o * D + i] * m[i];
        r[o] = acc;
        MF_UNROLL for (int i = 0; i < D; ++i) {
            T hs = T(0);
            MF_UNROLL for (int l = 0; l < D; ++l) hs += hk[o * D + l] * (l >= i ? S[l][i] : S[i][l]);
            HS[o][i] = hs;
        }
    }
    MF_UNROLL for (int o = 0; o < M; ++o) {
        T acc = T(0);
        MF_UNROLL for (int p = 0; p < M; ++p) acc += Rsh[o * M + p] * r[p];
        Rr[o] = acc;
    }
    MF_UNROLL for (int o = 0; o < M; ++o) {
        gy[o] = -wgt * Rr[o];
        MF_UNROLL for (int i = 0; i < D; ++i) {
            T acc = Rr[o] * m[i];
            MF_UNROLL for (int p = 0; p < M; ++p) acc -= Rsh[o * M + p] * HS[p][i];
            gH[o * D + i] = wgt * acc;
        }
        MF_UNROLL for (int p = 0; p < M; ++p) {
            T acc = r[o] * r[p];
            MF_UNROLL for (int i = 0; i < D; ++i) acc += HS[o][i] * hk[p * D + i];
            gOm[o * M + p] = wgt * acc;
        }
    }
}

// pump stand-in (host simulation): the device pump issues the next step's LDS-DMA at these sites
struct NoGradPump {
    template <int K> MF_HD void site() const {}
};
constexpr int GRAD_PUMP_SITES = 8;

// ---- one transition k -> k+1, forward in time -----------------------------------------------------------------------------
// (mk, Sk): smoothed moments of block k on entry, of block k+1 on exit (S in the lower triangle).  (hk, yk, Rsh): the observation
// of block k - its gradient is formed FIRST, from the moments on entry, so that H and y are dead for the rest of the step (the
// last block of a series is left to the caller).  C = cholQ_k (lower); A_k, G = cholQ'_k (lower), b_k and b'_k are read through
// `Aat(i, j)`, `Gat(i, j)`, `bqat(i)`, `bpat(i)` - on the device from the LDS image, A and G twice, so that none of them occupies
// registers between its uses (the step's live set is what limits the kernel: Ci, A', X, S_k and S_{k+1} at the peak).
// Outputs go to `sink` as soon as they exist: put_obs(gH, gy, gOm), put_gA<HALF>(rows), put_gb(v), put_gC<HALF>(rows).
// Pump sites (a stream's next rows may be fetched after the last read of the current ones): 0 after C, H, y; 3 after the last
// read of A, b, b'; 4 after the last read of G; 1, 2, 5, 6, 7 are free places to spread batches over.
template <typename T, int D, int M, typename AReader, typename GReader, typename BqReader, typename BpReader, typename Pump,
          typename Sink>
MF_HD void grad_step(T (&mk)[D], T (&Sk)[D][D], bool& bad, const T (&C)[D][D], const T (&hk)[M * D], const T (&yk)[M],
                     const T (&Rsh)[M * M], T wgt, const AReader& Aat, const GReader& Gat, const BqReader& bqat,
                     const BpReader& bpat, const Pump& pump, Sink& sink, bool active) {
    {
        T gH[M * D], gy[M], gOm[M * M];
        if (active) grad_obs<T, D, M>(hk, yk, Rsh, mk, Sk, wgt, gH, gy, gOm);
        sink.put_obs(gH, gy, gOm, active);
    }
    constexpr int H0 = (D + 1) / 2;
    T Ci[D][D];
    if (active) {
        LogAcc<T> unused;
        unused.init();
        tri_inv_lower<T, D>(C, Ci, unused, bad);
    }
    pump.template site<0>();
    T Ap[D][D], Sn[D][D];
    if (active) {
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Ap[i][j] = Aat(i, j);
        trimul_lower_inplace<T, D, D>(Ci, Ap);                       // C^-1 A
        trimulT_lower_inplace<T, D, D>(Ci, Ap);                      // Q^-1 A
        MF_UNROLL for (int i = 0; i < D; ++i) {                      // G^T Q^-1 A (top-down)
            const T gii = Gat(i, i);
            MF_UNROLL for (int j = 0; j < D; ++j) Ap[i][j] *= gii;
            MF_UNROLL for (int k = i + 1; k < D; ++k) {
                const T gki = Gat(k, i);
                MF_UNROLL for (int j = 0; j < D; ++j) Ap[i][j] += gki * Ap[k][j];
            }
        }
        MF_UNROLL for (int i = D - 1; i >= 0; --i) {                 // A' = G G^T Q^-1 A (bottom-up)
            const T gii = Gat(i, i);
            MF_UNROLL for (int j = 0; j < D; ++j) Ap[i][j] *= gii;
            MF_UNROLL for (int k = 0; k < i; ++k) {
                const T gik = Gat(i, k);
                MF_UNROLL for (int j = 0; j < D; ++j) Ap[i][j] += gik * Ap[k][j];
            }
        }
        // Q' = G G^T, the start of S_{k+1}
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                T acc = Gat(i, 0) * Gat(j, 0);
                MF_UNROLL for (int l = 1; l <= j; ++l) acc += Gat(i, l) * Gat(j, l);
                Sn[i][j] = acc;
            }
    }
    pump.template site<1>();
    T X[D][D], mn[D], eb[D];
    if (active) {
        // X = Cov(x_{k+1}, x_k) = A' S_k,  m_{k+1} = A' m_k + b',  S_{k+1} = X A'^T + Q'
        MF_UNROLL for (int i = 0; i < D; ++i) {
            mn[i] = bpat(i);
            MF_UNROLL for (int j = 0; j < D; ++j) X[i][j] = Ap[i][0] * (0 >= j ? Sk[0][j] : Sk[j][0]);
        }
        MF_UNROLL for (int l = 1; l < D; ++l)
            MF_UNROLL for (int i = 0; i < D; ++i)
                MF_UNROLL for (int j = 0; j < D; ++j) X[i][j] += Ap[i][l] * (l >= j ? Sk[l][j] : Sk[j][l]);
        MF_UNROLL for (int l = 0; l < D; ++l) MF_UNROLL for (int i = 0; i < D; ++i) mn[i] += Ap[i][l] * mk[l];
        MF_UNROLL for (int l = 0; l < D; ++l)
            MF_UNROLL for (int i = 0; i < D; ++i)
                MF_UNROLL for (int j = 0; j <= i; ++j) Sn[i][j] += X[i][l] * Ap[j][l];
    }
    pump.template site<2>();
    if (active) {
        // X <- (A' - A) S_k,  E[e] = m_{k+1} - A m_k - b;  then A' <- A' - A
        MF_UNROLL for (int i = 0; i < D; ++i) eb[i] = mn[i] - bqat(i);
        MF_UNROLL for (int l = 0; l < D; ++l)
            MF_UNROLL for (int i = 0; i < D; ++i) {
                const T ail = Aat(i, l);
                eb[i] -= ail * mk[l];
                MF_UNROLL for (int j = 0; j < D; ++j) X[i][j] -= ail * (l >= j ? Sk[l][j] : Sk[j][l]);
                Ap[i][l] -= ail;
            }
        // the moments of block k+1 (block k's are not needed any more)
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Sk[i][j] = Sn[i][j];
    }
    pump.template site<3>();
    T Psi[D][D];
    if (active) {
        // Psi = (A' - A) S_k (A' - A)^T + Q' + E[e] E[e]^T  (lower)
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                T acc = eb[i] * eb[j];
                MF_UNROLL for (int l = 0; l <= j; ++l) acc += Gat(i, l) * Gat(j, l);
                Psi[i][j] = acc;
            }
        MF_UNROLL for (int l = 0; l < D; ++l)
            MF_UNROLL for (int i = 0; i < D; ++i)
                MF_UNROLL for (int j = 0; j <= i; ++j) Psi[i][j] += X[i][l] * Ap[j][l];
        // E[e x_k^T] = (A' - A) S_k + E[e] m_k^T;  d/dA = Q^-1 E[e x^T]
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) X[i][j] += eb[i] * mk[j];
        MF_UNROLL for (int i = 0; i < D; ++i) mk[i] = mn[i];
        trimul_lower_inplace<T, D, D>(Ci, X);
    }
    pump.template site<4>();
    if (active) {
        trimulT_lower_inplace<T, D, D>(Ci, X);
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) X[i][j] *= wgt;
    }
    {
        T rows[H0][D];
        MF_UNROLL for (int i = 0; i < H0; ++i) MF_UNROLL for (int j = 0; j < D; ++j) rows[i][j] = X[i][j];
        sink.template put_gA<0>(rows, active);
    }
    if constexpr (D - H0 > 0) {
        T rows[D - H0][D];
        MF_UNROLL for (int i = H0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) rows[i - H0][j] = X[i][j];
        sink.template put_gA<1>(rows, active);
    }
    pump.template site<5>();
    {
        T u[D], db[D];
        if (active) {
            trimul_lower_vec<T, D>(Ci, eb, u);
            trimulT_lower_vec<T, D>(Ci, u, db);
            MF_UNROLL for (int i = 0; i < D; ++i) db[i] *= wgt;
        }
        sink.put_gb(db, active);
    }
    // d/dC = tril(C^-T N),  N = C^-1 Psi C^-T - I  (the lower triangles of C^-1 Psi and of N are all that is needed)
    T N[D][D];
    if (active) {
        T N1[D][D];
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int l = 0; l <= i; ++l) {
                T acc = Ci[i][0] * (0 >= l ? Psi[0][l] : Psi[l][0]);
                MF_UNROLL for (int k = 1; k <= i; ++k) acc += Ci[i][k] * (k >= l ? Psi[k][l] : Psi[l][k]);
                N1[i][l] = acc;
            }
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                T acc = (i == j) ? T(-1) : T(0);
                MF_UNROLL for (int l = 0; l <= j; ++l) acc += N1[i][l] * Ci[j][l];
                N[i][j] = acc;
            }
    }
    pump.template site<6>();
    {
        T rows[H0][D];
        if (active) {
            MF_UNROLL for (int i = 0; i < H0; ++i)
                MF_UNROLL for (int j = 0; j < D; ++j) {
                    T acc = T(0);
                    if (j <= i) { MF_UNROLL for (int k = i; k < D; ++k) acc += Ci[k][i] * N[k][j]; }
                    rows[i][j] = wgt * acc;
                }
        }
        sink.template put_gC<0>(rows, active);
    }
    if constexpr (D - H0 > 0) {
        T rows[D - H0][D];
        if (active) {
            MF_UNROLL for (int i = H0; i < D; ++i)
                MF_UNROLL for (int j = 0; j < D; ++j) {
                    T acc = T(0);
                    if (j <= i) { MF_UNROLL for (int k = i; k < D; ++k) acc += Ci[k][i] * N[k][j]; }
                    rows[i - H0][j] = wgt * acc;
                }
        }
        sink.template put_gC<1>(rows, active);
    }
    pump.template site<7>();
}

}  // namespace mf
