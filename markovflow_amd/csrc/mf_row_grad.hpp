// Row form (mf_row.hpp: one 16-lane DPP row per sub-problem, matrix rows across the lanes) of the LOCAL gradient kernel of
// KalmanFilter.log_likelihood, kf_grad_kernel (mf_kernels.hpp): Fisher's identity on the smoothed pairwise marginals, one row
// per (series, time point).  Same inputs, outputs and closed forms:
//   e_k = x_{k+1} - A_k x_k - b_k:  E[e] = m_{k+1} - A m_k - b,  Psi = E[e]E[e]^T + S_{k+1} - A X^T - X A^T + A S_k A^T  (X = Cov(x_{k+1}, x_k))
//   d/dA = Q^-1 (E[e] m_k^T + X - A S_k),  d/db = Q^-1 E[e],  d/dC = tril(C^-T (C^-1 Psi C^-T - I)) = tril((Q^-1 Psi - I) C^-T),
//   d/dmu0, d/dcholP0 likewise from (m_0 - mu0, S_0);  r = y - H x:  d/dH = R^-1 (E[r] m^T - H S),  d/dy = -R^-1 E[r],
//   Omega = E[r]E[r]^T + H S H^T.
// At d = 9 the lane-per-point kernel spills 3.7 KB per lane and takes 1.84 ms at BASELINE config 4's shape (it is also the
// q2-side of kl_divergence's backward); the products below are d^2 broadcast-FMAs for four points at a time.
// Reference: TensorFlow reverse mode through kalman_filter.py:184-255 (tests/integration/models/test_variational.py:123-132).
#pragma once
#include "mf_row_scan.hpp"

namespace mf {
namespace row {

// out[j] -= sum_k own a[k] * bcast_j(b[k])
template <typename T, int D> MF_DEV void row_mul_t_sub(const T (&a)[D], const T (&b)[D], T (&out)[D]) {
    using P = Dpp<T>;
    sfor<D>([&](auto k) {
        constexpr int kk = decltype(k)::value;
        sfor<D>([&](auto j) { P::template fnmac<decltype(j)::value>(out[decltype(j)::value], b[kk], a[kk]); });
    });
}

// G = (Q^-1 Psi - I) C^-T, rows; only j <= r is meaningful (the lower triangle)
template <typename T, int D> MF_DEV void row_chol_grad(const T (&Qi)[D], T (&Psi)[D], const T (&CiT)[D], int r, T (&G)[D]) {
    using P = Dpp<T>;
    T M1[D];
    sfor<D>([&](auto j) { M1[decltype(j)::value] = r == decltype(j)::value ? T(-1) : T(0); });
    fence(Psi);
    row_mul<T, D, D>(Qi, Psi, M1);
    sfor<D>([&](auto j) { G[decltype(j)::value] = T(0); });
    sfor<D>([&](auto l) {                                   // CiT of lane l, element j = Ci[j][l]: zero for j < l
        constexpr int ll = decltype(l)::value;
        sfor2<ll, D>([&](auto j) { P::template fmac<ll>(G[decltype(j)::value], CiT[decltype(j)::value], M1[ll]); });
    });
}

template <typename T, int D, int M>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, false)) row_kf_grad_kernel(
    KfArgs<T> a, const T* __restrict__ pm, const T* __restrict__ pS, const T* __restrict__ pX, T* __restrict__ gmu0,
    T* __restrict__ gC0, T* __restrict__ gA, T* __restrict__ gb, T* __restrict__ gC, T* __restrict__ gH, T* __restrict__ gy,
    T* __restrict__ gOm) {
    using P = Dpp<T>;
    const int lane = threadIdx.x, r = lane & 15, rc = r < D ? r : D - 1;
    const long total = a.B * a.Tn;
    const long id_raw = (long)blockIdx.x * 4 + (lane >> 4);
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;
    long s, k;
    if (total < (1L << 31)) { const unsigned su = (unsigned)id / (unsigned)a.Tn; s = su; k = (long)((unsigned)id - su * (unsigned)a.Tn); }
    else { s = id / a.Tn; k = id % a.Tn; }
    const bool st = valid && r < D;
    const T wgt = a.weights ? a.weights[s] : T(1);
    bool bad = false;
    T mk = pm[id * D + rc];                                           // own element of m_k
    T Sk[D];
    load_row<T, D>(pS + id * D * D, rc, Sk);
    // ---- observation terms of time point k (skipped without an emission model: the score of a bare chain) ----
    if (a.H != nullptr) {
        const T* __restrict__ Rv = a.Rinv + (a.rinv_per_step ? id * M * M : 0);
        const T one = T(1);
        T h[M], HS[M], rr[M], Rr[M];
        sfor<M>([&](auto o) { h[decltype(o)::value] = a.H[(id * M + decltype(o)::value) * D + rc] * (r < D ? T(1) : T(0)); });
        fence(h);
        sfor<M>([&](auto o) {
            constexpr int oo = decltype(o)::value;
            T prod = h[oo] * mk;                                      // lanes >= D: zero
            T acc = a.y[id * M + oo];
            fence1(prod);
            sfor<D>([&](auto l) { P::template fnmac<decltype(l)::value>(acc, prod, one); });     // E[r_o] = y_o - h_o . m, replicated
            rr[oo] = acc;
            T hs = T(0);
            sfor<D>([&](auto l) { P::template fmac<decltype(l)::value>(hs, h[oo], Sk[decltype(l)::value]); });   // (H S)[o][r]
            HS[oo] = hs;
        });
        sfor<M>([&](auto o) {
            constexpr int oo = decltype(o)::value;
            T acc = T(0);
            sfor<M>([&](auto p) { acc = __builtin_fma(Rv[oo * M + decltype(p)::value], rr[decltype(p)::value], acc); });
            Rr[oo] = acc;
        });
        sfor<M>([&](auto o) {
            constexpr int oo = decltype(o)::value;
            T acc = Rr[oo] * mk;                                      // dH = R^-1 (r m^T - H S)
            sfor<M>([&](auto p) { acc = __builtin_fma(-Rv[oo * M + decltype(p)::value], HS[decltype(p)::value], acc); });
            if (st) gH[(id * M + oo) * D + r] = wgt * acc;
            if (valid && r == 0) gy[id * M + oo] = -wgt * Rr[oo];
            sfor<M>([&](auto p) {                                     // Omega = r r^T + H S H^T
                constexpr int pp = decltype(p)::value;
                T prod = HS[oo] * h[pp];
                T om = rr[oo] * rr[pp];
                fence1(prod);
                sfor<D>([&](auto l) { P::template fmac<decltype(l)::value>(om, prod, one); });
                if (valid && r == 0) gOm[(id * M + oo) * M + pp] = wgt * om;
            });
        });
    }
    // ---- prior of the first state (rows with k = 0 only; whole rows take or skip the branch) ----
    if (k == 0) {
        asm volatile("s_nop 4");
        T C[D], CiT[D], Qi[D], Psi[D], G[D];
        load_row_lower<T, D>(a.cholP0 + s * D * D, rc, C);
        const T cd = a.cholP0[s * D * D + rc * (D + 1)];
        bad |= r < D && !(cd != T(0));
        row_qinv<T, D>(C, t_rcp<T>(cd), r, CiT, Qi);
        T dv = mk - a.mu0[s * D + rc];
        fence1(dv);
        T g = T(0);
        sfor<D>([&](auto l) { P::template fmac<decltype(l)::value>(g, dv, Qi[decltype(l)::value]); });          // P0^-1 (m0 - mu0)
        if (st) gmu0[s * D + r] = wgt * g;
        sfor<D>([&](auto j) { Psi[decltype(j)::value] = Sk[decltype(j)::value]; });
        sfor<D>([&](auto j) { P::template fmac<decltype(j)::value>(Psi[decltype(j)::value], dv, dv); });          // + dv dv^T
        row_chol_grad<T, D>(Qi, Psi, CiT, r, G);
        if (st) sfor<D>([&](auto j) { gC0[s * D * D + r * D + decltype(j)::value] = decltype(j)::value <= r ? wgt * G[decltype(j)::value] : T(0); });
    }
    // ---- transition k -> k + 1 (absent at the last point of a series) ----
    if (k + 1 < a.Tn) {
        asm volatile("s_nop 4");
        const long tid = s * (a.Tn - 1) + k;
        T Sn[D], X[D], Am[D], C[D], CiT[D], Qi[D];
        load_row<T, D>(pS + (id + 1) * D * D, rc, Sn);
        load_row<T, D>(pX + tid * D * D, rc, X);
        load_row<T, D>(a.A + tid * D * D, rc, Am);
        load_row_lower<T, D>(a.cholQ + tid * D * D, rc, C);
        const T cd = a.cholQ[tid * D * D + rc * (D + 1)];
        bad |= r < D && !(cd != T(0));
        T eb = pm[(id + 1) * D + rc] - a.b[tid * D + rc];
        fence1(mk);
        sfor<D>([&](auto l) { P::template fnmac<decltype(l)::value>(eb, mk, Am[decltype(l)::value]); });        // E[e] = m' - b - A m
        row_qinv<T, D>(C, t_rcp<T>(cd), r, CiT, Qi);
        // E[e x^T] = X - A S + E[e] m^T;  A S kept for Psi
        T AS[D], EX[D];
        sfor<D>([&](auto j) { AS[decltype(j)::value] = T(0); });
        fence(Sk);
        row_mul<T, D, D>(Am, Sk, AS);
        sfor<D>([&](auto j) { EX[decltype(j)::value] = X[decltype(j)::value] - AS[decltype(j)::value]; });
        sfor<D>([&](auto j) { P::template fmac<decltype(j)::value>(EX[decltype(j)::value], mk, eb); });
        // dA = Q^-1 E[e x^T], db = Q^-1 E[e]
        T dA[D], db = T(0);
        sfor<D>([&](auto j) { dA[decltype(j)::value] = T(0); });
        fence(EX);
        row_mul<T, D, D>(Qi, EX, dA);
        fence1(eb);
        sfor<D>([&](auto l) { P::template fmac<decltype(l)::value>(db, eb, Qi[decltype(l)::value]); });
        if (st) {
            sfor<D>([&](auto j) { gA[tid * D * D + r * D + decltype(j)::value] = wgt * dA[decltype(j)::value]; });
            gb[tid * D + r] = wgt * db;
        }
        // Psi = E[e]E[e]^T + S' - A X^T - X A^T + A S A^T
        T Psi[D], G[D];
        sfor<D>([&](auto j) { Psi[decltype(j)::value] = Sn[decltype(j)::value]; });
        sfor<D>([&](auto j) { P::template fmac<decltype(j)::value>(Psi[decltype(j)::value], eb, eb); });
        fence(Am);
        fence(X);
        row_mul_t<T, D>(AS, Am, Psi);
        row_mul_t_sub<T, D>(Am, X, Psi);
        row_mul_t_sub<T, D>(X, Am, Psi);
        row_chol_grad<T, D>(Qi, Psi, CiT, r, G);
        if (st) sfor<D>([&](auto j) { gC[tid * D * D + r * D + decltype(j)::value] = decltype(j)::value <= r ? wgt * G[decltype(j)::value] : T(0); });
    }
    if (valid && bad && a.info) raise_info(a.info);
}

// ---- parameter gradients of kl_divergence (with respect to its FIRST chain) and of `marginals`, local in time given the adjoint
// moments (M_k, lam_k) of the backward scans: row form of ssm_adjoint_local_kernel (mf_kl_grad.hpp), one row per (series, step) ----
//   KL:         db = w (Q2^-1 eps + lam'),  dA = w ((Q2^-1 eps + lam') m^T + (Q2^-1 dA + M' A1) S),  dC = w (tril((Q2^-1 + M') C1) - diag(1 / C1)),
//               step 0 also  dmu0 = w (P0_2^-1 d0 + lam_0),  dC0 = w (tril((P0_2^-1 + M_0) C0_1) - diag(1 / C0_1));
//   marginals:  db = lam',  dA = lam' m^T + M' A S,  dC = tril(M' C),  dmu0 = lam_0,  dC0 = tril(M_0 C0).
// Everything in row layout: Q2^-1 as rows (row_qinv), products as broadcast-FMAs, vectors distributed over the lanes.
template <typename T, int D, bool KL>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, false)) row_adjoint_local_kernel(AdjointLocalArgs<T, D> a, AdjointWs<T, D> ws) {
    using P = Dpp<T>;
    const int lane = threadIdx.x, r = lane & 15, rc = r < D ? r : D - 1;
    const long total = a.B * a.Tn;
    const long id_raw = (long)blockIdx.x * 4 + (lane >> 4);
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;
    long s, k;
    if (total < (1L << 31)) { const unsigned su = (unsigned)id / (unsigned)a.Tn; s = su; k = (long)((unsigned)id - su * (unsigned)a.Tn); }
    else { s = id / a.Tn; k = id % a.Tn; }
    const bool st = valid && r < D;
    const T w = (KL && a.weights) ? a.weights[s] : T(1);
    bool bad = false;
    // out(rows, lower part) = w (tril((Qi + Ms) C) - ent diag(1 / C)),  Qi, Ms rows; C: own row of the factor (zero above the diagonal)
    auto chol_grad = [&](const T (&Qi)[D], const T (&Ms)[D], T (&C)[D], T cd, T* dst) {
        T Pm[D], out[D];
        sfor<D>([&](auto j) { Pm[decltype(j)::value] = KL ? Qi[decltype(j)::value] + Ms[decltype(j)::value] : Ms[decltype(j)::value]; out[decltype(j)::value] = T(0); });
        fence(C);
        row_mul<T, D, D>(Pm, C, out);
        if (st) {
            const T ent = KL ? t_rcp<T>(cd) : T(0);
            sfor<D>([&](auto j) {
                constexpr int jj = decltype(j)::value;
                dst[r * D + jj] = jj < r ? w * out[jj] : (jj == r ? w * (out[jj] - ent) : T(0));
            });
        }
    };
    if (k == 0) {
        // ---- prior of the first state ----
        asm volatile("s_nop 4");
        T M0[D], C1[D], Qi[D];
        load_row<T, D>(ws.M + id * D * D, rc, M0);
        load_row_lower<T, D>(a.C0_1 + s * D * D, rc, C1);
        const T c1d = a.C0_1[s * D * D + rc * (D + 1)];
        T g = ws.lam[id * D + rc];
        if constexpr (KL) {
            T C2[D], CiT[D];
            load_row_lower<T, D>(a.C0_2 + s * D * D, rc, C2);
            const T c2d = a.C0_2[s * D * D + rc * (D + 1)];
            bad |= r < D && (!(c2d != T(0)) || !(c1d != T(0)));
            row_qinv<T, D>(C2, t_rcp<T>(c2d), r, CiT, Qi);
            T d0 = a.mu0_1[s * D + rc] - a.mu0_2[s * D + rc];
            fence1(d0);
            sfor<D>([&](auto l) { P::template fmac<decltype(l)::value>(g, d0, Qi[decltype(l)::value]); });
        } else {
            sfor<D>([&](auto j) { Qi[decltype(j)::value] = T(0); });
        }
        if (st) a.gmu0[s * D + r] = w * g;
        chol_grad(Qi, M0, C1, c1d, a.gC0 + s * D * D);
    }
    if (k + 1 < a.Tn) {
        // ---- transition k -> k + 1 ----
        asm volatile("s_nop 4");
        const long tid = s * (a.Tn - 1) + k;
        T Mn[D], A1[D], Sk[D], C1[D], Qi[D], G[D], out[D];
        load_row<T, D>(ws.M + (id + 1) * D * D, rc, Mn);             // M_{k+1} (symmetric, stored full)
        load_row<T, D>(a.A_1 + tid * D * D, rc, A1);
        load_row<T, D>(a.pS + id * D * D, rc, Sk);
        load_row_lower<T, D>(a.C_1 + tid * D * D, rc, C1);
        const T c1d = a.C_1[tid * D * D + rc * (D + 1)];
        T gl = ws.lam[(id + 1) * D + rc];                             // lam_{k+1}, own element
        T mk = a.pm[id * D + rc];
        sfor<D>([&](auto j) { G[decltype(j)::value] = T(0); out[decltype(j)::value] = T(0); });
        fence(A1);
        if constexpr (KL) {
            T C2[D], CiT[D], dA[D];
            load_row_lower<T, D>(a.C_2 + tid * D * D, rc, C2);
            const T c2d = a.C_2[tid * D * D + rc * (D + 1)];
            bad |= r < D && (!(c2d != T(0)) || !(c1d != T(0)));
            row_qinv<T, D>(C2, t_rcp<T>(c2d), r, CiT, Qi);
            {
                T a2[D];
                load_row<T, D>(a.A_2 + tid * D * D, rc, a2);
                sfor<D>([&](auto j) { dA[decltype(j)::value] = A1[decltype(j)::value] - a2[decltype(j)::value]; });
            }
            T eps = a.b_1[tid * D + rc] - a.b_2[tid * D + rc];
            fence1(mk);
            sfor<D>([&](auto l) { P::template fmac<decltype(l)::value>(eps, mk, dA[decltype(l)::value]); });      // eps = db + dA m
            fence1(eps);
            sfor<D>([&](auto l) { P::template fmac<decltype(l)::value>(gl, eps, Qi[decltype(l)::value]); });      // + Q2^-1 eps
            fence(dA);
            row_mul<T, D, D>(Qi, dA, G);                              // Q2^-1 dA
        } else {
            sfor<D>([&](auto j) { Qi[decltype(j)::value] = T(0); });
            fence1(mk);
        }
        row_mul<T, D, D>(Mn, A1, G);                                  // + M' A1
        fence(Sk);
        row_mul<T, D, D>(G, Sk, out);                                 // (...) S
        sfor<D>([&](auto j) { P::template fmac<decltype(j)::value>(out[decltype(j)::value], mk, gl); });            // + gl m^T
        if (st) {
            a.gb[tid * D + r] = w * gl;
            sfor<D>([&](auto j) { a.gA[tid * D * D + r * D + decltype(j)::value] = w * out[decltype(j)::value]; });
        }
        chol_grad(Qi, Mn, C1, c1d, a.gC + tid * D * D);
    }
    if (valid && bad && a.info) raise_info(a.info);
}

}   // namespace row
}   // namespace mf
