// Local step of the log-likelihood's gradient for 16 <= d <= 32 on the register tiles of mf_wave.hpp: Fisher's identity
// grad log p(y) = E_{x|y}[grad log p(x, y)] evaluated from the smoothed moments m_k, S_k, X_{k-1} = Cov(x_k, x_{k-1}), one wavefront
// per (series, time point), no dependence between them (the closed forms of mf_biggrad_impl.hpp, which does the same with a 256-thread
// workgroup per point on LDS tiles: 11 of the 23 ms of a forward + backward at B = 512, T = 1000, d = 16, 54 of 90 ms at d = 32;
// reference: TensorFlow reverse mode through kalman_filter.py:184-255).
//
//   observation k:   r = y - H m,  HS = H S:      dH = R^-1 (r m^T - HS),  dy = -R^-1 r,  Omega = r r^T + HS H^T
//   transition k-1:  e = m_k - A m_{k-1} - b,     E = X - A S_{k-1} + e m_{k-1}^T,
//                    Psi = S_k - A X^T - X A^T + A S_{k-1} A^T + e e^T,
//                    dA = Q^-1 E,  db = Q^-1 e,  dC = tril(C^-T (C^-1 Psi C^-T - I)),        Q = C C^T
//   k = 0:           the same with A absent, b = mu0, C = cholP0.
// Every product in the P^T Q form: A^T and X^T are read transposed, Q^-1 = Ci^T Ci and Psi are symmetric, Ci^T comes from one pass
// through the LDS image:  A S = tn(A^T, S),  A X^T = tn(A^T, X^T),  X A^T = tn(X^T, A^T),  A S A^T = tn(A^T, tn(S, A^T)),
// Q^-1 E = tn(Q^-1, E),  Ci Psi Ci^T = tn(tn(Psi, Ci^T), Ci^T),  C^-T (.) = tn(Ci, .).
#pragma once
#include "mf_wave.hpp"

namespace mf {
namespace wv {

template <typename T> struct WvGradArgs {
    long B, Tn;
    int d, m;
    const T *mu0, *cholP0, *A, *b, *cholQ, *H, *y, *Rinv;
    int rinv_per_step;
    const T *mean, *cov, *cross, *w;
    T *g_mu0, *g_cholP0, *g_A, *g_b, *g_cholQ, *g_H, *g_y, *g_om;
};

// out += u v^T with u by row and v by column
template <typename T, int NT> MF_DEV void rank1(Mat<T, NT>& out, const RV<T, NT>& u, const CV<T, NT>& v, T sign) {
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        MF_UNROLL for (int tj = 0; tj < NT; ++tj)
            MF_UNROLL for (int e = 0; e < 4; ++e) out.t[ti][tj][e] = __builtin_fma(sign * u.v[ti][e], v.v[tj], out.t[ti][tj][e]);
}
// g = w m (the d x d corner; LOWER: zeros above the diagonal)
template <typename T, int NT, bool LOWER> MF_DEV void store_scaled(T* __restrict__ g, const Mat<T, NT>& m, T w, int d, const Lane& ln) {
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        MF_UNROLL for (int tj = 0; tj < NT; ++tj)
            MF_UNROLL for (int e = 0; e < 4; ++e) {
                const int i = 16 * ti + Tr<T>::row(ln.q, e), j = 16 * tj + ln.r;
                if (i < d && j < d) g[i * d + j] = (!LOWER || j <= i) ? w * m.t[ti][tj][e] : T(0);
            }
}

template <typename T, int NT, int M>
__global__ void __launch_bounds__(64) wave_kf_grad_kernel(WvGradArgs<T> a) {
    using v4 = typename Tr<T>::v4;
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[(NT == 1 ? 2 : NT * NT) * TS];
    const Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long id = blockIdx.x, s = id / a.Tn, k = id % a.Tn;
    const int d = a.d, m = a.m;
    const long dd = long(d) * d, nt = a.Tn - 1;
    const T w = a.w[s];
    const bool tr = k > 0;

    Mat<T, NT> Sk;
    CV<T, NT> mk_cv;
    load_mat<T, NT, S_FULL>(Sk, a.cov + (s * a.Tn + k) * dd, d, false, false, ln);
    load_cv<T, NT>(mk_cv, a.mean + (s * a.Tn + k) * d, d, ln);

    // ---- observation k ----------------------------------------------------------------------------------------------------------
    if (a.H) {
        ObsRows<T, NT, M> ob;
        T Ri[M][M];
        load_rinv<T, M>(Ri, a.rinv_per_step ? a.Rinv + (s * a.Tn + k) * m * m : a.Rinv, m);
        ob.load(a.H + (s * a.Tn + k) * m * d, a.y + (s * a.Tn + k) * m, d, m, ln);
        CV<T, NT> hs[M];
        T r[M];
        MF_UNROLL for (int o = 0; o < M; ++o) {
            tn_mv<T, NT, S_FULL>(hs[o], Sk, ob.hr[o]);                                  // S h_o  (row o of H S)
            r[o] = ob.y[o] - sum16<T>(dot_cv<T, NT>(ob.hc[o], mk_cv));                  // y_o - h_o . m
        }
        MF_UNROLL for (int o = 0; o < M; ++o) {
            CV<T, NT> gh;
            T gy = T(0);
            MF_UNROLL for (int j = 0; j < NT; ++j) gh.v[j] = T(0);
            MF_UNROLL for (int p = 0; p < M; ++p) {
                MF_UNROLL for (int j = 0; j < NT; ++j) gh.v[j] = __builtin_fma(Ri[o][p], r[p] * mk_cv.v[j] - hs[p].v[j], gh.v[j]);
                gy = __builtin_fma(Ri[o][p], r[p], gy);
            }
            if (o < m) {
                MF_UNROLL for (int j = 0; j < NT; ++j) gh.v[j] *= w;
                store_cv<T, NT>(a.g_H + ((s * a.Tn + k) * m + o) * d, gh, d, ln);
                if (threadIdx.x == 0) a.g_y[(s * a.Tn + k) * m + o] = -w * gy;
            }
            MF_UNROLL for (int p = 0; p < M; ++p) {
                const T om = r[o] * r[p] + sum16<T>(dot_cv<T, NT>(hs[o], ob.hc[p]));    // r r^T + (H S) H^T
                if (o < m && p < m && threadIdx.x == 0) a.g_om[(s * a.Tn + k) * m * m + o * m + p] = w * om;
            }
        }
    }
    phase();
    // ---- transition k - 1 (k = 0: the prior) ---------------------------------------------------------------------------------------
    const T* cq = tr ? a.cholQ + (s * nt + k - 1) * dd : a.cholP0 + s * dd;
    const T* off = tr ? a.b + (s * nt + k - 1) * d : a.mu0 + s * d;
    Mat<T, NT> C, Ci, CiT, Qi, Psi;
    v4 c10t = {0, 0, 0, 0};
    load_mat<T, NT, S_LOWER>(C, cq, d, true, true, ln);
    if constexpr (NT == 2) load_tile_t<T>(c10t, cq, d, 1, 0, ln);
    CV<T, NT> e_cv, off_cv, mp_cv;
    load_cv<T, NT>(off_cv, off, d, ln);
    MF_UNROLL for (int j = 0; j < NT; ++j) { e_cv.v[j] = mk_cv.v[j] - off_cv.v[j]; mp_cv.v[j] = T(0); }
    LogAcc<T> la;
    la.init();
    bool bad = false;
    tri_inv_mat<T, NT>(C, c10t, Ci, lds, ln, la, bad);
    CiT.zero();
    transpose<T, NT, S_LOWER>(CiT, Ci, lds, ln);
    tn<T, NT, S_LOWER, S_LOWER, S_UPPER, OP_SET>(Qi, Ci, Ci);                            // Q^-1
    if constexpr (NT == 2) transpose_tile<T>(Qi.t[1][0], Qi.t[0][1], lds, ln);
    Psi = Sk;
    if (tr) {
        Mat<T, NT> AT, Sp, X, XT, E, tmp, SAt;
        RV<T, NT> mp_rv;
        const T* Ak = a.A + (s * nt + k - 1) * dd;
        const T* Xk = a.cross + (s * nt + k - 1) * dd;
        load_mat_t<T, NT>(AT, Ak, d, ln);
        load_mat<T, NT, S_FULL>(Sp, a.cov + (s * a.Tn + k - 1) * dd, d, false, false, ln);
        load_mat<T, NT, S_FULL>(X, Xk, d, false, false, ln);
        load_mat_t<T, NT>(XT, Xk, d, ln);
        load_cv<T, NT>(mp_cv, a.mean + (s * a.Tn + k - 1) * d, d, ln);
        load_rv<T, NT>(mp_rv, a.mean + (s * a.Tn + k - 1) * d, d, ln);
        CV<T, NT> amp;
        tn_mv<T, NT, S_FULL>(amp, AT, mp_rv);                                            // A m_{k-1}
        MF_UNROLL for (int j = 0; j < NT; ++j) e_cv.v[j] -= amp.v[j];
        RV<T, NT> e_rv;
        cv_to_rv<T, NT>(e_rv, e_cv, ln);
        E = X;
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SUB>(E, AT, Sp);                            // X - A S_{k-1}
        rank1<T, NT>(E, e_rv, mp_cv, T(1));                                              //   + e m_{k-1}^T
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SUB>(Psi, AT, XT);                          // - A X^T
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SUB>(Psi, XT, AT);                          // - X A^T
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(SAt, Sp, AT);                          // S_{k-1} A^T
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_ADD>(Psi, AT, SAt);                         // + A S_{k-1} A^T
        rank1<T, NT>(Psi, e_rv, e_cv, T(1));                                             // + e e^T
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(tmp, Qi, E);                           // dA = Q^-1 E
        store_scaled<T, NT, false>(a.g_A + (s * nt + k - 1) * dd, tmp, w, d, ln);
    } else {
        RV<T, NT> e_rv;
        cv_to_rv<T, NT>(e_rv, e_cv, ln);
        rank1<T, NT>(Psi, e_rv, e_cv, T(1));
    }
    phase();
    {
        RV<T, NT> e_rv;
        CV<T, NT> gb;
        cv_to_rv<T, NT>(e_rv, e_cv, ln);
        tn_mv<T, NT, S_FULL>(gb, Qi, e_rv);                                              // db = Q^-1 e
        MF_UNROLL for (int j = 0; j < NT; ++j) gb.v[j] *= w;
        store_cv<T, NT>(tr ? a.g_b + (s * nt + k - 1) * d : a.g_mu0 + s * d, gb, d, ln);
    }
    Mat<T, NT> PsiCt, M2, dC;
    tn<T, NT, S_FULL, S_UPPER, S_FULL, OP_SET>(PsiCt, Psi, CiT);                         // Psi Ci^T
    tn<T, NT, S_FULL, S_UPPER, S_FULL, OP_SET>(M2, PsiCt, CiT);                          // Ci Psi Ci^T
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        MF_UNROLL for (int e = 0; e < 4; ++e) M2.t[ti][ti][e] -= (Tr<T>::row(ln.q, e) == ln.r) ? T(1) : T(0);
    tn<T, NT, S_LOWER, S_FULL, S_FULL, OP_SET>(dC, Ci, M2);                              // C^-T (Ci Psi Ci^T - I)
    store_scaled<T, NT, true>(tr ? a.g_cholQ + (s * nt + k - 1) * dd : a.g_cholP0 + s * dd, dC, w, d, ln);
    (void)bad;
}

}  // namespace wv
}  // namespace mf
