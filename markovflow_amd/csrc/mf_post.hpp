// posterior_state_space_model (kalman_filter.py:109-182) as ONE backward sweep per series: the posterior precision and the
// information vector (state_space_model.py:431-483, kalman_filter.py:86-101,149-156) are assembled INSIDE the UDU^T sweep
// (block_tri_diag.py:438-545) instead of being written by one kernel and read back by the next.  Per block the sweep reads
// A_k, cholQ_k, b_k, H_k, y_k once ((2 d^2 + d + m d + m) s bytes) and writes the posterior chain's A'_k, cholQ'_k, b'_k
// ((2 d^2 + d) s): 4 d^2 s per block instead of the 8 d^2 s of assembly + sweep.  Every transition feeds two blocks (Q_k^-1
// into block k+1, A_k^T Q_k^-1 A_k and the coupling into block k): its inverted factor is formed when block k+1 is visited
// and carried to block k in registers.  One lane per series: for batches that fill the chip.
//   Delta_{T-1} = D_{T-1},  Delta_k = D_k - S_k^T Delta_{k+1}^-1 S_k,  U_k^T = Delta_{k+1}^-1 S_k,  x_k = eta_k - U_k x_{k+1},
//   posterior chain: A'_{k+1} = -U_k^T,  (mu0', b'_k) = Delta_k^-1 x_k,  (cholP0', cholQ'_k) = chol(Delta_k^-1).
#pragma once
#include "mf_kernels.hpp"

namespace mf {

template <typename T, int D, int M>
__global__ void __launch_bounds__(64) kf_posterior_chain_kernel(KfArgs<T> a, T* __restrict__ a_post, T* __restrict__ mu0_post,
                                                                T* __restrict__ b_post, T* __restrict__ cp0_post,
                                                                T* __restrict__ cq_post) {
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.B) return;
    const long n = a.Tn;
    const int m = a.m;
    LogAcc<T> la;
    la.init();
    bool bad = false;
    T Lp[D][D], Lpi[D], xp[D];           // chol(Delta_{k+1}), 1 / its diagonal, x_{k+1}
    T Cn[D][D], wn[D];                   // transition k -> k+1: C_k^-1 and C_k^-1 b_k (formed as block k+1's own part)
    MF_UNROLL for (int i = 0; i < D; ++i) {
        Lpi[i] = T(0); xp[i] = T(0); wn[i] = T(0);
        MF_UNROLL for (int j = 0; j < D; ++j) { Lp[i][j] = T(0); Cn[i][j] = T(0); }
    }
    // The loads of a block are issued together and, where a second set fits the registers, one block AHEAD of their use: with
    // a quarter of a wave per SIMD (16 384 series) nothing else hides a dependent global load.
    constexpr int MO = (M > 0) ? M : 1;
    struct Step { T C[D][D]; T mv[D]; T A[D][D]; T h[MO * D]; T y[MO]; };
    auto load = [&](long k, Step& d) {
        load_lower<T, D>(k == 0 ? a.cholP0 + s * D * D : a.cholQ + (s * (n - 1) + k - 1) * D * D, d.C);
        load_vec<T, D>(k == 0 ? a.mu0 + s * D : a.b + (s * (n - 1) + k - 1) * D, d.mv);
        const long kt = k + 1 < n ? k : (n > 1 ? n - 2 : 0);                      // last block: clamped, unused
        if (n > 1) load_mat<T, D, D>(a.A + (s * (n - 1) + kt) * D * D, d.A);
        if (M > 0) {
            MF_UNROLL for (int e = 0; e < MO * D; ++e) d.h[e] = a.H[(s * n + k) * MO * D + e];
            MF_UNROLL for (int e = 0; e < MO; ++e) d.y[e] = a.y[(s * n + k) * MO + e];
        }
    };
#ifndef MF_POST_PF
#define MF_POST_PF 1
#endif
    constexpr bool PF = MF_POST_PF && (M > 0) && (sizeof(T) == 4 ? (D <= 8) : (D <= 6));
    Step cur, nxt;
    if (PF) load(n - 1, cur);
    for (long k = n - 1; k >= 0; --k) {
        if (PF) load(k > 0 ? k - 1 : 0, nxt);
        else load(k, cur);
        __builtin_amdgcn_sched_barrier(0);
        const bool has_next = k + 1 < n;
        T U[D][D];
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) U[i][j] = cur.A[i][j];
        const T (&C)[D][D] = cur.C;
        const T (&mv)[D] = cur.mv;
        // ---- block k of the posterior precision and of the information vector ----------------------------------------------
        T Co[D][D], wo[D], Dl[D][D], x[D];
        tri_inv_lower<T, D>(C, Co, la, bad);
        la.init();
        trimul_lower_vec<T, D>(Co, mv, wo);
        trimulT_self_lower<T, D>(Co, Dl);                         // Q_{k-1}^-1   (P0^-1 for k = 0)
        trimulT_lower_vec<T, D>(Co, wo, x);                       // Q_{k-1}^-1 b_{k-1}
        {
            const T* Ri = a.rinv_per_step ? a.Rinv + (s * n + k) * m * m : a.Rinv;
            if (M > 0) Obs<T, D, M>::apply(cur.h, cur.y, Ri, m, Dl, x);
            else Obs<T, D, M>::apply(a.H + (s * n + k) * m * D, a.y + (s * n + k) * m, Ri, m, Dl, x);
        }
        if (has_next) {
            T btw[D];
            trimul_lower_inplace<T, D, D>(Cn, U);                 // B = C_k^-1 A_k
            syrk_tn_lower<T, D, D>(U, Dl, T(1));                  // + A_k^T Q_k^-1 A_k
            gemv_t<T, D, D>(U, wn, btw);
            MF_UNROLL for (int i = 0; i < D; ++i) x[i] -= btw[i]; // - A_k^T Q_k^-1 b_k
            neg_trimulT_lower_inplace<T, D, D>(Cn, U);            // S_k = -Q_k^-1 A_k
            // ---- U D U^T step -------------------------------------------------------------------------------------------
            trsm_left_lower<T, D, D>(Lp, Lpi, U);                 // L^-1 S
            syrk_tn_lower<T, D, D>(U, Dl, T(-1));                 // Delta_k = D_k - S^T Delta_{k+1}^-1 S
            trsm_left_lower_t<T, D, D>(Lp, Lpi, U);               // U_k^T = Delta_{k+1}^-1 S
            T ux[D];
            gemv_t<T, D, D>(U, xp, ux);
            MF_UNROLL for (int i = 0; i < D; ++i) {
                x[i] -= ux[i];
                MF_UNROLL for (int j = 0; j < D; ++j) U[i][j] = -U[i][j];
            }
            store_mat<T, D, D>(a_post + (s * (n - 1) + k) * D * D, U);          // A'_{k+1} = -U_k^T
        }
        chol_lower<T, D>(Dl, Lpi, la, bad);
        la.init();
        MF_UNROLL for (int i = 0; i < D; ++i) {
            xp[i] = x[i];
            wn[i] = wo[i];
            MF_UNROLL for (int j = 0; j <= i; ++j) { Lp[i][j] = Dl[i][j]; Cn[i][j] = Co[i][j]; }
        }
        trsv_lower<T, D>(Lp, Lpi, x);
        trsv_lower_t<T, D>(Lp, Lpi, x);                           // Delta_k^-1 x_k
        store_vec<T, D>(k == 0 ? mu0_post + s * D : b_post + (s * (n - 1) + k - 1) * D, x);
        {
            T Linv[D][D], Q[D][D], Qi[D];
            tri_inv_lower<T, D>(Lp, Linv, la, bad);
            la.init();
            trimulT_self_lower<T, D>(Linv, Q);                    // Delta_k^-1 = L^-T L^-1
            chol_lower<T, D>(Q, Qi, la, bad);
            la.init();
            store_lower<T, D>(k == 0 ? cp0_post + s * D * D : cq_post + (s * (n - 1) + k - 1) * D * D, Q);
        }
        if (PF) cur = nxt;
    }
    if (bad && a.info) raise_info(a.info);
}

}  // namespace mf
