// Row forms (mf_row.hpp) of the kernels behind SymmetricBlockTriDiagonal.upper_diagonal_lower (block_tri_diag.py:438-545) and
// KalmanFilter.posterior_state_space_model (kalman_filter.py:109-182) with few series: the level-0 emit of the reversed factorisation with the posterior chain's
// transitions and factors (par_udl_emit_kernel) and the finish of the offsets (par_post_emit_kernel, chain layout).  (The assembly of the precision, ssm_precision_kernel, stays a lane per block: a row
// form that inverts every transition factor twice measured 552 us against 544 us at d = 9.)  The reversed
// up-sweep is row_chol_up_kernel with ParLevel::rev, the offsets' affine scan row_means_*<REV> with the chain's -U^T as its matrix.
#pragma once
#include "mf_row_grad.hpp"

namespace mf {
namespace row {

// Level-0 emit of the reversed factorisation (par_udl_emit_kernel): chunk c covers positions [c len, ...) (position p = block
// n-1-p) and restarts  Delta_k = D_k - S_k^T Delta_{k+1}^-1 S_k  from the pivot at position c len - 1 (`up`).  Writes U_k^T =
// Delta_{k+1}^-1 S_k (chain: -U_k^T, the posterior transition), chol(Delta_k) (optional) and - chain - chol(Delta_k^-1) at its
// place in the chain (block 0 of all series first, then [B, n-1]).
// Posterior precision + information vector, a row per (series, block k) - used for d >= 10 only: up to d = 9 the lane-per-block
// ssm_precision_kernel is as fast (552 against 544 us at config 4's shape), beyond it that kernel does not exist:
//   diag_k = Q_k^-1 + A_{k+1}^T Q_{k+1}^-1 A_{k+1} (+ H^T R^-1 H),  sub_k = -Q_{k+1}^-1 A_{k+1},
//   eta_k  = Q_k^-1 m_k - A_{k+1}^T Q_{k+1}^-1 m_{k+1} (+ H^T R^-1 y)          (state_space_model.py:431-483, kalman_filter.py:86-101,153-156)
// H == null: the prior precision; eta == null: precision only; y == null: no observation term in eta.
template <typename T, int D, int M>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, false)) row_ssm_precision_kernel(KfArgs<T> a, T* __restrict__ diag,
                                                                                                  T* __restrict__ sub, T* __restrict__ eta) {
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
    const long nt = a.Tn - 1;
    // own part: Q_k^-1 (k = 0: P0^-1) and Q_k^-1 m_k
    T Dn[D], e = T(0);
    {
        T C[D], CiT[D];
        const T* cblk = k == 0 ? a.cholP0 + s * D * D : a.cholQ + (s * nt + k - 1) * D * D;
        load_row_lower<T, D>(cblk, rc, C);
        row_qinv<T, D>(C, t_rcp<T>(cblk[rc * (D + 1)]), r, CiT, Dn);
        if (eta) {
            T mv = k == 0 ? a.mu0[s * D + rc] : a.b[(s * nt + k - 1) * D + rc];
            fence1(mv);
            sfor<D>([&](auto l) { P::template fmac<decltype(l)::value>(e, mv, Dn[decltype(l)::value]); });
        }
    }
    // the transition out of block k (absent at the last block: whole rows take or skip the branch)
    if (k + 1 < a.Tn) {
        asm volatile("s_nop 4");
        const long tid = s * nt + k;
        T C2[D], CiT2[D], Qi2[D], Ar[D], At[D], J[D];
        load_row_lower<T, D>(a.cholQ + tid * D * D, rc, C2);
        row_qinv<T, D>(C2, t_rcp<T>(a.cholQ[tid * D * D + rc * (D + 1)]), r, CiT2, Qi2);
        load_row<T, D>(a.A + tid * D * D, rc, Ar);
        load_col<T, D>(a.A + tid * D * D, rc, At);                     // own row of A^T
        sfor<D>([&](auto j) { J[decltype(j)::value] = T(0); });
        fence(Ar);
        row_mul<T, D, D>(Qi2, Ar, J);                                 // Q^-1 A
        if (st) sfor<D>([&](auto j) { sub[tid * D * D + r * D + decltype(j)::value] = -J[decltype(j)::value]; });
        fence(J);
        row_mul<T, D, D>(At, J, Dn);                                  // + A^T Q^-1 A
        if (eta) {
            T m2 = a.b[tid * D + rc], v = T(0);
            fence1(m2);
            sfor<D>([&](auto l) { P::template fmac<decltype(l)::value>(v, m2, Qi2[decltype(l)::value]); });      // Q^-1 m'
            fence1(v);
            sfor<D>([&](auto l) { P::template fnmac<decltype(l)::value>(e, v, At[decltype(l)::value]); });       // - A^T Q^-1 m'
        }
    }
    if (a.H) {
        asm volatile("s_nop 4");
        const T* __restrict__ Rv = a.Rinv + (a.rinv_per_step ? id * M * M : 0);
        T h[M], RH[M];
        sfor<M>([&](auto o) { h[decltype(o)::value] = a.H[(id * M + decltype(o)::value) * D + rc]; });
        sfor<M>([&](auto o) {
            constexpr int oo = decltype(o)::value;
            T acc = T(0), ry = T(0);
            sfor<M>([&](auto p) {
                acc = __builtin_fma(Rv[oo * M + decltype(p)::value], h[decltype(p)::value], acc);
                if (eta && a.y) ry = __builtin_fma(Rv[oo * M + decltype(p)::value], a.y[id * M + decltype(p)::value], ry);
            });
            RH[oo] = acc;                                             // (R^-1 H)[o][r]
            e = __builtin_fma(h[oo], ry, e);                          // + H^T R^-1 y
        });
        fence(RH);
        sfor<M>([&](auto o) {
            constexpr int oo = decltype(o)::value;
            sfor<D>([&](auto j) { P::template fmac<decltype(j)::value>(Dn[decltype(j)::value], RH[oo], h[oo]); });
        });
    }
    if (st) {
        sfor<D>([&](auto j) { diag[id * D * D + r * D + decltype(j)::value] = Dn[decltype(j)::value]; });
        if (eta) eta[id * D + r] = e;
    }
}

template <typename T, int D>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, false)) row_udl_emit_kernel(long B, long n, long len, long P,
                                                                                             const T* __restrict__ diag, const T* __restrict__ sub,
                                                                                             const T* __restrict__ up, T* __restrict__ ut,
                                                                                             T* __restrict__ chol_d, T* __restrict__ chol_dinv,
                                                                                             int chain, int* info) {
    using Pp = Dpp<T>;
    const RowChunkId q = row_chunk_id<D>(B, P);
    const long p0 = q.c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    const int r = q.r;
    const bool st = q.valid && r < D;
    bool bad = false;
    T L[D], dinv = T(0);                                              // rows of chol(Delta) of the previous position, own 1 / L[r][r]
    // in-place Cholesky of the rows in L; dinv = own reciprocal diagonal element
    auto factor = [&]() {
        sfor<D>([&](auto j) {
            constexpr int jj = decltype(j)::value;
            fence1(L[jj]);
            const T sv = Pp::template bcast<jj>(L[jj]);
            bad |= !(sv > T(0));
            const T inv = row_rsqrt(sv);
            L[jj] *= inv;
            dinv = r == jj ? inv : dinv;
            fence1(L[jj]);
            sfor2<jj + 1, D>([&](auto kq) { Pp::template fnmac<decltype(kq)::value>(L[decltype(kq)::value], L[jj], L[jj]); });
        });
        sfor<D>([&](auto j) { L[decltype(j)::value] = decltype(j)::value <= r ? L[decltype(j)::value] : T(0); });
    };
    sfor<D>([&](auto j) { L[decltype(j)::value] = T(0); });
    if (q.c > 0) {
        load_row<T, D>(up + (q.s * P + q.c - 1) * D * D, q.rc, L);
        factor();
    }
    for (long p = p0; p < p1; ++p) {
        asm volatile("s_nop 4");
        const long k = n - 1 - p;
        T Dl[D];
        load_row<T, D>(diag + (q.s * n + k) * D * D, q.rc, Dl);
        if (p > 0) {
            T U[D];                                                   // own column of S_k, then of L^-1 S, then of Delta^-1 S
            load_col<T, D>(sub + (q.s * (n - 1) + k) * D * D, q.rc, U);
            fence(L);
            fence1(dinv);
            sfor<D>([&](auto kq) {                                    // forward substitution: L^-1 S
                constexpr int kk = decltype(kq)::value;
                U[kk] *= Pp::template bcast<kk>(dinv);
                sfor2<kk + 1, D>([&](auto i) { Pp::template fnmac<decltype(i)::value>(U[decltype(i)::value], L[kk], U[kk]); });
            });
            fence(U);
            sfor<D>([&](auto l) {                                     // Delta_k = D_k - (L^-1 S)^T (L^-1 S)
                constexpr int ll = decltype(l)::value;
                sfor<D>([&](auto j) { Pp::template fnmac<decltype(j)::value>(Dl[decltype(j)::value], U[ll], U[ll]); });
            });
            sfor<D>([&](auto kq) {                                    // backward substitution: L^-T (...)
                constexpr int kk = D - 1 - decltype(kq)::value;
                U[kk] *= Pp::template bcast<kk>(dinv);
                sfor<kk>([&](auto i) { Pp::template fnmac<kk>(U[decltype(i)::value], L[decltype(i)::value], U[kk]); });
            });
            if (st) {
                const T sg = chain ? T(-1) : T(1);
                sfor<D>([&](auto i) { ut[(q.s * (n - 1) + k) * D * D + decltype(i)::value * D + r] = sg * U[decltype(i)::value]; });
            }
        }
        sfor<D>([&](auto j) { L[decltype(j)::value] = Dl[decltype(j)::value]; });
        factor();
        if (chol_d && st) sfor<D>([&](auto j) { chol_d[(q.s * n + k) * D * D + r * D + decltype(j)::value] = L[decltype(j)::value]; });
        if (chain) {
            // chol(Delta_k^-1) = chol(L^-T L^-1)
            T Lc[D], CiT[D], Qm[D];
            sfor<D>([&](auto j) { Lc[decltype(j)::value] = L[decltype(j)::value]; });
            row_qinv<T, D>(Lc, dinv, r, CiT, Qm);
            sfor<D>([&](auto j) {
                constexpr int jj = decltype(j)::value;
                fence1(Qm[jj]);
                const T sv = Pp::template bcast<jj>(Qm[jj]);
                bad |= !(sv > T(0));
                Qm[jj] *= row_rsqrt(sv);
                fence1(Qm[jj]);
                sfor2<jj + 1, D>([&](auto kq) { Pp::template fnmac<decltype(kq)::value>(Qm[decltype(kq)::value], Qm[jj], Qm[jj]); });
            });
            const long ci = k == 0 ? q.s : B + q.s * (n - 1) + k - 1;
            if (st) sfor<D>([&](auto j) { chol_dinv[ci * D * D + r * D + decltype(j)::value] = decltype(j)::value <= r ? Qm[decltype(j)::value] : T(0); });
        }
    }
    if (q.valid && bad && info) raise_info(info);
}

// Finish of the posterior offsets in chain layout (par_post_emit_kernel, chain = 1): x_k = eta_k + (ut_k)^T x_{k+1} (ut holds the
// posterior transition -U_k^T), m_k = Delta_k^-1 x_k = C (C^T x_k) with C = chol(Delta_k^-1) written by row_udl_emit_kernel.
template <typename T, int D>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, false)) row_post_emit_kernel(long B, long n, long len, long P,
                                                                                              const T* __restrict__ ut, const T* __restrict__ eta,
                                                                                              const T* __restrict__ up, T* __restrict__ m_post,
                                                                                              const T* __restrict__ chol_dinv) {
    using Pp = Dpp<T>;
    const RowChunkId q = row_chunk_id<D>(B, P);
    const long p0 = q.c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T x = T(0);
    if (q.c > 0) x = up[(q.s * P + q.c - 1) * D + q.rc];
    for (long p = p0; p < p1; ++p) {
        asm volatile("s_nop 4");
        const long k = n - 1 - p;
        T acc = eta[(q.s * n + k) * D + q.rc];
        if (p > 0) {
            T Uc[D];
            load_col<T, D>(ut + (q.s * (n - 1) + k) * D * D, q.rc, Uc);      // own row of (ut_k)^T
            fence1(x);
            sfor<D>([&](auto l) { Pp::template fmac<decltype(l)::value>(acc, x, Uc[decltype(l)::value]); });
        }
        x = acc;
        const long ci = k == 0 ? q.s : B + q.s * (n - 1) + k - 1;
        T Cr[D], Cc[D];
        load_row_lower<T, D>(chol_dinv + ci * D * D, q.rc, Cr);
        sfor<D>([&](auto l) { Cc[decltype(l)::value] = decltype(l)::value >= q.rc ? chol_dinv[ci * D * D + decltype(l)::value * D + q.rc] : T(0); });
        T u = T(0), mk = T(0);
        fence1(x);
        sfor<D>([&](auto l) { Pp::template fmac<decltype(l)::value>(u, x, Cc[decltype(l)::value]); });           // C^T x
        fence1(u);
        sfor<D>([&](auto l) { Pp::template fmac<decltype(l)::value>(mk, u, Cr[decltype(l)::value]); });          // C (C^T x)
        if (q.valid && q.r < D) m_post[ci * D + q.r] = mk;
    }
}

}   // namespace row
}   // namespace mf
