// HIP kernels of the Kalman / block-tridiagonal hot path, templated on scalar type T and state dim D.
//
// Execution model (MI355X / gfx950): ONE LANE = ONE sub-problem, everything register resident
// (mf_small.hpp), 64 independent sub-problems per wavefront, no cross-lane traffic.  The sequential
// scan over time is parallelised by PARTITIONED block elimination ("one level of block cyclic
// reduction with chunk size c"): a chain of n blocks is cut into chunks; every chunk eliminates its
// interior blocks in natural order while carrying the fill-in ("spike") towards the separator on its
// left, and hands a reduced block-tridiagonal system of one block per chunk to the next level.
// log|M| and r^T M^-1 r are invariant under the elimination order, so the log-likelihood comes out
// exact (to rounding) from any partition.  Notation follows SURVEY.md Appendix B.
#pragma once
#include "mf_small.hpp"

namespace mf {

// Reduced block-tridiagonal system with n blocks per series ("RedSys").  For block j:
//   pivot   D'_j   = Dv[j] + GU[j+1]   (GU[j+1] absent for the last block)
//   rhs     eta'_j = tv[j] + gU[j+1]
//   coupling M'_{j,j-1} = F[j]          (F[0] unused)
//   sc[j]   additive scalar already earned by the blocks folded into j
// GU / gU may be null (plain user-supplied systems).
template <typename T> struct RedSys {
    T* Dv; T* GU; T* F; T* tv; T* gU; T* sc;
    long n;          // blocks per series
    long f_stride;   // blocks per series in F's allocation (n, or n-1 with f_off = -1 for a user `sub`)
    long f_off;      // F block index for coupling j is (j + f_off)
};

// Elimination state of one chunk.
template <typename T, int D, bool SPIKE> struct Elim {
    T Phi[D][D];   // lower: partial pivot of the current block, then its Cholesky factor
    T Li[D];
    T t[D];        // partial rhs of the current block, then z = L^-1 t
    T X[D][D];     // coupling current block <-> left separator, then V = L^-1 X
    T GU[D][D];    // lower: accumulated contribution to the left separator's pivot
    T gU[D];       // accumulated contribution to the left separator's rhs
    T quad;        // sum |z|^2
    LogAcc<T> laL; // prod diag(L)
    bool bad;

    MF_HD void init() {
        MF_UNROLL for (int i = 0; i < D; ++i) {
            t[i] = T(0); gU[i] = T(0); Li[i] = T(0);
            MF_UNROLL for (int j = 0; j < D; ++j) { Phi[i][j] = T(0); X[i][j] = T(0); GU[i][j] = T(0); }
        }
        quad = T(0);
        laL.init();
        bad = false;
    }
    // Factor the (complete) pivot in Phi, solve for z and the spike V, fold V into the separator.
    MF_HD void eliminate() {
        eliminate_main();
        eliminate_spike();
    }
    MF_HD void eliminate_main() {
        chol_lower<T, D>(Phi, Li, laL, bad);
        laL.renorm();
        trsv_lower<T, D>(Phi, Li, t);
        quad += dot_self<T, D>(t);
    }
    MF_HD void eliminate_spike() {
        if (SPIKE) {
            trsm_left_lower<T, D, D>(Phi, Li, X);
            syrk_tn_lower<T, D, D>(X, GU, T(-1));
            T vz[D];
            gemv_t<T, D, D>(X, t, vz);
            MF_UNROLL for (int i = 0; i < D; ++i) gU[i] -= vz[i];
        }
    }
    // After eliminate(): move to the next block whose coupling to the eliminated one is W L^T
    // (W = S L^-T already formed by the caller); Dn / rn are the next block's own pivot / rhs parts.
    MF_HD void advance(const T (&W)[D][D], const T (&Dn)[D][D], const T (&rn)[D]) {
        T wz[D];
        gemv_n<T, D, D>(W, t, wz);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            t[i] = rn[i] - wz[i];
            MF_UNROLL for (int j = 0; j <= i; ++j) Phi[i][j] = Dn[i][j];
        }
        syrk_nt_lower<T, D, D>(W, Phi, T(-1));
        if (SPIKE) neg_mul_inplace<T, D>(W, X);
    }
};

template <typename T, int D, bool SPIKE>
MF_HD void store_chunk(const RedSys<T>& out, long idx, const Elim<T, D, SPIKE>& E, T scalar) {
    store_sym<T, D>(out.Dv + idx * D * D, E.Phi);
    store_vec<T, D>(out.tv + idx * D, E.t);
    store_sym<T, D>(out.GU + idx * D * D, E.GU);
    store_vec<T, D>(out.gU + idx * D, E.gU);
    store_mat<T, D, D>(out.F + idx * D * D, E.X);
    out.sc[idx] = scalar;
}

// -------------------------------------------------------------------------------------------------
// K0 level 0: state-space model + observations -> reduced system of P blocks per series.
// Replaces, fused and without materialising anything:  StateSpaceModel._build_precision
// (state_space_model.py:431-483), _k_inv_post (kalman_filter.py:86-101), the natural-order banded
// Cholesky (block_tri_diag.py:423-436), the forward solve (:339-351) and the log-dets
// (kalman_filter.py:229-253).  The prior mean enters through the information vector
// eta = G^T S^-1 y + K^-1 mu  (kalman_filter.py:129-145) instead of the marginal-mean recursion, so a
// chunk needs nothing from its predecessors.
// -------------------------------------------------------------------------------------------------
template <typename T> struct KfArgs {
    long B, Tn;          // series, time points
    int m;               // output dim (runtime, <= MAXM)
    const T* mu0; const T* cholP0; const T* A; const T* b; const T* cholQ;   // [B,d] [B,d,d] [B,T-1,d,d] [B,T-1,d] [B,T-1,d,d]
    const T* H; const T* y;                                                 // [B,T,m,d] [B,T,m]
    const T* Rinv; int rinv_per_step;                                       // [m,m] or [B,T,m,m]
    long P;              // chunks per series
    int* info;
    int debug;           // timing experiments only: bit 0 = DMA descriptors with zero records (no memory traffic)
    const T* weights;    // gradient kernel only: per-series factor applied to every output (NULL = 1)
};

constexpr int MF_MAXM = 4;

template <typename T, int D, int M> struct Obs {
    // adds H^T R^-1 H to Phi (lower), H^T R^-1 y to t and returns y^T R^-1 y
    static MF_HD T apply(const T* __restrict__ Hk, const T* __restrict__ yk, const T* __restrict__ Ri, int m,
                          T (&Phi)[D][D], T (&t)[D]) {
        constexpr int MM = (M > 0) ? M : MF_MAXM;
        T h[MM][D], yv[MM], rh[MM][D], ry[MM];
        MF_UNROLL for (int o = 0; o < MM; ++o) {
            const bool on = (M > 0) || (o < m);
            yv[o] = on ? yk[o] : T(0);
            MF_UNROLL for (int i = 0; i < D; ++i) h[o][i] = on ? Hk[o * D + i] : T(0);
        }
        MF_UNROLL for (int o = 0; o < MM; ++o) {
            ry[o] = T(0);
            MF_UNROLL for (int i = 0; i < D; ++i) rh[o][i] = T(0);
            MF_UNROLL for (int p = 0; p < MM; ++p) {
                const bool on = (M > 0) || (o < m && p < m);
                const T r = on ? Ri[o * ((M > 0) ? M : m) + p] : T(0);
                ry[o] += r * yv[p];
                MF_UNROLL for (int i = 0; i < D; ++i) rh[o][i] += r * h[p][i];
            }
        }
        T yry = T(0);
        MF_UNROLL for (int o = 0; o < MM; ++o) {
            yry += yv[o] * ry[o];
            MF_UNROLL for (int i = 0; i < D; ++i) {
                t[i] += h[o][i] * ry[o];
                MF_UNROLL for (int j = 0; j <= i; ++j) Phi[i][j] += h[o][i] * rh[o][j];
            }
        }
        return yry;
    }
};

template <typename T, int D, int M, bool SPIKE>
__global__ void __launch_bounds__(64) kf_chunk_kernel(KfArgs<T> a, RedSys<T> out) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= a.B * a.P) return;
    const long s = id / a.P, c = id % a.P;
    const long k0 = (c * a.Tn) / a.P, k1 = ((c + 1) * a.Tn) / a.P;   // blocks [k0, k1)
    const int m = a.m;
    const T* As = a.A + s * (a.Tn - 1) * D * D;
    const T* Qs = a.cholQ + s * (a.Tn - 1) * D * D;
    const T* bs = a.b + s * (a.Tn - 1) * D;
    const T* Hs = a.H + s * a.Tn * m * D;
    const T* ys = a.y + s * a.Tn * m;

    Elim<T, D, SPIKE> E;
    E.init();
    LogAcc<T> laC;
    laC.init();
    T acc_yry = T(0), acc_ww = T(0);
    long first_bad = -1;          // the block whose elimination step first met a non-positive pivot

    for (long k = k0; k < k1; ++k) {
        T C[D][D], Ci[D][D], mvec[D], w[D];
        if (k == 0) {
            load_lower<T, D>(a.cholP0 + s * D * D, C);
            load_vec<T, D>(a.mu0 + s * D, mvec);
        } else {
            load_lower<T, D>(Qs + (k - 1) * D * D, C);
            load_vec<T, D>(bs + (k - 1) * D, mvec);
        }
        tri_inv_lower<T, D>(C, Ci, laC, E.bad);
        laC.renorm();
        trimul_lower_vec<T, D>(Ci, mvec, w);
        acc_ww += dot_self<T, D>(w);

        T Dn[D][D], rn[D];
        trimulT_self_lower<T, D>(Ci, Dn);          // Q_k^-1
        trimulT_lower_vec<T, D>(Ci, w, rn);        // Q_k^-1 m_k
        const T* Ri = a.rinv_per_step ? a.Rinv + (s * a.Tn + k) * m * m : a.Rinv;
        acc_yry += Obs<T, D, M>::apply(Hs + k * m * D, ys + k * m, Ri, m, Dn, rn);

        if (k == 0) {
            MF_UNROLL for (int i = 0; i < D; ++i) {
                E.t[i] = rn[i];
                MF_UNROLL for (int j = 0; j <= i; ++j) E.Phi[i][j] = Dn[i][j];
            }
            continue;
        }
        T Bm[D][D];
        {
            T Am[D][D];
            load_mat<T, D, D>(As + (k - 1) * D * D, Am);
            trimul_lower<T, D, D>(Ci, Am, Bm);     // B = C^-1 A
        }
        T btw[D], W[D][D];
        gemv_t<T, D, D>(Bm, w, btw);               // A^T Q^-1 m
        if (k == k0) {
            // block k-1 is the separator on the left: it is not eliminated here.
            syrk_tn_lower<T, D, D>(Bm, E.GU, T(1));
            MF_UNROLL for (int i = 0; i < D; ++i) E.gU[i] = -btw[i];
            trimulT_lower<T, D, D>(Ci, Bm, W);
            MF_UNROLL for (int i = 0; i < D; ++i) {
                E.t[i] = rn[i];
                MF_UNROLL for (int j = 0; j < D; ++j) { E.X[i][j] = -W[i][j]; }
                MF_UNROLL for (int j = 0; j <= i; ++j) E.Phi[i][j] = Dn[i][j];
            }
        } else {
            syrk_tn_lower<T, D, D>(Bm, E.Phi, T(1));    // D_{k-1} complete
            MF_UNROLL for (int i = 0; i < D; ++i) E.t[i] -= btw[i];
            E.eliminate();
            if (E.bad && first_bad < 0) first_bad = k - 1;
            trsm_right_lower_t<T, D, D>(E.Phi, E.Li, Bm);   // Y = B L^-T
            trimulT_lower<T, D, D>(Ci, Bm, W);              // -W = C^-T Y
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) W[i][j] = -W[i][j];
            E.advance(W, Dn, rn);
        }
    }
    const T scalar = T(-0.5) * (acc_yry + acc_ww) + T(0.5) * E.quad - laC.value() - E.laL.value();
    store_chunk<T, D, SPIKE>(out, id, E, scalar);
    if (E.bad && a.info) raise_pivot(a.info, s * a.Tn + (first_bad < 0 ? k0 : first_bad));
}

// -------------------------------------------------------------------------------------------------
// Generic level: RedSys(n) -> RedSys(P), same elimination on explicit blocks.
// -------------------------------------------------------------------------------------------------
template <typename T, int D>
MF_DEV void load_red_block(const RedSys<T>& in, long s, long j, T (&Dn)[D][D], T (&rn)[D], T& sc) {
    const long idx = s * in.n + j;
    load_lower<T, D>(in.Dv + idx * D * D, Dn);
    load_vec<T, D>(in.tv + idx * D, rn);
    sc = in.sc ? in.sc[idx] : T(0);
    if (in.GU && j + 1 < in.n) {
        T g[D][D], gv[D];
        load_lower<T, D>(in.GU + (idx + 1) * D * D, g);
        load_vec<T, D>(in.gU + (idx + 1) * D, gv);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            rn[i] += gv[i];
            MF_UNROLL for (int jj = 0; jj <= i; ++jj) Dn[i][jj] += g[i][jj];
        }
    }
}

// All loads of one block of a reduction level, issued together and branch-free (a load under a divergent branch is
// followed by a wait at the merge; the plain form paid three dependent round trips per block: pivot part, the contribution of
// the next block's interior, the coupling).  Block 0 has no coupling and the last block no GU term: clamped, flagged.
template <typename T, int D> struct RedStep {
    T Dn[D][D];   // lower
    T rn[D];
    T g[D][D];    // lower: GU of block j+1
    T gv[D];
    T W[D][D];
    T sc;
    bool hasg;
};
template <typename T, int D> MF_DEV void load_red_step(const RedSys<T>& in, long s, long j, RedStep<T, D>& d) {
    const long idx = s * in.n + j;
    load_lower<T, D>(in.Dv + idx * D * D, d.Dn);
    load_vec<T, D>(in.tv + idx * D, d.rn);
    d.sc = in.sc ? in.sc[idx] : T(0);
    d.hasg = in.GU && (j + 1 < in.n);
    if (in.GU) {
        const long i2 = (j + 1 < in.n) ? idx + 1 : idx;
        load_lower<T, D>(in.GU + i2 * D * D, d.g);
        load_vec<T, D>(in.gU + i2 * D, d.gv);
    }
    if (in.n > 1) load_mat<T, D, D>(in.F + (s * in.f_stride + (j > 0 ? j : 1) + in.f_off) * D * D, d.W);
}
template <typename T, int D> MF_DEV void red_step_fold(RedStep<T, D>& d) {
    MF_UNROLL for (int i = 0; i < D; ++i) {
        d.rn[i] += d.hasg ? d.gv[i] : T(0);
        MF_UNROLL for (int jj = 0; jj <= i; ++jj) d.Dn[i][jj] += d.hasg ? d.g[i][jj] : T(0);
    }
}

template <typename T, int D, bool SPIKE>
__global__ void __launch_bounds__(64) red_chunk_kernel(RedSys<T> in, RedSys<T> out, long B, long P, int* info) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * P) return;
    const long s = id / P, c = id % P;
    const long k0 = (c * in.n) / P, k1 = ((c + 1) * in.n) / P;
    Elim<T, D, SPIKE> E;
    E.init();
    T acc_sc = T(0);
    // small blocks: the next block's loads are in flight while the current one is eliminated (a second set of block data fits
    // the registers next to the spike up to d = 4 in fp64, d = 6 in fp32); these levels are pure latency - 8 dependent steps
    // on a fraction of the chip - and are a third of an evaluation at BASELINE config 2 (B=256, T=4096, d=4)
#ifndef MF_RED_PF
#define MF_RED_PF 1
#endif
    constexpr bool PF = MF_RED_PF && (sizeof(T) == 4 ? (D <= 6) : (D <= 4));
    RedStep<T, D> d, nxt;
    if (PF && k0 < k1) load_red_step<T, D>(in, s, k0, d);
    for (long k = k0; k < k1; ++k) {
        if (PF) load_red_step<T, D>(in, s, k + 1 < k1 ? k + 1 : k, nxt);
        else load_red_step<T, D>(in, s, k, d);
        __builtin_amdgcn_sched_barrier(0);
        red_step_fold<T, D>(d);
        acc_sc += d.sc;
        if (k == 0) {
            MF_UNROLL for (int i = 0; i < D; ++i) {
                E.t[i] = d.rn[i];
                MF_UNROLL for (int j = 0; j <= i; ++j) E.Phi[i][j] = d.Dn[i][j];
            }
            if (PF) d = nxt;
            continue;
        }
        if (k == k0) {
            MF_UNROLL for (int i = 0; i < D; ++i) {
                E.t[i] = d.rn[i];
                MF_UNROLL for (int j = 0; j < D; ++j) E.X[i][j] = d.W[i][j];
                MF_UNROLL for (int j = 0; j <= i; ++j) E.Phi[i][j] = d.Dn[i][j];
            }
        } else {
            E.eliminate();
            trsm_right_lower_t<T, D, D>(E.Phi, E.Li, d.W);    // W = S L^-T
            E.advance(d.W, d.Dn, d.rn);
        }
        if (PF) d = nxt;
    }
    const T scalar = acc_sc + T(0.5) * E.quad - E.laL.value();
    store_chunk<T, D, SPIKE>(out, id, E, scalar);
    if (E.bad && info) raise_info(info);
}

// Final level: one lane per series walks the remaining n blocks; out[s] = add_const + sum of scalars.  The next block is
// loaded while the current one is eliminated where two sets of block data fit the registers (fp32; fp64 up to d = 6).
template <typename T, int D>
__global__ void __launch_bounds__(64) red_final_kernel(RedSys<T> in, long B, T add_const, T* __restrict__ out, int* info) {
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    Elim<T, D, false> E;
    E.init();
    T acc_sc = T(0);
    constexpr bool PF = sizeof(T) == 4 || D <= 6;
    if constexpr (!PF) {
        // large blocks in fp64: the grouped form spills; pivot part first, coupling where it is used
        for (long k = 0; k < in.n; ++k) {
            T Dn[D][D], rn[D], sc;
            load_red_block<T, D>(in, s, k, Dn, rn, sc);
            acc_sc += sc;
            if (k == 0) {
                MF_UNROLL for (int i = 0; i < D; ++i) {
                    E.t[i] = rn[i];
                    MF_UNROLL for (int j = 0; j <= i; ++j) E.Phi[i][j] = Dn[i][j];
                }
            } else {
                T W[D][D];
                load_mat<T, D, D>(in.F + (s * in.f_stride + k + in.f_off) * D * D, W);
                E.eliminate();
                trsm_right_lower_t<T, D, D>(E.Phi, E.Li, W);
                E.advance(W, Dn, rn);
            }
        }
        E.eliminate();
        out[s] = add_const + acc_sc + T(0.5) * E.quad - E.laL.value();
        if (E.bad && info) raise_info(info);
        return;
    }
    RedStep<T, D> cur, nxt;
    if (in.n > 0) load_red_step<T, D>(in, s, 0, cur);
    for (long k = 0; k < in.n; ++k) {
        load_red_step<T, D>(in, s, k + 1 < in.n ? k + 1 : k, nxt);
        __builtin_amdgcn_sched_barrier(0);
        red_step_fold<T, D>(cur);
        acc_sc += cur.sc;
        if (k == 0) {
            MF_UNROLL for (int i = 0; i < D; ++i) {
                E.t[i] = cur.rn[i];
                MF_UNROLL for (int j = 0; j <= i; ++j) E.Phi[i][j] = cur.Dn[i][j];
            }
        } else {
            E.eliminate();
            trsm_right_lower_t<T, D, D>(E.Phi, E.Li, cur.W);
            E.advance(cur.W, cur.Dn, cur.rn);
        }
        cur = nxt;
    }
    E.eliminate();
    out[s] = add_const + acc_sc + T(0.5) * E.quad - E.laL.value();
    if (E.bad && info) raise_info(info);
}

// -------------------------------------------------------------------------------------------------
// Operator kernels (natural order, one lane per series) - the API-parity forms of
// cholesky / solve / dense_mult / abs_log_det / block_diagonal_of_inverse / upper_diagonal_lower.
// -------------------------------------------------------------------------------------------------

// K1  SymmetricBlockTriDiagonal.cholesky (block_tri_diag.py:423-436)
template <typename T, int D>
__global__ void __launch_bounds__(64) btd_cholesky_kernel(long B, long n, const T* __restrict__ diag,
                                                          const T* __restrict__ sub, T* __restrict__ ldiag,
                                                          T* __restrict__ lsub, int* info) {
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    T L[D][D], Li[D];
    LogAcc<T> la;
    la.init();
    bool bad = false;
    long first_bad = -1;
    for (long k = 0; k < n; ++k) {
        T S[D][D];
        load_lower<T, D>(diag + (s * n + k) * D * D, S);
        if (sub && k > 0) {
            T W[D][D];
            load_mat<T, D, D>(sub + (s * (n - 1) + k - 1) * D * D, W);
            trsm_right_lower_t<T, D, D>(L, Li, W);
            store_mat<T, D, D>(lsub + (s * (n - 1) + k - 1) * D * D, W);
            syrk_nt_lower<T, D, D>(W, S, T(-1));
        }
        chol_lower<T, D>(S, Li, la, bad);
        if (bad && first_bad < 0) first_bad = k;
        la.init();
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) L[i][j] = S[i][j];
        store_lower<T, D>(ldiag + (s * n + k) * D * D, L);
    }
    if (bad && info) raise_pivot(info, s * n + first_bad);
}

// K2  LowerTriangularBlockTriDiagonal.solve (block_tri_diag.py:339-351); rhs series r uses factor r % Bl
template <typename T, int D>
__global__ void __launch_bounds__(64) btd_solve_kernel(long Bl, long Br, long n, const T* __restrict__ ldiag,
                                                       const T* __restrict__ lsub, const T* __restrict__ rhs,
                                                       T* __restrict__ out, int transpose) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= Br) return;
    const long s = r % Bl;
    T z[D];
    MF_UNROLL for (int i = 0; i < D; ++i) z[i] = T(0);
    for (long kk = 0; kk < n; ++kk) {
        const long k = transpose ? n - 1 - kk : kk;
        T L[D][D], Li[D], x[D];
        load_lower<T, D>(ldiag + (s * n + k) * D * D, L);
        MF_UNROLL for (int i = 0; i < D; ++i) Li[i] = T(1) / L[i][i];
        load_vec<T, D>(rhs + (r * n + k) * D, x);
        if (lsub && kk > 0) {
            T W[D][D], wz[D];
            if (!transpose) {
                load_mat<T, D, D>(lsub + (s * (n - 1) + k - 1) * D * D, W);
                gemv_n<T, D, D>(W, z, wz);
            } else {
                load_mat<T, D, D>(lsub + (s * (n - 1) + k) * D * D, W);
                gemv_t<T, D, D>(W, z, wz);
            }
            MF_UNROLL for (int i = 0; i < D; ++i) x[i] -= wz[i];
        }
        if (!transpose) trsv_lower<T, D>(L, Li, x); else trsv_lower_t<T, D>(L, Li, x);
        MF_UNROLL for (int i = 0; i < D; ++i) z[i] = x[i];
        store_vec<T, D>(out + (r * n + k) * D, z);
    }
}

// K3  BlockTriDiagonal.dense_mult (block_tri_diag.py:175-199): one lane per (series, block)
//     mode 0: lower-triangular M x; 1: M^T x; 2: symmetric M x (lower triangle of diag mirrored)
template <typename T, int D>
__global__ void __launch_bounds__(256) btd_matvec_kernel(long Bl, long Br, long n, const T* __restrict__ diag,
                                                         const T* __restrict__ sub, const T* __restrict__ x,
                                                         T* __restrict__ out, int mode) {
    const long id_raw = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = id_raw < Br * n;
    const long id = valid ? id_raw : Br * n - 1;       // lanes past the end redo the last block (cross-lane moves need them) and store nothing
    const long r = id / n, k = id % n, s = r % Bl;
    T Dk[D][D], xv[D], acc[D];
    load_lower<T, D>(diag + (s * n + k) * D * D, Dk);
    load_vec<T, D>(x + (r * n + k) * D, xv);
    MF_UNROLL for (int i = 0; i < D; ++i) {
        T a = T(0);
        MF_UNROLL for (int j = 0; j < D; ++j) {
            T e;
            if (mode == 0) e = (j <= i) ? Dk[i][j] : T(0);
            else if (mode == 1) e = (j >= i) ? Dk[j][i] : T(0);
            else e = (j <= i) ? Dk[i][j] : Dk[j][i];
            a += e * xv[j];
        }
        acc[i] = a;
    }
    if (sub) {
        if (mode == 2) {
            // symmetric product: every coupling block is needed by two neighbouring blocks.  A lane loads S_k once, keeps
            // S_k^T x_{k+1} for its own block and hands S_k x_k to the lane of block k+1 (one cross-lane move of d values);
            // only the first lane of a wave (and of a series) loads the block of its left neighbour itself.
            T u[D];
            MF_UNROLL for (int i = 0; i < D; ++i) u[i] = T(0);
            if (k + 1 < n) {
                T S[D][D], xn[D], t[D];
                load_mat<T, D, D>(sub + (s * (n - 1) + k) * D * D, S);
                load_vec<T, D>(x + (r * n + k + 1) * D, xn);
                gemv_t<T, D, D>(S, xn, t);
                gemv_n<T, D, D>(S, xv, u);
                MF_UNROLL for (int i = 0; i < D; ++i) acc[i] += t[i];
            }
            const int lane = threadIdx.x & 63;
            T from_left[D];
            MF_UNROLL for (int i = 0; i < D; ++i) from_left[i] = __shfl_up(u[i], 1, 64);
            if (k > 0) {
                if (lane == 0) {
                    T S[D][D], xp[D];
                    load_mat<T, D, D>(sub + (s * (n - 1) + k - 1) * D * D, S);
                    load_vec<T, D>(x + (r * n + k - 1) * D, xp);
                    gemv_n<T, D, D>(S, xp, from_left);
                }
                MF_UNROLL for (int i = 0; i < D; ++i) acc[i] += from_left[i];
            }
        } else if (mode == 0 && k > 0) {
            T S[D][D], xp[D], t[D];
            load_mat<T, D, D>(sub + (s * (n - 1) + k - 1) * D * D, S);
            load_vec<T, D>(x + (r * n + k - 1) * D, xp);
            gemv_n<T, D, D>(S, xp, t);
            MF_UNROLL for (int i = 0; i < D; ++i) acc[i] += t[i];
        } else if (mode == 1 && k + 1 < n) {
            T S[D][D], xn[D], t[D];
            load_mat<T, D, D>(sub + (s * (n - 1) + k) * D * D, S);
            load_vec<T, D>(x + (r * n + k + 1) * D, xn);
            gemv_t<T, D, D>(S, xn, t);
            MF_UNROLL for (int i = 0; i < D; ++i) acc[i] += t[i];
        }
    }
    if (valid) store_vec<T, D>(out + (r * n + k) * D, acc);
}

// abs_log_det (block_tri_diag.py:353-366): one wavefront per series
template <typename T, int D>
__global__ void __launch_bounds__(64) btd_logdet_kernel(long B, long n, const T* __restrict__ ldiag, T* __restrict__ out) {
    const long s = blockIdx.x;
    T acc = T(0);
    for (long e = threadIdx.x; e < n * D; e += 64) {
        const long k = e / D;
        const int i = (int)(e % D);
        const T v = ldiag[(s * n + k) * D * D + i * D + i];
        acc += T(0.5) * log(v * v);
    }
    MF_UNROLL for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (threadIdx.x == 0) out[s] = acc;
}

// K4  block_diagonal_of_inverse (block_tri_diag.py:318-337), block Takahashi, backward
template <typename T, int D>
__global__ void __launch_bounds__(64) btd_diag_of_inverse_kernel(long B, long n, const T* __restrict__ ldiag,
                                                                 const T* __restrict__ lsub, T* __restrict__ odiag,
                                                                 T* __restrict__ osub) {
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    T Sig[D][D];   // full symmetric Sigma_{k+1,k+1}
    for (long k = n - 1; k >= 0; --k) {
        T L[D][D], Li[D], Linv[D][D];
        load_lower<T, D>(ldiag + (s * n + k) * D * D, L);
        LogAcc<T> la;
        la.init();
        bool bad = false;
        tri_inv_lower<T, D>(L, Linv, la, bad);
        (void)Li;
        T Out[D][D];
        trimulT_self_lower<T, D>(Linv, Out);        // L^-T L^-1 (lower)
        if (lsub && k + 1 < n) {
            T W[D][D], G[D][D], SG[D][D];
            load_mat<T, D, D>(lsub + (s * (n - 1) + k) * D * D, W);
            // G = W L^-1
            MF_UNROLL for (int i = 0; i < D; ++i)
                MF_UNROLL for (int j = 0; j < D; ++j) {
                    T a = T(0);
                    MF_UNROLL for (int l = j; l < D; ++l) a += W[i][l] * Linv[l][j];
                    G[i][j] = a;
                }
            // SG = Sigma_{k+1} G
            MF_UNROLL for (int i = 0; i < D; ++i)
                MF_UNROLL for (int j = 0; j < D; ++j) {
                    T a = T(0);
                    MF_UNROLL for (int l = 0; l < D; ++l) a += Sig[i][l] * G[l][j];
                    SG[i][j] = a;
                }
            if (osub) {
                T neg[D][D];
                MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) neg[i][j] = -SG[i][j];
                store_mat<T, D, D>(osub + (s * (n - 1) + k) * D * D, neg);
            }
            MF_UNROLL for (int i = 0; i < D; ++i)
                MF_UNROLL for (int j = 0; j <= i; ++j) {
                    T a = T(0);
                    MF_UNROLL for (int l = 0; l < D; ++l) a += G[l][i] * SG[l][j];
                    Out[i][j] += a;
                }
        }
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) Sig[i][j] = (i >= j) ? Out[i][j] : Out[j][i];
        store_mat<T, D, D>(odiag + (s * n + k) * D * D, Sig);
    }
}

// K5  upper_diagonal_lower (block_tri_diag.py:438-545): backward UDU^T; writes U_k^T and chol(Delta_k).
//     With eta != null it also produces the posterior chain of posterior_state_space_model
//     (kalman_filter.py:149-182): m_post, chol(Delta_k^-1).
template <typename T, int D>
__global__ void __launch_bounds__(64) btd_udl_kernel(long B, long n, const T* __restrict__ diag,
                                                     const T* __restrict__ sub, T* __restrict__ ut,
                                                     T* __restrict__ chol_d, const T* __restrict__ eta,
                                                     T* __restrict__ m_post, T* __restrict__ chol_dinv, int chain, int* info) {
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    // chain layout (see the header): block 0 of the posterior chain first for all series, then blocks 1.. as [B, n-1, ...]
    auto chain_idx = [&](long k) { return chain ? (k == 0 ? s : B + s * (n - 1) + k - 1) : s * n + k; };
    T Lp[D][D], Lpi[D], xp[D];   // chol(Delta_{k+1}), its inverse diagonal, x_{k+1}
    bool bad = false;
    // the loads of a block are issued together, one block ahead of their use where two sets fit the registers
    struct Step { T Dl[D][D]; T S[D][D]; T x[D]; };
    auto load = [&](long k, Step& d) {
        load_lower<T, D>(diag + (s * n + k) * D * D, d.Dl);
        if (eta) load_vec<T, D>(eta + (s * n + k) * D, d.x);
        if (n > 1) load_mat<T, D, D>(sub + (s * (n - 1) + (k < n - 1 ? k : n - 2)) * D * D, d.S);   // last block: clamped, unused
    };
    constexpr bool PF = sizeof(T) == 4 ? (D <= 8) : (D <= 6);
    Step cur, nxt;
    if (PF) load(n - 1, cur);
    for (long k = n - 1; k >= 0; --k) {
        if (PF) load(k > 0 ? k - 1 : 0, nxt);
        else load(k, cur);
        __builtin_amdgcn_sched_barrier(0);
        T Dl[D][D], x[D];
        MF_UNROLL for (int i = 0; i < D; ++i) { x[i] = eta ? cur.x[i] : T(0); MF_UNROLL for (int j = 0; j <= i; ++j) Dl[i][j] = cur.Dl[i][j]; }
        if (k + 1 < n) {
            T U[D][D];
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) U[i][j] = cur.S[i][j];
            trsm_left_lower<T, D, D>(Lp, Lpi, U);          // L^-1 S
            syrk_tn_lower<T, D, D>(U, Dl, T(-1));          // Delta_k = D_k - S^T Delta_{k+1}^-1 S
            trsm_left_lower_t<T, D, D>(Lp, Lpi, U);        // U_k^T = Delta_{k+1}^-1 S
            if (eta) {
                T ux[D];
                gemv_t<T, D, D>(U, xp, ux);                // U_k x_{k+1}
                MF_UNROLL for (int i = 0; i < D; ++i) x[i] -= ux[i];
            }
            if (chain) { MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) U[i][j] = -U[i][j]; }
            store_mat<T, D, D>(ut + (s * (n - 1) + k) * D * D, U);
        }
        if (PF) cur = nxt;
        LogAcc<T> la;
        la.init();
        chol_lower<T, D>(Dl, Lpi, la, bad);
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Lp[i][j] = Dl[i][j];
        if (chol_d) store_lower<T, D>(chol_d + (s * n + k) * D * D, Lp);
        if (eta) {
            MF_UNROLL for (int i = 0; i < D; ++i) xp[i] = x[i];
            trsv_lower<T, D>(Lp, Lpi, x);
            trsv_lower_t<T, D>(Lp, Lpi, x);                // m_k = Delta_k^-1 x_k
            store_vec<T, D>(m_post + chain_idx(k) * D, x);
            // chol(Delta_k^-1) = chol(L^-T L^-1)
            T Linv[D][D], Q[D][D], Qi[D];
            LogAcc<T> lb;
            lb.init();
            tri_inv_lower<T, D>(Lp, Linv, lb, bad);
            trimulT_self_lower<T, D>(Linv, Q);
            chol_lower<T, D>(Q, Qi, lb, bad);
            store_lower<T, D>(chol_dinv + chain_idx(k) * D * D, Q);
        }
    }
    if (bad && info) raise_info(info);
}

// Posterior precision + information vector, one lane per (series, block):
//   diag_k = Q_k^-1 + A_{k+1}^T Q_{k+1}^-1 A_{k+1} (+ H^T R^-1 H),  sub_k = -Q_{k+1}^-1 A_{k+1},
//   eta_k  = Q_k^-1 m_k - A_{k+1}^T Q_{k+1}^-1 m_{k+1} (+ H^T R^-1 y)
// (state_space_model.py:431-483, kalman_filter.py:86-101,153-156).  H == null gives the prior precision.
template <typename T, int D, int M>
__global__ void __launch_bounds__(256) ssm_precision_kernel(KfArgs<T> a, T* __restrict__ diag, T* __restrict__ sub,
                                                            T* __restrict__ eta) {
    // Every transition k (A_k, cholQ_k, b_k) feeds TWO blocks: Q_k^-1 into block k+1 and A_k^T Q_k^-1 A_k into block k.  The
    // lane of block k loads transition k once, keeps the second and hands the first (d(d+1)/2 + d values) to the lane of
    // block k+1; only block 0 (prior) and the first lane of a wave load "their" Cholesky factor themselves.
    const long total = a.B * a.Tn;
    const long id_raw = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;       // lanes past the end redo the last block (the cross-lane moves need them)
    const long s = id / a.Tn, k = id % a.Tn;
    const int m = a.m, lane = threadIdx.x & 63;
    LogAcc<T> la;
    la.init();
    bool bad = false;
    T Dn[D][D], rn[D];                                  // block k: pivot (lower) and information vector
    T nxtD[D][D], nxtr[D];                              // what transition k gives to block k+1: Q_k^-1, Q_k^-1 b_k
    MF_UNROLL for (int i = 0; i < D; ++i) {
        rn[i] = T(0); nxtr[i] = T(0);
        MF_UNROLL for (int j = 0; j <= i; ++j) { Dn[i][j] = T(0); nxtD[i][j] = T(0); }
    }
    if (k + 1 < a.Tn) {
        T C2[D][D], Ci2[D][D], Bm[D][D];
        load_lower<T, D>(a.cholQ + (s * (a.Tn - 1) + k) * D * D, C2);
        load_mat<T, D, D>(a.A + (s * (a.Tn - 1) + k) * D * D, Bm);
        tri_inv_lower<T, D>(C2, Ci2, la, bad);
        trimul_lower_inplace<T, D, D>(Ci2, Bm);                     // B = C^-1 A
        syrk_tn_lower<T, D, D>(Bm, Dn, T(1));                       // A^T Q^-1 A
        trimulT_self_lower<T, D>(Ci2, nxtD);                        // Q^-1
        if (eta) {
            T m2[D], w2[D], btw[D];
            load_vec<T, D>(a.b + (s * (a.Tn - 1) + k) * D, m2);
            trimul_lower_vec<T, D>(Ci2, m2, w2);
            gemv_t<T, D, D>(Bm, w2, btw);
            MF_UNROLL for (int i = 0; i < D; ++i) rn[i] = -btw[i];
            trimulT_lower_vec<T, D>(Ci2, w2, nxtr);
        }
        neg_trimulT_lower_inplace<T, D, D>(Ci2, Bm);                // -C^-T B = -Q^-1 A
        if (valid) store_mat<T, D, D>(sub + (s * (a.Tn - 1) + k) * D * D, Bm);
    }
    // own part of block k: from the lane on the left (transition k-1), or computed here
    T ownD[D][D], ownr[D];
    MF_UNROLL for (int i = 0; i < D; ++i) {
        ownr[i] = __shfl_up(nxtr[i], 1, 64);
        MF_UNROLL for (int j = 0; j <= i; ++j) ownD[i][j] = __shfl_up(nxtD[i][j], 1, 64);
    }
    if (k == 0 || lane == 0) {
        T C[D][D], Ci[D][D];
        load_lower<T, D>(k == 0 ? a.cholP0 + s * D * D : a.cholQ + (s * (a.Tn - 1) + k - 1) * D * D, C);
        tri_inv_lower<T, D>(C, Ci, la, bad);
        trimulT_self_lower<T, D>(Ci, ownD);
        MF_UNROLL for (int i = 0; i < D; ++i) ownr[i] = T(0);
        if (eta) {
            T mvec[D], w[D];
            load_vec<T, D>(k == 0 ? a.mu0 + s * D : a.b + (s * (a.Tn - 1) + k - 1) * D, mvec);
            trimul_lower_vec<T, D>(Ci, mvec, w);
            trimulT_lower_vec<T, D>(Ci, w, ownr);
        }
    }
    MF_UNROLL for (int i = 0; i < D; ++i) {
        rn[i] += ownr[i];
        MF_UNROLL for (int j = 0; j <= i; ++j) Dn[i][j] += ownD[i][j];
    }
    if (a.H) {
        const T* Ri = a.rinv_per_step ? a.Rinv + (s * a.Tn + k) * m * m : a.Rinv;
        T dummy[D];
        MF_UNROLL for (int i = 0; i < D; ++i) dummy[i] = T(0);
        if (a.y) Obs<T, D, M>::apply(a.H + (s * a.Tn + k) * m * D, a.y + (s * a.Tn + k) * m, Ri, m, Dn, rn);
        else {
            // precision only: feed zeros for y
            T zero[MF_MAXM] = {T(0), T(0), T(0), T(0)};
            Obs<T, D, M>::apply(a.H + (s * a.Tn + k) * m * D, zero, Ri, m, Dn, dummy);
        }
    }
    if (valid) {
        store_sym<T, D>(diag + id * D * D, Dn);
        if (eta) store_vec<T, D>(eta + id * D, rn);
    }
}

// Block-wise product out[s, k] = X[s, k] Y[s, k] of two [B, n, d, d] block arrays whose series may be strided
// (StateSpaceModel.subsequent_covariances, state_space_model.py:326-341: A_k P_k with P = covs[..., :-1, :, :]).
// One lane per block; a batched d x d GEMM through a BLAS library costs a launch-bound eternity at d <= 9.
template <typename T, int D>
__global__ void __launch_bounds__(256) block_matmul_kernel(long B, long n, const T* __restrict__ X, long xs,
                                                           const T* __restrict__ Y, long ys, T* __restrict__ out) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * n) return;
    const long s = id / n, k = id % n;
    T Xm[D][D], Ym[D][D], Om[D][D];
    load_mat<T, D, D>(X + (s * xs + k) * D * D, Xm);
    load_mat<T, D, D>(Y + (s * ys + k) * D * D, Ym);
    MF_UNROLL for (int i = 0; i < D; ++i) {
        MF_UNROLL for (int j = 0; j < D; ++j) Om[i][j] = Xm[i][0] * Ym[0][j];
        MF_UNROLL for (int l = 1; l < D; ++l)
            MF_UNROLL for (int j = 0; j < D; ++j) Om[i][j] += Xm[i][l] * Ym[l][j];
    }
    store_mat<T, D, D>(out + id * D * D, Om);
}

// StateSpaceModel.marginal_means (state_space_model.py:232-251): mu_{k+1} = A_k mu_k + b_k.
// rhs series r uses transitions of series r % Bl (sample() passes sample_shape + batch_shape).
template <typename T, int D>
__global__ void __launch_bounds__(64) ssm_means_kernel(long Bl, long Br, long Tn, const T* __restrict__ A,
                                                       const T* __restrict__ offs, T* __restrict__ out) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= Br) return;
    const long s = r % Bl;
    T mu[D];
    load_vec<T, D>(offs + r * Tn * D, mu);
    store_vec<T, D>(out + r * Tn * D, mu);
    for (long k = 1; k < Tn; ++k) {
        T Am[D][D], bv[D], nx[D];
        load_mat<T, D, D>(A + (s * (Tn - 1) + k - 1) * D * D, Am);
        load_vec<T, D>(offs + (r * Tn + k) * D, bv);
        gemv_n<T, D, D>(Am, mu, nx);
        MF_UNROLL for (int i = 0; i < D; ++i) mu[i] = nx[i] + bv[i];
        store_vec<T, D>(out + (r * Tn + k) * D, mu);
    }
}

// Conditional prediction of the state at new time points (conditionals.py:29-83,122-203,380-420 of the reference):
// p(x_t) = N(D mu_- + E mu_+,  T + [D E] S [D E]^T) with
//   Q-+ = Q_tp + A_tp Q_mt A_tp^T,  E = Q_mt A_tp^T Q-+^-1,  D = A_mt - E A_tp A_mt,  T = Q_mt - Q_mt A_tp^T Q-+^-1 A_tp Q_mt
// and (mu_-, mu_+, S) the pairwise posterior marginal of the two training points around t (the prior beyond the ends,
// conditionals.py:424-485).  One lane per (series, new point); idx = insertion index of t among the training points.
template <typename T, int D>
__global__ void __launch_bounds__(64) sde_predict_kernel(long B, long N, long Np, const long long* __restrict__ idx,
                                                         const T* __restrict__ Amt, const T* __restrict__ Qmt,
                                                         const T* __restrict__ Atp, const T* __restrict__ Qtp,
                                                         const T* __restrict__ means, const T* __restrict__ covs,
                                                         const T* __restrict__ subseq, const T* __restrict__ m0,
                                                         const T* __restrict__ P0, T* __restrict__ omean,
                                                         T* __restrict__ ocov, int* info) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * Np) return;
    const long s = id / Np;
    const long i = (long)idx[id];                        // 0 .. N
    T Am[D][D], Qm[D][D], Ap[D][D], Qp[D][D];
    load_mat<T, D, D>(Amt + id * D * D, Am);
    load_mat<T, D, D>(Qmt + id * D * D, Qm);
    load_mat<T, D, D>(Atp + id * D * D, Ap);
    load_lower<T, D>(Qtp + id * D * D, Qp);
    // G = A_tp Q_mt ;  Q-+ (lower) = Q_tp + G A_tp^T
    T G[D][D];
    MF_UNROLL for (int r = 0; r < D; ++r)
        MF_UNROLL for (int c = 0; c < D; ++c) {
            T a = T(0);
            MF_UNROLL for (int l = 0; l < D; ++l) a += Ap[r][l] * Qm[l][c];
            G[r][c] = a;
        }
    MF_UNROLL for (int r = 0; r < D; ++r)
        MF_UNROLL for (int c = 0; c <= r; ++c) {
            T a = Qp[r][c];
            MF_UNROLL for (int l = 0; l < D; ++l) a += G[r][l] * Ap[c][l];
            Qp[r][c] = a;
        }
    T Li[D];
    LogAcc<T> la;
    la.init();
    bool bad = false;
    chol_lower<T, D>(Qp, Li, la, bad);                   // L
    trsm_left_lower<T, D, D>(Qp, Li, G);                 // V = L^-1 A_tp Q_mt
    T Tm[D][D];
    MF_UNROLL for (int r = 0; r < D; ++r) MF_UNROLL for (int c = 0; c <= r; ++c) Tm[r][c] = Qm[r][c];
    syrk_tn_lower<T, D, D>(G, Tm, T(-1));                // T = Q_mt - V^T V   (lower)
    trsm_left_lower_t<T, D, D>(Qp, Li, G);               // E^T = L^-T V
    // D = A_mt - E A_tp A_mt  with E[r][c] = G[c][r]
    T EA[D][D], Dm[D][D];
    MF_UNROLL for (int r = 0; r < D; ++r)
        MF_UNROLL for (int c = 0; c < D; ++c) {
            T a = T(0);
            MF_UNROLL for (int l = 0; l < D; ++l) a += G[l][r] * Ap[l][c];
            EA[r][c] = a;
        }
    MF_UNROLL for (int r = 0; r < D; ++r)
        MF_UNROLL for (int c = 0; c < D; ++c) {
            T a = Am[r][c];
            MF_UNROLL for (int l = 0; l < D; ++l) a -= EA[r][l] * Am[l][c];
            Dm[r][c] = a;
        }
    // pairwise marginal around the point
    T mu_m[D], mu_p[D];
    const bool has_m = i > 0, has_p = i < N;
    load_vec<T, D>(has_m ? means + (s * N + i - 1) * D : m0 + s * D, mu_m);
    load_vec<T, D>(has_p ? means + (s * N + i) * D : m0 + s * D, mu_p);
    T mean[D];
    MF_UNROLL for (int r = 0; r < D; ++r) {
        T a = T(0);
        MF_UNROLL for (int l = 0; l < D; ++l) a += Dm[r][l] * mu_m[l] + G[l][r] * mu_p[l];
        mean[r] = a;
    }
    store_vec<T, D>(omean + id * D, mean);
    if (ocov) {
        // cov = T + D P- D^T + E P+ E^T + E C D^T + (E C D^T)^T,  C = Cov(x+, x-)
        T Pm[D][D], Pp[D][D], X1[D][D], X2[D][D], Out[D][D];
        load_mat<T, D, D>(has_m ? covs + (s * N + i - 1) * D * D : P0 + s * D * D, Pm);
        load_mat<T, D, D>(has_p ? covs + (s * N + i) * D * D : P0 + s * D * D, Pp);
        MF_UNROLL for (int r = 0; r < D; ++r)
            MF_UNROLL for (int c = 0; c < D; ++c) {
                T a = T(0), b = T(0);
                MF_UNROLL for (int l = 0; l < D; ++l) { a += Dm[r][l] * Pm[l][c]; b += G[l][r] * Pp[l][c]; }
                X1[r][c] = a;                            // D P-
                X2[r][c] = b;                            // E P+
            }
        if (has_m && has_p) {
            T C[D][D];
            load_mat<T, D, D>(subseq + (s * (N - 1) + i - 1) * D * D, C);
            // X1 += E C   (so that X1 D^T collects D P- D^T + E C D^T), X2 += D C^T
            MF_UNROLL for (int r = 0; r < D; ++r)
                MF_UNROLL for (int c = 0; c < D; ++c) {
                    T a = T(0), b = T(0);
                    MF_UNROLL for (int l = 0; l < D; ++l) { a += G[l][r] * C[l][c]; b += Dm[r][l] * C[c][l]; }
                    X1[r][c] += a;
                    X2[r][c] += b;
                }
        }
        MF_UNROLL for (int r = 0; r < D; ++r)
            MF_UNROLL for (int c = 0; c < D; ++c) {
                T a = (r >= c) ? Tm[r][c] : Tm[c][r];
                MF_UNROLL for (int l = 0; l < D; ++l) a += X1[r][l] * Dm[c][l] + X2[r][l] * G[l][c];
                Out[r][c] = a;
            }
        store_mat<T, D, D>(ocov + id * D * D, Out);
    }
    if (bad && info) raise_info(info);
}

// conditional_statistics (markovflow/conditionals.py:87-203) alone: the statistics of p(x_t | x_-, x_+) = N(P_t [x_-, x_+], T_t) for
// every new point from the transitions x_- -> x_t (A_mt, Q_mt) and x_t -> x_+ (A_tp, Q_tp) - the first half of sde_predict_kernel
// with P_t = [D | E] (d x 2d) and T_t (d x d, symmetric) written out.  One lane per point.
template <typename T, int D>
__global__ void __launch_bounds__(64) sde_cond_stats_kernel(long n, const T* __restrict__ Amt, const T* __restrict__ Qmt,
                                                            const T* __restrict__ Atp, const T* __restrict__ Qtp,
                                                            T* __restrict__ proj, T* __restrict__ cov, int* info) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    T Am[D][D], Qm[D][D], Ap[D][D], Qp[D][D];
    load_mat<T, D, D>(Amt + id * D * D, Am);
    load_mat<T, D, D>(Qmt + id * D * D, Qm);
    load_mat<T, D, D>(Atp + id * D * D, Ap);
    load_lower<T, D>(Qtp + id * D * D, Qp);
    T G[D][D];
    MF_UNROLL for (int r = 0; r < D; ++r)
        MF_UNROLL for (int c = 0; c < D; ++c) {
            T a = T(0);
            MF_UNROLL for (int l = 0; l < D; ++l) a += Ap[r][l] * Qm[l][c];
            G[r][c] = a;                                 // A_tp Q_mt
        }
    MF_UNROLL for (int r = 0; r < D; ++r)
        MF_UNROLL for (int c = 0; c <= r; ++c) {
            T a = Qp[r][c];
            MF_UNROLL for (int l = 0; l < D; ++l) a += G[r][l] * Ap[c][l];
            Qp[r][c] = a;                                // Q-+ (lower)
        }
    T Li[D];
    LogAcc<T> la;
    la.init();
    bool bad = false;
    chol_lower<T, D>(Qp, Li, la, bad);
    trsm_left_lower<T, D, D>(Qp, Li, G);                 // V = L^-1 A_tp Q_mt
    T Tm[D][D];
    MF_UNROLL for (int r = 0; r < D; ++r) MF_UNROLL for (int c = 0; c <= r; ++c) Tm[r][c] = Qm[r][c];
    syrk_tn_lower<T, D, D>(G, Tm, T(-1));                // T = Q_mt - V^T V (lower)
    trsm_left_lower_t<T, D, D>(Qp, Li, G);               // E^T = L^-T V
    T* pr = proj + id * D * 2 * D;
    MF_UNROLL for (int r = 0; r < D; ++r) {
        T ea[D];                                         // row r of E A_tp
        MF_UNROLL for (int c = 0; c < D; ++c) {
            T a = T(0);
            MF_UNROLL for (int l = 0; l < D; ++l) a += G[l][r] * Ap[l][c];
            ea[c] = a;
        }
        MF_UNROLL for (int c = 0; c < D; ++c) {
            T a = Am[r][c];
            MF_UNROLL for (int l = 0; l < D; ++l) a -= ea[l] * Am[l][c];
            pr[r * 2 * D + c] = a;                       // D = A_mt - E A_tp A_mt
            pr[r * 2 * D + D + c] = G[c][r];             // E
        }
    }
    store_sym<T, D>(cov + id * D * D, Tm);
    if (bad && info) raise_info(info);
}

// Gradient of KalmanFilter.log_likelihood with respect to every tensor of the model, by Fisher's
// identity: grad log p(y) = E_{x|y}[grad log p(x, y)], evaluated from the SMOOTHED pairwise marginals (means m_k, covariances
// S_k, cross-covariances S_{k+1,k} = Cov(x_{k+1}, x_k)) - exact, and local in time: one lane per (series, time point).
//   e_k = x_{k+1} - A_k x_k - b_k:  E[e] = m_{k+1} - A m_k - b,  Psi = E[e e^T] = E[e]E[e]^T + S_{k+1} - A S_{k+1,k}^T - S_{k+1,k} A^T + A S_k A^T
//   d/dA_k = Q^-1 (E[e] m_k^T + S_{k+1,k} - A S_k),  d/db_k = Q^-1 E[e],  d/dcholQ_k = tril(C^-T (C^-1 Psi C^-T - I)),
//   d/dmu0 = P0^-1 (m_0 - mu0),  d/dcholP0 likewise with Psi_0 = (m_0 - mu0)(m_0 - mu0)^T + S_0,
//   r_k = y_k - H_k x_k:  d/dH_k = R^-1 (E[r] m_k^T - H S_k),  d/dy_k = -R^-1 E[r],
//   per-point contribution to d/dR^-1-side: Omega_k = E[r]E[r]^T + H S_k H^T  (reduced over time by the caller).
// The reference obtains these through TensorFlow's reverse mode over the banded ops (banded_matrices registers gradients).
template <typename T, int D, int M>
__global__ void __launch_bounds__(64) kf_grad_kernel(KfArgs<T> a, const T* __restrict__ pm, const T* __restrict__ pS,
                                                     const T* __restrict__ pX, T* __restrict__ gmu0, T* __restrict__ gC0,
                                                     T* __restrict__ gA, T* __restrict__ gb, T* __restrict__ gC,
                                                     T* __restrict__ gH, T* __restrict__ gy, T* __restrict__ gOm) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= a.B * a.Tn) return;
    const long s = id / a.Tn, k = id % a.Tn;
    const int m = a.m;
    const T wgt = a.weights ? a.weights[s] : T(1);      // the incoming gradient of this series' value
    constexpr int MM = (M > 0) ? M : MF_MAXM;
    T mk[D], Sk[D][D];
    load_vec<T, D>(pm + id * D, mk);
    load_mat<T, D, D>(pS + id * D * D, Sk);
    LogAcc<T> la;
    la.init();
    bool bad = false;
    // ---- observation terms of time point k (skipped without an emission model: the score of a bare chain) ----------------
    if (a.H != nullptr) {
        const T* __restrict__ Rv = a.Rinv + (a.rinv_per_step ? id * m * m : 0);      // shared [m,m] or per step [B,T,m,m]
        T h[MM][D], r[MM], Rr[MM], HS[MM][D];
        MF_UNROLL for (int o = 0; o < MM; ++o) {
            const bool on = (M > 0) || (o < m);
            T acc = on ? a.y[id * m + o] : T(0);
            MF_UNROLL for (int i = 0; i < D; ++i) { h[o][i] = on ? a.H[(id * m + o) * D + i] : T(0); acc -= h[o][i] * mk[i]; }
            r[o] = acc;                                            // E[r]
        }
        MF_UNROLL for (int o = 0; o < MM; ++o)
            MF_UNROLL for (int i = 0; i < D; ++i) {
                T acc = T(0);
                MF_UNROLL for (int l = 0; l < D; ++l) acc += h[o][l] * Sk[l][i];
                HS[o][i] = acc;
            }
        MF_UNROLL for (int o = 0; o < MM; ++o) {
            T acc = T(0);
            MF_UNROLL for (int p = 0; p < MM; ++p) {
                const bool on = (M > 0) || (o < m && p < m);
                acc += (on ? Rv[o * ((M > 0) ? M : m) + p] : T(0)) * r[p];
            }
            Rr[o] = acc;
        }
        MF_UNROLL for (int o = 0; o < MM; ++o) {
            if (!((M > 0) || (o < m))) continue;
            gy[id * m + o] = -wgt * Rr[o];
            // dH = R^-1 (r m^T - H S)
            MF_UNROLL for (int i = 0; i < D; ++i) {
                T acc = Rr[o] * mk[i];
                MF_UNROLL for (int p = 0; p < MM; ++p) {
                    const bool on = (M > 0) || (p < m);
                    acc -= (on ? Rv[o * ((M > 0) ? M : m) + p] : T(0)) * HS[p][i];
                }
                gH[(id * m + o) * D + i] = wgt * acc;
            }
            // Omega = r r^T + H S H^T
            MF_UNROLL for (int p = 0; p < MM; ++p) {
                if (!((M > 0) || (p < m))) continue;
                T acc = r[o] * r[p];
                MF_UNROLL for (int i = 0; i < D; ++i) acc += HS[o][i] * h[p][i];
                gOm[(id * m + o) * m + p] = wgt * acc;
            }
        }
    }
    // ---- prior of the first state -----------------------------------------------------------------------------------------
    if (k == 0) {
        T C[D][D], Ci[D][D], dv[D], u[D], Psi[D][D];
        load_lower<T, D>(a.cholP0 + s * D * D, C);
        tri_inv_lower<T, D>(C, Ci, la, bad);
        MF_UNROLL for (int i = 0; i < D; ++i) dv[i] = mk[i] - a.mu0[s * D + i];
        trimul_lower_vec<T, D>(Ci, dv, u);                         // C^-1 (m0 - mu0)
        T g[D];
        trimulT_lower_vec<T, D>(Ci, u, g);                         // P0^-1 (m0 - mu0)
        MF_UNROLL for (int i = 0; i < D; ++i) g[i] *= wgt;
        store_vec<T, D>(gmu0 + s * D, g);
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Psi[i][j] = dv[i] * dv[j] + Sk[i][j];
        // N = C^-1 Psi C^-T - I ;  dC = tril(C^-T N)
        T N1[D][D], N[D][D], G[D][D];
        trimul_lower<T, D, D>(Ci, Psi, N1);
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T acc = (i == j) ? T(-1) : T(0);
                MF_UNROLL for (int l = 0; l <= j; ++l) acc += N1[i][l] * Ci[j][l];
                N[i][j] = acc;
            }
        trimulT_lower<T, D, D>(Ci, N, G);
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) G[i][j] *= wgt;
        store_lower<T, D>(gC0 + s * D * D, G);
    }
    // ---- transition k -> k+1 ----------------------------------------------------------------------------------------------
    if (k + 1 < a.Tn) {
        const long tid = s * (a.Tn - 1) + k;
        T mn[D], Sn[D][D], X[D][D], Am[D][D], C[D][D], Ci[D][D];
        load_vec<T, D>(pm + (id + 1) * D, mn);
        load_mat<T, D, D>(pS + (id + 1) * D * D, Sn);
        load_mat<T, D, D>(pX + tid * D * D, X);                    // Cov(x_{k+1}, x_k)
        load_mat<T, D, D>(a.A + tid * D * D, Am);
        load_lower<T, D>(a.cholQ + tid * D * D, C);
        tri_inv_lower<T, D>(C, Ci, la, bad);
        T eb[D];
        MF_UNROLL for (int i = 0; i < D; ++i) {
            T acc = mn[i] - a.b[tid * D + i];
            MF_UNROLL for (int l = 0; l < D; ++l) acc -= Am[i][l] * mk[l];
            eb[i] = acc;
        }
        // E[e x_k^T] = eb m_k^T + X - A S_k ;  A S_k kept for Psi
        T AS[D][D], EX[D][D];
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T acc = T(0);
                MF_UNROLL for (int l = 0; l < D; ++l) acc += Am[i][l] * Sk[l][j];
                AS[i][j] = acc;
                EX[i][j] = eb[i] * mk[j] + X[i][j] - acc;
            }
        // dA = Q^-1 E[e x^T] = C^-T (C^-1 EX) ; db = Q^-1 eb
        T t1[D][D], dA[D][D], u[D], db[D];
        trimul_lower<T, D, D>(Ci, EX, t1);
        trimulT_lower<T, D, D>(Ci, t1, dA);
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) dA[i][j] *= wgt;
        store_mat<T, D, D>(gA + tid * D * D, dA);
        trimul_lower_vec<T, D>(Ci, eb, u);
        trimulT_lower_vec<T, D>(Ci, u, db);
        MF_UNROLL for (int i = 0; i < D; ++i) db[i] *= wgt;
        store_vec<T, D>(gb + tid * D, db);
        // Psi = eb eb^T + S_{k+1} - A X^T - X A^T + A S_k A^T
        T Psi[D][D];
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T acc = eb[i] * eb[j] + Sn[i][j];
                MF_UNROLL for (int l = 0; l < D; ++l) acc += AS[i][l] * Am[j][l] - Am[i][l] * X[j][l] - X[i][l] * Am[j][l];
                Psi[i][j] = acc;
            }
        T N1[D][D], N[D][D], G[D][D];
        trimul_lower<T, D, D>(Ci, Psi, N1);
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T acc = (i == j) ? T(-1) : T(0);
                MF_UNROLL for (int l = 0; l <= j; ++l) acc += N1[i][l] * Ci[j][l];
                N[i][j] = acc;
            }
        trimulT_lower<T, D, D>(Ci, N, G);
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) G[i][j] *= wgt;
        store_lower<T, D>(gC + tid * D * D, G);
    }
    if (bad && a.info) raise_info(a.info);
}

}  // namespace mf
