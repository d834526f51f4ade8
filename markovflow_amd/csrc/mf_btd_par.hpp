// Parallel-in-time forms of SymmetricBlockTriDiagonal.cholesky and LowerTriangularBlockTriDiagonal.solve
// (block_tri_diag.py:423-436, :339-351) for FEW, LONG chains (BASELINE config 3: B=1, T=100000, d=6).
//
// The API contract is the NATURAL-ORDER factor (tests/unit/test_block_tri_diag.py:104-107 of the reference), which a
// plain odd-even cyclic reduction does not produce.  What is used instead is a multi-level partitioned
// elimination that keeps natural-order information:
//
//   up-sweep    every chunk of `len` consecutive blocks eliminates its interior in natural order (carrying the
//               fill-in towards the block on its left, exactly as the log-likelihood reduction does) and leaves
//               its LAST block as one block of the next, `len` times shorter, level.  Contributions that come
//               from blocks located AFTER a block in natural order (the interior of the next chunk) are kept
//               apart from it ("future" part: Gf + GU) instead of being folded into its pivot.
//   down-sweep  the natural-order pivot Sigma_j = L_j L_j^T of a block is the Schur complement of everything
//               BEFORE it, and it obeys, on every level,
//                   Sigma_j = Dv_j - F_j (Sigma_{j-1} + future_{j-1})^-1 F_j^T ,
//               so from the pivots at the chunk boundaries (known from the coarser level) every chunk recovers the
//               pivots of its own blocks independently; at level 0 the "future" parts vanish and the recursion
//               is the textbook one, which emits L_k and W_k.
//
// Sequential depth: 2 * len * (number of levels) block steps instead of T; every level is one launch of
// independent lanes (one lane = one chunk, register resident, mf_small.hpp).  The solve is the same idea on
// the affine recursion z_k = M_k z_{k-1} + c_k,  M_k = -L_k^-1 W_{k-1},  c_k = L_k^-1 r_k.
#pragma once
#include "mf_small.hpp"
#include "mf_kf_x.hpp"

namespace mf {

// One level of the factorisation hierarchy, n blocks per series.
template <typename T> struct ParLevel {
    const T* Dv;   // [B, n, D, D]  pivot part that stems from the block itself and from blocks before it
    const T* Gf;   // [B, n, D, D]  future part inherited from the levels below (null on level 0)
    const T* GU;   // [B, n, D, D]  GU[j]: what eliminating the interior of (lower-level) chunk j adds to block j-1
    const T* F;    // coupling of block j with block j-1 (block index j + f_off, f_stride blocks per series)
    long n, f_stride, f_off;
    int rev;       // level 0 only: positions run backwards over the user's blocks (p -> block n-1-p, coupling transposed):
                   // the natural-order pivots of the REVERSED matrix are the Delta_k of the U D U^T factorisation
};

template <typename T, int D> MF_DEV void par_load_dv(const ParLevel<T>& lv, long s, long p, T (&Dn)[D][D]) {
    const long k = lv.rev ? lv.n - 1 - p : p;
    load_lower<T, D>(lv.Dv + (s * lv.n + k) * D * D, Dn);
}
// coupling of position p with position p-1 (rows: p)
template <typename T, int D> MF_DEV void par_load_f(const ParLevel<T>& lv, long s, long p, T (&W)[D][D]) {
    if (!lv.rev) {
        load_mat<T, D, D>(lv.F + (s * lv.f_stride + p + lv.f_off) * D * D, W);
    } else {
        T St[D][D];
        load_mat<T, D, D>(lv.F + (s * (lv.n - 1) + lv.n - 1 - p) * D * D, St);   // sub[k], k = n-1-p: rows k+1, cols k
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) W[i][j] = St[j][i];
    }
}

template <typename T, int D> MF_DEV void load_sym_lower(const T* __restrict__ p, T (&m)[D][D]) { load_lower<T, D>(p, m); }

// future part of block j on a level: Gf[j] + GU[j+1]   (lower triangle), returns false if identically zero
template <typename T, int D> MF_DEV bool par_future(const ParLevel<T>& lv, long s, long j, T (&fut)[D][D]) {
    bool any = false;
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int jj = 0; jj <= i; ++jj) fut[i][jj] = T(0);
    if (lv.Gf) {
        T g[D][D];
        load_lower<T, D>(lv.Gf + (s * lv.n + j) * D * D, g);
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int jj = 0; jj <= i; ++jj) fut[i][jj] += g[i][jj];
        any = true;
    }
    if (lv.GU && j + 1 < lv.n) {
        T g[D][D];
        load_lower<T, D>(lv.GU + (s * lv.n + j + 1) * D * D, g);
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int jj = 0; jj <= i; ++jj) fut[i][jj] += g[i][jj];
        any = true;
    }
    return any;
}

// ---- Cholesky: up-sweep -----------------------------------------------------------------------------------------
// chunk c of series s covers blocks [c len, min(n, (c+1) len)) of `in` and becomes block c of the next level:
//   oDv = pivot of its last block after the chunk's interior is gone (future part excluded), oGf = that block's
//   future part, oGU = what the interior adds to the block left of the chunk, oF = coupling last block <-> that block.
// Loads of one block step, issued together, branch-free (a load under a divergent branch is followed by a wait for its
// merge) and, where the registers allow, one step AHEAD of their use: with one chain per lane nothing else hides the
// ~2 us of a dependent global load, and the plain form of these loops paid three to four of them per block step - pivot
// part, the two future parts, the coupling - because each was loaded where it was used.
// MODE 0: level 0 (the user's blocks, no future parts); 1: level 0 of the block-reversed matrix; 2: a reduced level.
template <typename T, int D> struct ParStepData {
    T Dn[D][D];   // lower
    T g1[D][D];   // lower: Gf part of the future term (MODE 2)
    T g2[D][D];   // lower: GU part of the future term (MODE 2), valid if has2
    T W[D][D];    // coupling with the previous block (garbage for block 0)
    bool has2;
};
// one step ahead: fp32 up to d = 7, fp64 up to d = 4 (beyond, two sets of step data do not fit the register file)
#ifndef MF_PAR_PF
#define MF_PAR_PF 1
#endif
template <typename T, int D> constexpr bool par_prefetch() { return MF_PAR_PF == 2 || (MF_PAR_PF && ((sizeof(T) == 4 && D <= 7) || (sizeof(T) == 8 && D <= 4))); }

template <typename T, int D, int MODE> MF_DEV void par_load_pivot_and_coupling(const ParLevel<T>& in, long s, long k, ParStepData<T, D>& d) {
    const long kc = k > 0 ? k : 1;                     // block 0 has no coupling: load block 1's, never used
    if (MODE == 1) {
        load_lower<T, D>(in.Dv + (s * in.n + (in.n - 1 - k)) * D * D, d.Dn);
        T St[D][D];
        load_mat<T, D, D>(in.F + (s * (in.n - 1) + in.n - 1 - kc) * D * D, St);   // sub[n-1-p]: rows k+1, cols k -> transposed
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) d.W[i][j] = St[j][i];
    } else {
        load_lower<T, D>(in.Dv + (s * in.n + k) * D * D, d.Dn);
        load_mat<T, D, D>(in.F + (s * in.f_stride + kc + in.f_off) * D * D, d.W);
    }
}
// step data of block k for the up-sweep (future parts of block k itself)
template <typename T, int D, int MODE> MF_DEV void par_up_load(const ParLevel<T>& in, long s, long k, ParStepData<T, D>& d) {
    par_load_pivot_and_coupling<T, D, MODE>(in, s, k, d);
    if (MODE == 2) {
        load_lower<T, D>(in.Gf + (s * in.n + k) * D * D, d.g1);
        d.has2 = k + 1 < in.n;
        load_lower<T, D>(in.GU + (s * in.n + (d.has2 ? k + 1 : k)) * D * D, d.g2);
    }
}

template <typename T, int D, int MODE>
__global__ void __launch_bounds__(64) par_chol_up_kernel(ParLevel<T> in, long B, long len, long P, T* __restrict__ oDv,
                                                         T* __restrict__ oGf, T* __restrict__ oGU, T* __restrict__ oF,
                                                         int* info) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * P) return;
    const long s = id / P, c = id % P;
    const long k0 = c * len;
    long k1 = k0 + len;
    if (k1 > in.n) k1 = in.n;
    T Phi[D][D], Li[D], X[D][D], GU[D][D];
    MF_UNROLL for (int i = 0; i < D; ++i) {
        Li[i] = T(0);
        MF_UNROLL for (int j = 0; j < D; ++j) { Phi[i][j] = T(0); X[i][j] = T(0); GU[i][j] = T(0); }
    }
    LogAcc<T> la;
    la.init();
    bool bad = false;
    constexpr bool PF = par_prefetch<T, D>();
    ParStepData<T, D> cur, nxt;
    if (PF && k0 < k1) par_up_load<T, D, MODE>(in, s, k0, cur);
    for (long k = k0; k < k1; ++k) {
        if (PF) par_up_load<T, D, MODE>(in, s, k + 1 < k1 ? k + 1 : k, nxt);
        else par_up_load<T, D, MODE>(in, s, k, cur);            // all of this step's loads together, before its arithmetic
        __builtin_amdgcn_sched_barrier(0);
        const bool last = (k + 1 == k1);
        if (MODE == 2) {
            MF_UNROLL for (int i = 0; i < D; ++i)
                MF_UNROLL for (int j = 0; j <= i; ++j) cur.g1[i][j] += cur.has2 ? cur.g2[i][j] : T(0);   // future part
        } else {
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) cur.g1[i][j] = T(0);
        }
        if (last) {
            // the future part of the chunk's last block travels separately
            store_sym<T, D>(oGf + id * D * D, cur.g1);
        } else if (MODE == 2) {
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) cur.Dn[i][j] += cur.g1[i][j];
        }
        if (k == k0) {
            if (k > 0) { MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) X[i][j] = cur.W[i][j]; }
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Phi[i][j] = cur.Dn[i][j];
        } else {
            // eliminate block k-1 (pivot complete in Phi): factor, spike towards the block left of the chunk
            chol_lower<T, D>(Phi, Li, la, bad);
            la.init();
            if (k0 > 0) {
                trsm_left_lower<T, D, D>(Phi, Li, X);            // V = L^-1 X
                syrk_tn_lower<T, D, D>(X, GU, T(-1));            // GU -= V^T V
            }
            trsm_right_lower_t<T, D, D>(Phi, Li, cur.W);         // W = F_k L^-T
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Phi[i][j] = cur.Dn[i][j];
            syrk_nt_lower<T, D, D>(cur.W, Phi, T(-1));           // next pivot: Dn - W W^T
            if (k0 > 0) neg_mul_inplace<T, D>(cur.W, X);         // next coupling to the left block: -W V
        }
        if (PF) cur = nxt;
    }
    store_sym<T, D>(oDv + id * D * D, Phi);
    store_sym<T, D>(oGU + id * D * D, GU);
    store_mat<T, D, D>(oF + id * D * D, X);
    if (bad && info) raise_info(info);
}

// The same up-sweep with the spike (X, GU) in LDS - state dimensions whose chunk state does not fit a lane's registers
// (mf_kf_x.hpp: d >= 7 in fp64, d = 9 in fp32; par_chol_up_kernel<double, 9> spilt 1.4 KB per lane).
template <typename T, int D>
__global__ void __launch_bounds__(64) par_chol_up_x_kernel(ParLevel<T> in, long B, long len, long P, T* __restrict__ oDv,
                                                           T* __restrict__ oGf, T* __restrict__ oGU, T* __restrict__ oF,
                                                           int* info) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x;
    const long total = B * P;
    const long id_raw = (long)blockIdx.x * 64 + lane;
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;
    const long s = id / P, c = id % P;
    const long k0 = c * len;
    long k1 = k0 + len;
    if (k1 > in.n) k1 = in.n;
    if (!valid) k1 = k0;
    ElimX<T, D> E;
    E.init(reinterpret_cast<T*>(smem_raw), lane);
    for (long k = k0; k < k1; ++k) {
        const bool last = (k + 1 == k1);
        T Dn[D][D];
        {
            T fut[D][D];
            par_load_dv<T, D>(in, s, k, Dn);
            par_future<T, D>(in, s, k, fut);
            if (last) {
                store_sym<T, D>(oGf + id * D * D, fut);
            } else {
                MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Dn[i][j] += fut[i][j];
            }
        }
        if (k == k0) {
            if (k > 0) {
                T X0[D][D];
                par_load_f<T, D>(in, s, k, X0);
                MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) E.sp.setX(i, j, X0[i][j]);
            }
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) E.Phi[i][j] = Dn[i][j];
            continue;
        }
        chol_lower<T, D>(E.Phi, E.Li, E.laL, E.bad);
        E.laL.init();
        if (k0 > 0) E.eliminate_spike();                      // V = L^-1 X, GU -= V^T V   (t = 0: gU stays 0)
        T W[D][D];
        par_load_f<T, D>(in, s, k, W);
        trsm_right_lower_t<T, D, D>(E.Phi, E.Li, W);          // W = F_k L^-T
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) E.Phi[i][j] = Dn[i][j];
        syrk_nt_lower<T, D, D>(W, E.Phi, T(-1));
        if (k0 > 0) E.propagate_spike(W);
    }
    if (valid) {
        store_sym<T, D>(oDv + id * D * D, E.Phi);
        T* gu = oGU + id * D * D;
        T* f = oF + id * D * D;
        MF_UNROLL for (int i = 0; i < D; ++i) {
            MF_UNROLL for (int j = 0; j <= i; ++j) { const T v = E.sp.G(i, j); gu[i * D + j] = v; gu[j * D + i] = v; }
            MF_UNROLL for (int j = 0; j < D; ++j) f[i * D + j] = E.sp.X(i, j);
        }
        if (E.bad && info) raise_info(info);
    }
}

// ---- Cholesky: down-sweep on a level >= 1 (also the serial walk of the coarsest level: len >= n, up = null) --------
// writes the natural-order pivots Pn[j] of every block of `lv`; `up` holds the pivots of the next coarser level,
// whose block c is the last block of chunk c here.
// step data of block k for the down-sweep (always a reduced level): its own pivot part, the future parts of block k-1
// (Gf[k-1] + GU[k]), the coupling; for block 0 the loads are clamped to valid addresses and not used
template <typename T, int D> MF_DEV void par_down_load(const ParLevel<T>& lv, long s, long k, ParStepData<T, D>& d) {
    par_load_pivot_and_coupling<T, D, 2>(lv, s, k, d);
    const long kc = k > 0 ? k : 1;
    load_lower<T, D>(lv.Gf + (s * lv.n + kc - 1) * D * D, d.g1);
    load_lower<T, D>(lv.GU + (s * lv.n + (kc < lv.n ? kc : 0)) * D * D, d.g2);
}

template <typename T, int D>
__global__ void __launch_bounds__(64) par_chol_down_kernel(ParLevel<T> lv, long B, long len, long P,
                                                           const T* __restrict__ up, T* __restrict__ Pn, int* info) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * P) return;
    const long s = id / P, c = id % P;
    const long k0 = c * len;
    long k1 = k0 + len;
    if (k1 > lv.n) k1 = lv.n;
    T Sig[D][D], Li[D];
    LogAcc<T> la;
    la.init();
    bool bad = false;
    constexpr bool PF = par_prefetch<T, D>();
    ParStepData<T, D> cur, nxt;
    if (PF && k0 < k1) par_down_load<T, D>(lv, s, k0, cur);
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Sig[i][j] = T(0);
    if (c > 0) load_lower<T, D>(up + (s * P + c - 1) * D * D, Sig);
    for (long k = k0; k < k1; ++k) {
        if (PF) par_down_load<T, D>(lv, s, k + 1 < k1 ? k + 1 : k, nxt);
        else par_down_load<T, D>(lv, s, k, cur);
        __builtin_amdgcn_sched_barrier(0);
        if (k > 0) {
            // pivot of block k-1 at the moment block k is reached: natural pivot + its future part
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Sig[i][j] += cur.g1[i][j] + cur.g2[i][j];
            chol_lower<T, D>(Sig, Li, la, bad);
            la.init();
            trsm_right_lower_t<T, D, D>(Sig, Li, cur.W);
            syrk_nt_lower<T, D, D>(cur.W, cur.Dn, T(-1));
        }
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Sig[i][j] = cur.Dn[i][j];
        store_sym<T, D>(Pn + (s * lv.n + k) * D * D, Sig);
        if (PF) cur = nxt;
    }
    if (bad && info) raise_info(info);
}

// ---- Cholesky: level 0, emits the factor ----------------------------------------------------------------------
// chunk c restarts the natural-order recursion from the pivot of block c len - 1 (`up`, block c-1 of level 1).
template <typename T, int D>
__global__ void __launch_bounds__(64) par_chol_emit_kernel(long B, long n, long len, long P, const T* __restrict__ diag,
                                                           const T* __restrict__ sub, const T* __restrict__ up,
                                                           T* __restrict__ ldiag, T* __restrict__ lsub, int* info) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * P) return;
    const long s = id / P, c = id % P;
    const long k0 = c * len;
    long k1 = k0 + len;
    if (k1 > n) k1 = n;
    T L[D][D], Li[D];
    LogAcc<T> la;
    la.init();
    bool bad = false;
    long first_bad = -1;
    struct Step { T S[D][D]; T W[D][D]; };
    auto load = [&](long k, Step& d) {
        load_lower<T, D>(diag + (s * n + k) * D * D, d.S);
        load_mat<T, D, D>(sub + (s * (n - 1) + (k > 0 ? k - 1 : 0)) * D * D, d.W);     // block 0: loaded, not used
    };
    // only two matrices per step here: loads run one step ahead.  Two steps ahead (MF_CHOL_PF2) measured SLOWER for the
    // factorisation (2.05 -> 2.43 ms at B = 16384, T = 500, d = 6 fp64) and neutral for the solve: both off.
#ifndef MF_CHOL_PF2
#define MF_CHOL_PF2 0
#endif
    constexpr bool PF2 = MF_CHOL_PF2 && ((sizeof(T) == 8 && D <= 6) || (sizeof(T) == 4 && D <= 8));
    constexpr bool PF = par_prefetch<T, D>() || (sizeof(T) == 8 && D <= 6) || PF2;
    Step cur, nxt, nx2;
    if (PF && k0 < k1) load(k0, cur);
    if (PF2 && k0 < k1) load(k0 + 1 < k1 ? k0 + 1 : k0, nxt);
    if (c > 0) {
        load_lower<T, D>(up + (s * P + c - 1) * D * D, L);
        chol_lower<T, D>(L, Li, la, bad);
        la.init();
    }
    for (long k = k0; k < k1; ++k) {
        if (PF2) load(k + 2 < k1 ? k + 2 : k1 - 1, nx2);
        else if (PF) load(k + 1 < k1 ? k + 1 : k, nxt);
        else load(k, cur);
        __builtin_amdgcn_sched_barrier(0);
        if (k > 0) {
            trsm_right_lower_t<T, D, D>(L, Li, cur.W);
            store_mat<T, D, D>(lsub + (s * (n - 1) + k - 1) * D * D, cur.W);
            syrk_nt_lower<T, D, D>(cur.W, cur.S, T(-1));
        }
        chol_lower<T, D>(cur.S, Li, la, bad);
        if (bad && first_bad < 0) first_bad = k;
        la.init();
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) L[i][j] = cur.S[i][j];
        store_lower<T, D>(ldiag + (s * n + k) * D * D, L);
        if (PF) cur = nxt;
        if (PF2) nxt = nx2;
    }
    if (bad && info) raise_pivot(info, s * n + (first_bad < 0 ? k0 : first_bad));
}

// ---- Solve: affine recursion z_p = M_p z_{p-1} + c_p over positions p (p = k, or n-1-k for the transposed solve) -----
// level 0 -> level 1: composite map of every chunk from the factor and the right-hand side
// loads of one position of the level-0 solve kernels, together and branch-free (see ParStepData)
template <typename T, int D> struct ParSolveStep { T L[D][D]; T x[D]; T W[D][D]; };
template <typename T, int D>
MF_DEV void par_solve_load(const T* __restrict__ ldiag, const T* __restrict__ lsub, const T* __restrict__ rhs, long s, long r,
                           long n, long p, int transpose, ParSolveStep<T, D>& d) {
    const long k = transpose ? n - 1 - p : p;
    load_lower<T, D>(ldiag + (s * n + k) * D * D, d.L);
    load_vec<T, D>(rhs + (r * n + k) * D, d.x);
    long kw = transpose ? k : k - 1;                   // coupling of position p with p-1; position 0: clamped, not used
    if (kw < 0) kw = 0;
    if (kw > n - 2) kw = n - 2;
    load_mat<T, D, D>(lsub + (s * (n - 1) + kw) * D * D, d.W);
}
template <typename T, int D> constexpr bool par_solve_prefetch() { return par_prefetch<T, D>() || (sizeof(T) == 8 && D <= 6); }

template <typename T, int D>
__global__ void __launch_bounds__(64) par_solve_up0_kernel(long Bl, long Br, long n, long len, long P,
                                                           const T* __restrict__ ldiag, const T* __restrict__ lsub,
                                                           const T* __restrict__ rhs, int transpose, T* __restrict__ oM,
                                                           T* __restrict__ oc) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= Br * P) return;
    const long r = id / P, c = id % P, s = r % Bl;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T Pm[D][D], q[D];
    MF_UNROLL for (int i = 0; i < D; ++i) { q[i] = T(0); MF_UNROLL for (int j = 0; j < D; ++j) Pm[i][j] = T(0); }
    constexpr bool PF = par_solve_prefetch<T, D>();
    ParSolveStep<T, D> cur, nxt;
    if (PF && p0 < p1) par_solve_load<T, D>(ldiag, lsub, rhs, s, r, n, p0, transpose, cur);
    for (long p = p0; p < p1; ++p) {
        if (PF) par_solve_load<T, D>(ldiag, lsub, rhs, s, r, n, p + 1 < p1 ? p + 1 : p, transpose, nxt);
        else par_solve_load<T, D>(ldiag, lsub, rhs, s, r, n, p, transpose, cur);
        __builtin_amdgcn_sched_barrier(0);
        T Li[D];
        MF_UNROLL for (int i = 0; i < D; ++i) Li[i] = t_rcp<T>(cur.L[i][i]);
        if (p > 0) {
            T W[D][D], wq[D], WP[D][D];
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) W[i][j] = transpose ? cur.W[j][i] : cur.W[i][j];
            gemv_n<T, D, D>(W, q, wq);
            MF_UNROLL for (int i = 0; i < D; ++i) cur.x[i] -= wq[i];
            if (p == p0) {
                MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) WP[i][j] = -W[i][j];
            } else {
                MF_UNROLL for (int i = 0; i < D; ++i)
                    MF_UNROLL for (int j = 0; j < D; ++j) {
                        T a = T(0);
                        MF_UNROLL for (int l = 0; l < D; ++l) a += W[i][l] * Pm[l][j];
                        WP[i][j] = -a;
                    }
            }
            if (!transpose) trsm_left_lower<T, D, D>(cur.L, Li, WP); else trsm_left_lower_t<T, D, D>(cur.L, Li, WP);
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Pm[i][j] = WP[i][j];
        }
        if (!transpose) trsv_lower<T, D>(cur.L, Li, cur.x); else trsv_lower_t<T, D>(cur.L, Li, cur.x);
        MF_UNROLL for (int i = 0; i < D; ++i) q[i] = cur.x[i];
        if (PF) cur = nxt;
    }
    store_mat<T, D, D>(oM + id * D * D, Pm);
    store_vec<T, D>(oc + id * D, q);
}

// level l -> level l+1 (l >= 1): compose explicit maps
template <typename T, int D>
__global__ void __launch_bounds__(64) par_affine_up_kernel(long Br, long n, long len, long P, const T* __restrict__ M,
                                                           const T* __restrict__ cv, T* __restrict__ oM,
                                                           T* __restrict__ oc) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= Br * P) return;
    const long r = id / P, c = id % P;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T Pm[D][D], q[D];
    load_mat<T, D, D>(M + (r * n + p0) * D * D, Pm);
    load_vec<T, D>(cv + (r * n + p0) * D, q);
    for (long p = p0 + 1; p < p1; ++p) {
        T Mp[D][D], cp[D], nq[D], nP[D][D];
        load_mat<T, D, D>(M + (r * n + p) * D * D, Mp);
        load_vec<T, D>(cv + (r * n + p) * D, cp);
        gemv_n<T, D, D>(Mp, q, nq);
        MF_UNROLL for (int i = 0; i < D; ++i) q[i] = nq[i] + cp[i];
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T a = T(0);
                MF_UNROLL for (int l = 0; l < D; ++l) a += Mp[i][l] * Pm[l][j];
                nP[i][j] = a;
            }
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Pm[i][j] = nP[i][j];
    }
    store_mat<T, D, D>(oM + id * D * D, Pm);
    store_vec<T, D>(oc + id * D, q);
}

// down-sweep on a level >= 1 (and, with len >= n and up = null, the serial walk of the coarsest level):
// Z[p] for every position of the level; `up` = Z of the next coarser level.
template <typename T, int D>
__global__ void __launch_bounds__(64) par_affine_down_kernel(long Br, long n, long len, long P, const T* __restrict__ M,
                                                             const T* __restrict__ cv, const T* __restrict__ up,
                                                             T* __restrict__ Z) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= Br * P) return;
    const long r = id / P, c = id % P;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T z[D];
    MF_UNROLL for (int i = 0; i < D; ++i) z[i] = T(0);
    if (c > 0) load_vec<T, D>(up + (r * P + c - 1) * D, z);
    for (long p = p0; p < p1; ++p) {
        T cp[D];
        load_vec<T, D>(cv + (r * n + p) * D, cp);
        if (p > 0) {
            T Mp[D][D], nz[D];
            load_mat<T, D, D>(M + (r * n + p) * D * D, Mp);
            gemv_n<T, D, D>(Mp, z, nz);
            MF_UNROLL for (int i = 0; i < D; ++i) z[i] = nz[i] + cp[i];
        } else {
            MF_UNROLL for (int i = 0; i < D; ++i) z[i] = cp[i];
        }
        store_vec<T, D>(Z + (r * n + p) * D, z);
    }
}

// level 0: every chunk redoes its substitution from the known incoming vector and writes the solution
template <typename T, int D>
__global__ void __launch_bounds__(64) par_solve_emit_kernel(long Bl, long Br, long n, long len, long P,
                                                            const T* __restrict__ ldiag, const T* __restrict__ lsub,
                                                            const T* __restrict__ rhs, const T* __restrict__ up,
                                                            int transpose, T* __restrict__ out) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= Br * P) return;
    const long r = id / P, c = id % P, s = r % Bl;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
#ifndef MF_SOLVE_PF2
#define MF_SOLVE_PF2 0
#endif
    // two positions ahead: measured neutral (1.32 vs 1.37 ms at B = 16384, T = 500, d = 6 fp64, within box-to-box noise): off
    constexpr bool PF2 = MF_SOLVE_PF2 && ((sizeof(T) == 8 && D <= 6) || (sizeof(T) == 4 && D <= 8));
    constexpr bool PF = par_solve_prefetch<T, D>() || PF2;
    ParSolveStep<T, D> cur, nxt, nx2;
    if (PF && p0 < p1) par_solve_load<T, D>(ldiag, lsub, rhs, s, r, n, p0, transpose, cur);
    if (PF2 && p0 < p1) par_solve_load<T, D>(ldiag, lsub, rhs, s, r, n, p0 + 1 < p1 ? p0 + 1 : p0, transpose, nxt);
    T z[D];
    MF_UNROLL for (int i = 0; i < D; ++i) z[i] = T(0);
    if (c > 0) load_vec<T, D>(up + (r * P + c - 1) * D, z);
    for (long p = p0; p < p1; ++p) {
        const long k = transpose ? n - 1 - p : p;
        if (PF2) par_solve_load<T, D>(ldiag, lsub, rhs, s, r, n, p + 2 < p1 ? p + 2 : p1 - 1, transpose, nx2);
        else if (PF) par_solve_load<T, D>(ldiag, lsub, rhs, s, r, n, p + 1 < p1 ? p + 1 : p, transpose, nxt);
        else par_solve_load<T, D>(ldiag, lsub, rhs, s, r, n, p, transpose, cur);
        __builtin_amdgcn_sched_barrier(0);
        T Li[D];
        MF_UNROLL for (int i = 0; i < D; ++i) Li[i] = t_rcp<T>(cur.L[i][i]);
        if (p > 0) {
            T wz[D];
            if (!transpose) gemv_n<T, D, D>(cur.W, z, wz); else gemv_t<T, D, D>(cur.W, z, wz);
            MF_UNROLL for (int i = 0; i < D; ++i) cur.x[i] -= wz[i];
        }
        if (!transpose) trsv_lower<T, D>(cur.L, Li, cur.x); else trsv_lower_t<T, D>(cur.L, Li, cur.x);
        MF_UNROLL for (int i = 0; i < D; ++i) z[i] = cur.x[i];
        store_vec<T, D>(out + (r * n + k) * D, z);
        if (PF) cur = nxt;
        if (PF2) nxt = nx2;
    }
}


// ---- StateSpaceModel.marginal_means in parallel in time: x_p = A_{p-1} x_{p-1} + o_p is already an affine recursion ------
// REV: the transposed recursion run backwards, lam_k = o_k + A_k^T lam_{k+1} (the adjoint of the means, mf_kl_grad.hpp):
// position p is block n-1-p and its matrix is A_{n-1-p}^T.
template <typename T, int D, bool REV>
MF_DEV void means_load_a(const T* __restrict__ A, long s, long n, long p, T (&Am)[D][D]) {
    if (!REV) {
        load_mat<T, D, D>(A + (s * (n - 1) + (p > 0 ? p - 1 : 0)) * D * D, Am);      // position 0: clamped, unused
    } else {
        T At[D][D];
        load_mat<T, D, D>(A + (s * (n - 1) + (p > 0 ? n - 1 - p : n - 2)) * D * D, At);
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Am[i][j] = At[j][i];
    }
}
template <typename T, int D, bool REV = false>
__global__ void __launch_bounds__(64) par_means_up0_kernel(long Bl, long Br, long n, long len, long P,
                                                           const T* __restrict__ A, const T* __restrict__ offs,
                                                           T* __restrict__ oM, T* __restrict__ oc) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= Br * P) return;
    const long r = id / P, c = id % P, s = r % Bl;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T Pm[D][D], q[D];
    MF_UNROLL for (int i = 0; i < D; ++i) { q[i] = T(0); MF_UNROLL for (int j = 0; j < D; ++j) Pm[i][j] = T(0); }
    for (long p = p0; p < p1; ++p) {
        T o[D];
        load_vec<T, D>(offs + (r * n + (REV ? n - 1 - p : p)) * D, o);
        if (p > 0) {
            T Am[D][D], nq[D];
            means_load_a<T, D, REV>(A, s, n, p, Am);
            gemv_n<T, D, D>(Am, q, nq);
            MF_UNROLL for (int i = 0; i < D; ++i) q[i] = nq[i] + o[i];
            if (p == p0) {
                MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Pm[i][j] = Am[i][j];
            } else {
                T nP[D][D];
                MF_UNROLL for (int i = 0; i < D; ++i)
                    MF_UNROLL for (int j = 0; j < D; ++j) {
                        T a = T(0);
                        MF_UNROLL for (int l = 0; l < D; ++l) a += Am[i][l] * Pm[l][j];
                        nP[i][j] = a;
                    }
                MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Pm[i][j] = nP[i][j];
            }
        } else {
            MF_UNROLL for (int i = 0; i < D; ++i) q[i] = o[i];
        }
    }
    store_mat<T, D, D>(oM + id * D * D, Pm);
    store_vec<T, D>(oc + id * D, q);
}

template <typename T, int D, bool REV = false>
__global__ void __launch_bounds__(64) par_means_emit_kernel(long Bl, long Br, long n, long len, long P,
                                                            const T* __restrict__ A, const T* __restrict__ offs,
                                                            const T* __restrict__ up, T* __restrict__ out) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= Br * P) return;
    const long r = id / P, c = id % P, s = r % Bl;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T x[D];
    MF_UNROLL for (int i = 0; i < D; ++i) x[i] = T(0);
    // a step is one d x d block and one vector: loads run a GROUP of G steps ahead of their use (G = 2 where two groups fit
    // the registers) - with one recursion per lane the bytes in flight, not the 2 d^2 flops of a step, set the rate
    constexpr int G = ((sizeof(T) == 8 && D <= 6) || (sizeof(T) == 4 && D <= 8)) ? 2 : 1;
    struct St { T A[D][D]; T o[D]; };
    auto load = [&](long p, St& d) {
        const long pc = p < p1 ? p : p1 - 1;
        load_vec<T, D>(offs + (r * n + (REV ? n - 1 - pc : pc)) * D, d.o);
        if (n > 1) means_load_a<T, D, REV>(A, s, n, pc, d.A);
    };
    St cur[G], nxt[G];
    if (p0 < p1) { MF_UNROLL for (int q = 0; q < G; ++q) load(p0 + q, cur[q]); }
    if (c > 0) load_vec<T, D>(up + (r * P + c - 1) * D, x);
    for (long p = p0; p < p1; p += G) {
        MF_UNROLL for (int q = 0; q < G; ++q) load(p + G + q, nxt[q]);
        __builtin_amdgcn_sched_barrier(0);
        MF_UNROLL for (int q = 0; q < G; ++q) {
            if (p + q < p1) {
                if (p + q > 0) {
                    T nx[D];
                    gemv_n<T, D, D>(cur[q].A, x, nx);
                    MF_UNROLL for (int i = 0; i < D; ++i) x[i] = nx[i] + cur[q].o[i];
                } else {
                    MF_UNROLL for (int i = 0; i < D; ++i) x[i] = cur[q].o[i];
                }
                store_vec<T, D>(out + (r * n + (REV ? n - 1 - (p + q) : p + q)) * D, x);
            }
        }
        MF_UNROLL for (int q = 0; q < G; ++q) cur[q] = nxt[q];
    }
}

// ---- block_diagonal_of_inverse (block Takahashi) in parallel in time ------------------------------------------------------------
// Backward congruence recursion over positions p = n-1-k:  Sigma(p) = N_p + G_p^T Sigma(p-1) G_p,  Sigma(0) = N_0, with
// N = L_k^-T L_k^-1 and G = W_k L_k^-1.  A run of positions composes into one (Gc, Nc):
//   Gc <- Gc G_p,   Nc <- N_p + G_p^T Nc G_p.
template <typename T, int D> MF_DEV void congruence_step(const T (&G)[D][D], const T (&N)[D][D], T (&Sig)[D][D]) {
    // Sig(full symmetric) <- N(lower) + G^T Sig G, one row of Sig G at a time (no d x d temporary for the product)
    T Out[D][D];
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Out[i][j] = N[i][j];
    MF_UNROLL for (int l = 0; l < D; ++l) {
        T sg[D];
        MF_UNROLL for (int j = 0; j < D; ++j) sg[j] = Sig[l][0] * G[0][j];
        MF_UNROLL for (int q = 1; q < D; ++q) MF_UNROLL for (int j = 0; j < D; ++j) sg[j] += Sig[l][q] * G[q][j];
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Out[i][j] += G[l][i] * sg[j];
    }
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Sig[i][j] = (i >= j) ? Out[i][j] : Out[j][i];
}
// The same with Sig held as its lower triangle only (45 instead of 81 registers at d = 9)
template <typename T, int D> MF_DEV void congruence_step_lower(const T (&G)[D][D], const T (&N)[D][D], T (&S)[D][D]) {
    T Out[D][D];
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Out[i][j] = N[i][j];
    MF_UNROLL for (int l = 0; l < D; ++l) {
        T sg[D];
        MF_UNROLL for (int j = 0; j < D; ++j) sg[j] = S[l][0] * G[0][j];                      // S(l, 0) = S[l][0]
        MF_UNROLL for (int q = 1; q < D; ++q) {
            const T slq = (l >= q) ? S[l][q] : S[q][l];
            MF_UNROLL for (int j = 0; j < D; ++j) sg[j] += slq * G[q][j];
        }
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Out[i][j] += G[l][i] * sg[j];
    }
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) S[i][j] = Out[i][j];
}
// Gc <- Gc G for a D x D matrix kept in LDS (element e of this lane at gc[e * 64]); one row at a time, loop not unrolled
template <typename T, int D> MF_DEV void lds_rows_times(T* gc, const T (&G)[D][D]) {
#pragma unroll 1
    for (int i = 0; i < D; ++i) {
        T* r = gc + i * D * 64;
        T row[D], o[D];
        MF_UNROLL for (int l = 0; l < D; ++l) row[l] = r[l * 64];
        MF_UNROLL for (int j = 0; j < D; ++j) o[j] = row[0] * G[0][j];
        MF_UNROLL for (int l = 1; l < D; ++l) MF_UNROLL for (int j = 0; j < D; ++j) o[j] += row[l] * G[l][j];
        MF_UNROLL for (int j = 0; j < D; ++j) r[j * 64] = o[j];
    }
}
// N (lower) and G of block k from the factor
template <typename T, int D>
MF_DEV void takahashi_terms(const T* __restrict__ ldiag, const T* __restrict__ lsub, long s, long n, long k, bool has_g,
                            T (&N)[D][D], T (&G)[D][D]) {
    T L[D][D], Linv[D][D];
    load_lower<T, D>(ldiag + (s * n + k) * D * D, L);
    LogAcc<T> la;
    la.init();
    bool bad = false;
    tri_inv_lower<T, D>(L, Linv, la, bad);
    trimulT_self_lower<T, D>(Linv, N);
    if (has_g) {
        T W[D][D];
        load_mat<T, D, D>(lsub + (s * (n - 1) + k) * D * D, W);
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T a = T(0);
                MF_UNROLL for (int l = j; l < D; ++l) a += W[i][l] * Linv[l][j];
                G[i][j] = a;
            }
    }
}

// the same from blocks that are already in registers
template <typename T, int D>
MF_DEV void takahashi_from(const T (&L)[D][D], const T (&W)[D][D], bool has_g, T (&N)[D][D], T (&G)[D][D]) {
    T Linv[D][D];
    LogAcc<T> la;
    la.init();
    bool bad = false;
    tri_inv_lower<T, D>(L, Linv, la, bad);
    trimulT_self_lower<T, D>(Linv, N);
    if (has_g) {
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T a = T(0);
                MF_UNROLL for (int l = j; l < D; ++l) a += W[i][l] * Linv[l][j];
                G[i][j] = a;
            }
    }
}

// Source of the congruence recursion  Sigma(p) = N_p + G_p^T Sigma(p-1) G_p  of the level-0 kernels below:
//   SRC 0  block Takahashi on a Cholesky factor (block_diagonal_of_inverse): position p = block n-1-p,
//          N = L_k^-T L_k^-1, G = W_k L_k^-1;  a = ldiag, b = lsub;
//   SRC 1  marginal covariances of a state space model, Sigma_{k+1} = A_k Sigma_k A_k^T + Q_k (state_space_model.py:254-275
//          computes them as the block diagonal of the inverse precision; the forward recursion gives the same blocks without
//          assembling or factorising the precision): position p = block p, N = C C^T with C = cholP0 (p = 0) or cholQ_{p-1},
//          G = A_{p-1}^T;  a = cholQ, b = A, c0 = cholP0.
//   SRC 2  adjoint of those marginal covariances, M_k = N_k + A_k^T M_{k+1} A_k (the reverse-mode sweep of `marginals` and of
//          kl_divergence, mf_kl_grad.hpp): position p = block n-1-p, N read from a buffer of symmetric blocks, G = A_k;
//          a = N [B,n,D,D], b = A.
//   SRC 3  the FORWARD recursion with explicit terms, Sigma_{p} = N_p + M_{p-1} Sigma_{p-1} M_{p-1}^T (the adjoint sweep of the block
//          Takahashi recursion, mf_btd_diag_of_inverse_grad): position p = block p, N read from a buffer of symmetric blocks,
//          G = M_{p-1}^T;  a = N [B,n,D,D], b = M [B,n-1,D,D].
template <typename T> struct TakSrc {
    const T* a;
    const T* b;
    const T* c0;
};
template <typename T, int D, int SRC>
MF_DEV void tak_load(const TakSrc<T>& src, long s, long n, long p, T (&L)[D][D], T (&W)[D][D]) {
    if (SRC == 0) {
        const long k = n - 1 - p;                        // the coupling of position 0 does not exist: clamped, not used
        load_lower<T, D>(src.a + (s * n + k) * D * D, L);
        load_mat<T, D, D>(src.b + (s * (n - 1) + (k < n - 1 ? k : n - 2)) * D * D, W);
    } else if (SRC == 2) {
        const long k = n - 1 - p;                        // block n-1 (position 0) has no transition: clamped, not used
        load_lower<T, D>(src.a + (s * n + k) * D * D, L);
        if (n > 1) load_mat<T, D, D>(src.b + (s * (n - 1) + (k < n - 1 ? k : n - 2)) * D * D, W);
    } else if (SRC == 3) {
        load_lower<T, D>(src.a + (s * n + p) * D * D, L);
        if (n > 1) load_mat<T, D, D>(src.b + (s * (n - 1) + (p > 0 ? p - 1 : 0)) * D * D, W);
    } else {
        const T* lp = p > 0 ? src.a + (s * (n - 1) + p - 1) * D * D : src.c0 + s * D * D;
        load_lower<T, D>(lp, L);
        load_mat<T, D, D>(src.b + (s * (n - 1) + (p > 0 ? p - 1 : 0)) * D * D, W);
    }
}
template <typename T, int D, int SRC>
MF_DEV void tak_terms(const T (&L)[D][D], const T (&W)[D][D], bool has_g, T (&N)[D][D], T (&G)[D][D]) {
    if (SRC == 0) {
        takahashi_from<T, D>(L, W, has_g, N, G);
    } else if (SRC == 2) {
        MF_UNROLL for (int i = 0; i < D; ++i) {
            MF_UNROLL for (int j = 0; j <= i; ++j) N[i][j] = L[i][j];
            MF_UNROLL for (int j = 0; j < D; ++j) G[i][j] = W[i][j];
        }
    } else if (SRC == 3) {
        MF_UNROLL for (int i = 0; i < D; ++i) {
            MF_UNROLL for (int j = 0; j <= i; ++j) N[i][j] = L[i][j];
            MF_UNROLL for (int j = 0; j < D; ++j) G[i][j] = W[j][i];
        }
    } else {
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                T a = L[i][0] * L[j][0];
                MF_UNROLL for (int q = 1; q <= j; ++q) a += L[i][q] * L[j][q];
                N[i][j] = a;
            }
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) G[i][j] = W[j][i];
    }
}

// MEAN (SRC 1): the level-0 up-sweep of the marginal MEANS rides along (the chunk's affine map x -> M x + q: M is the transpose
// of the composed G, q needs one matrix-vector product per step with the transition that is loaded anyway) - outputs in the
// layout of par_means_up0_kernel, so that the affine levels above run unchanged
template <typename T> struct TakMeanUp {
    const T* mu0;      // [B, D]
    const T* b;        // [B, n-1, D]
    T* oM;             // [B, P, D, D]
    T* oc;             // [B, P, D]
};
template <typename T, int D, bool MEAN>
MF_DEV void tak_mean_step(const TakMeanUp<T>& m, long s, long n, long p, const T (&G)[D][D], T (&q)[MEAN ? D : 1]) {
    if constexpr (MEAN) {
        T o[D];
        load_vec<T, D>(p > 0 ? m.b + (s * (n - 1) + p - 1) * D : m.mu0 + s * D, o);
        if (p > 0) {
            T nq[D];
            gemv_t<T, D, D>(G, q, nq);                       // A q with G = A^T
            MF_UNROLL for (int i = 0; i < D; ++i) q[i] = nq[i] + o[i];
        } else {
            MF_UNROLL for (int i = 0; i < D; ++i) q[i] = o[i];
        }
    }
}
template <typename T, int D, int SRC, bool MEAN = false>
__global__ void __launch_bounds__(64) par_tak_up0_kernel(long B, long n, long len, long P, TakSrc<T> src, T* __restrict__ oG,
                                                         T* __restrict__ oN, TakMeanUp<T> mup = TakMeanUp<T>{}) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * P) return;
    const long s = id / P, c = id % P;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T Gc[D][D], Nc[D][D];
    T q[MEAN ? D : 1];
    MF_UNROLL for (int i = 0; i < (MEAN ? D : 1); ++i) q[i] = T(0);
    for (long p = p0; p < p1; ++p) {
        T N[D][D], G[D][D];
        {
            T L[D][D], W[D][D];
            tak_load<T, D, SRC>(src, s, n, p, L, W);
            tak_terms<T, D, SRC>(L, W, p > 0, N, G);
        }
        tak_mean_step<T, D, MEAN>(mup, s, n, p, G, q);
        if (p == p0) {
            MF_UNROLL for (int i = 0; i < D; ++i)
                MF_UNROLL for (int j = 0; j < D; ++j) { Nc[i][j] = (i >= j) ? N[i][j] : N[j][i]; Gc[i][j] = (p > 0) ? G[i][j] : T(0); }
        } else {
            congruence_step<T, D>(G, N, Nc);
            T nG[D][D];
            MF_UNROLL for (int i = 0; i < D; ++i)
                MF_UNROLL for (int j = 0; j < D; ++j) {
                    T a = T(0);
                    MF_UNROLL for (int l = 0; l < D; ++l) a += Gc[i][l] * G[l][j];
                    nG[i][j] = a;
                }
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Gc[i][j] = nG[i][j];
        }
    }
    store_mat<T, D, D>(oG + id * D * D, Gc);
    store_mat<T, D, D>(oN + id * D * D, Nc);
    if constexpr (MEAN) {
        T Mt[D][D];
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Mt[i][j] = Gc[j][i];
        store_mat<T, D, D>(mup.oM + id * D * D, Mt);
        store_vec<T, D>(mup.oc + id * D, q);
    }
}

template <typename T, int D>
__global__ void __launch_bounds__(64) par_tak_up_kernel(long B, long n, long len, long P, const T* __restrict__ Gs,
                                                        const T* __restrict__ Ns, T* __restrict__ oG, T* __restrict__ oN) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * P) return;
    const long s = id / P, c = id % P;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T Gc[D][D], Nc[D][D];
    load_mat<T, D, D>(Gs + (s * n + p0) * D * D, Gc);
    load_mat<T, D, D>(Ns + (s * n + p0) * D * D, Nc);
    for (long p = p0 + 1; p < p1; ++p) {
        T G[D][D], N[D][D], nG[D][D];
        load_mat<T, D, D>(Gs + (s * n + p) * D * D, G);
        load_lower<T, D>(Ns + (s * n + p) * D * D, N);
        congruence_step<T, D>(G, N, Nc);
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T a = T(0);
                MF_UNROLL for (int l = 0; l < D; ++l) a += Gc[i][l] * G[l][j];
                nG[i][j] = a;
            }
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Gc[i][j] = nG[i][j];
    }
    store_mat<T, D, D>(oG + id * D * D, Gc);
    store_mat<T, D, D>(oN + id * D * D, Nc);
}

// The two up-sweeps with the composed G of the run in LDS and Nc as a lower triangle (d >= 7 in fp64, see mf_kf_x.hpp)
template <typename T, int D, int SRC, bool MEAN = false>
__global__ void __launch_bounds__(64) par_tak_up0_x_kernel(long B, long n, long len, long P, TakSrc<T> src, T* __restrict__ oG,
                                                           T* __restrict__ oN, TakMeanUp<T> mup = TakMeanUp<T>{}) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* gc = reinterpret_cast<T*>(smem_raw) + threadIdx.x;
    const long total = B * P;
    const long id_raw = (long)blockIdx.x * 64 + threadIdx.x;
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;
    const long s = id / P, c = id % P;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    if (!valid) p1 = p0;
    T Nc[D][D];
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Nc[i][j] = T(0);
    T q[MEAN ? D : 1];
    MF_UNROLL for (int i = 0; i < (MEAN ? D : 1); ++i) q[i] = T(0);
    for (long p = p0; p < p1; ++p) {
        T N[D][D], G[D][D];
        {
            T L[D][D], W[D][D];
            tak_load<T, D, SRC>(src, s, n, p, L, W);
            tak_terms<T, D, SRC>(L, W, p > 0, N, G);
        }
        tak_mean_step<T, D, MEAN>(mup, s, n, p, G, q);
        if (p == p0) {
            MF_UNROLL for (int i = 0; i < D; ++i)
                MF_UNROLL for (int j = 0; j < D; ++j) {
                    if (j <= i) Nc[i][j] = N[i][j];
                    gc[(i * D + j) * 64] = (p > 0) ? G[i][j] : T(0);
                }
        } else {
            congruence_step_lower<T, D>(G, N, Nc);
            lds_rows_times<T, D>(gc, G);
        }
    }
    if (valid) {
        store_sym<T, D>(oN + id * D * D, Nc);
        T* g = oG + id * D * D;
        MF_UNROLL for (int e = 0; e < D * D; ++e) g[e] = gc[e * 64];
        if constexpr (MEAN) {
            T* mt = mup.oM + id * D * D;
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) mt[i * D + j] = gc[(j * D + i) * 64];
            store_vec<T, D>(mup.oc + id * D, q);
        }
    }
}

template <typename T, int D>
__global__ void __launch_bounds__(64) par_tak_up_x_kernel(long B, long n, long len, long P, const T* __restrict__ Gs,
                                                          const T* __restrict__ Ns, T* __restrict__ oG, T* __restrict__ oN) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* gc = reinterpret_cast<T*>(smem_raw) + threadIdx.x;
    const long total = B * P;
    const long id_raw = (long)blockIdx.x * 64 + threadIdx.x;
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;
    const long s = id / P, c = id % P;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T Nc[D][D];
    load_lower<T, D>(Ns + (s * n + p0) * D * D, Nc);
    {
        const T* g0 = Gs + (s * n + p0) * D * D;
        MF_UNROLL for (int e = 0; e < D * D; ++e) gc[e * 64] = g0[e];
    }
    if (!valid) p1 = p0;
    for (long p = p0 + 1; p < p1; ++p) {
        T G[D][D], N[D][D];
        load_mat<T, D, D>(Gs + (s * n + p) * D * D, G);
        load_lower<T, D>(Ns + (s * n + p) * D * D, N);
        congruence_step_lower<T, D>(G, N, Nc);
        lds_rows_times<T, D>(gc, G);
    }
    if (valid) {
        store_sym<T, D>(oN + id * D * D, Nc);
        T* g = oG + id * D * D;
        MF_UNROLL for (int e = 0; e < D * D; ++e) g[e] = gc[e * 64];
    }
}

// Sigma at every position of a level >= 1 (coarsest level: len >= n, up = null)
template <typename T, int D>
__global__ void __launch_bounds__(64) par_tak_down_kernel(long B, long n, long len, long P, const T* __restrict__ Gs,
                                                          const T* __restrict__ Ns, const T* __restrict__ up,
                                                          T* __restrict__ Z) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * P) return;
    const long s = id / P, c = id % P;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T Sig[D][D];                                     // lower triangle
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Sig[i][j] = T(0);
    if (c > 0) load_lower<T, D>(up + (s * P + c - 1) * D * D, Sig);
    for (long p = p0; p < p1; ++p) {
        if (p > 0) {
            T G[D][D], N[D][D];
            load_mat<T, D, D>(Gs + (s * n + p) * D * D, G);
            load_lower<T, D>(Ns + (s * n + p) * D * D, N);
            congruence_step_lower<T, D>(G, N, Sig);
        } else {
            load_lower<T, D>(Ns + (s * n) * D * D, Sig);
        }
        store_sym<T, D>(Z + (s * n + p) * D * D, Sig);
    }
}

// MEAN (SRC 1): the marginal means mu_p = A_{p-1} mu_{p-1} + b_{p-1} ride along - `marginals` in one sweep that reads A once
// (mf_ssm_marginals_*).  One chunk per series, or - with the means at the chunk ends from the affine scan (`up`) - many.
template <typename T> struct TakMean {
    const T* mu0;      // [B, D]
    const T* b;        // [B, n-1, D]
    T* out;            // [B, n, D]
    const T* up;       // [B, P, D]: mean at the last position of every chunk (P > 1), else NULL
};
template <typename T, int D, int SRC, bool MEAN = false>
__global__ void __launch_bounds__(64) par_tak_emit_kernel(long B, long n, long len, long P, TakSrc<T> src,
                                                          const T* __restrict__ up, T* __restrict__ odiag,
                                                          T* __restrict__ osub, TakMean<T> mean = TakMean<T>{}) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * P) return;
    const long s = id / P, c = id % P;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    struct Step { T L[D][D]; T W[D][D]; T o[MEAN ? D : 1]; };
    auto load = [&](long p, Step& d) {
        tak_load<T, D, SRC>(src, s, n, p, d.L, d.W);
        if constexpr (MEAN) load_vec<T, D>(p > 0 ? mean.b + (s * (n - 1) + p - 1) * D : mean.mu0 + s * D, d.o);
    };
    constexpr bool PF = par_prefetch<T, D>() || (sizeof(T) == 8 && D <= 6);
    T mu[MEAN ? D : 1];
    Step cur, nxt;
    if (PF && p0 < p1) load(p0, cur);
    T Sig[D][D];                                     // lower triangle
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Sig[i][j] = T(0);
    if (c > 0) load_lower<T, D>(up + (s * P + c - 1) * D * D, Sig);
    if constexpr (MEAN) {
        if (c > 0) load_vec<T, D>(mean.up + (s * P + c - 1) * D, mu);
    }
    for (long p = p0; p < p1; ++p) {
        const long k = (SRC == 1 || SRC == 3) ? p : n - 1 - p;
        if (PF) load(p + 1 < p1 ? p + 1 : p, nxt);
        else load(p, cur);
        __builtin_amdgcn_sched_barrier(0);
        T N[D][D], G[D][D];
        tak_terms<T, D, SRC>(cur.L, cur.W, p > 0, N, G);
        if constexpr (MEAN) {
            if (p > 0) {
                T nm[D];
                gemv_n<T, D, D>(cur.W, mu, nm);
                MF_UNROLL for (int i = 0; i < D; ++i) mu[i] = nm[i] + cur.o[i];
            } else {
                MF_UNROLL for (int i = 0; i < D; ++i) mu[i] = cur.o[i];
            }
            store_vec<T, D>(mean.out + (s * n + p) * D, mu);
        }
        if (PF) cur = nxt;
        if (p > 0) {
            if (osub) {
                // SRC 0: block (k+1, k) of the inverse = -Sigma_{k+1} G_k;  SRC 1: Cov(x_p, x_{p-1}) = A Sigma_{p-1} = (Sigma_{p-1} G)^T
                T* o = osub + (s * (n - 1) + (SRC == 1 ? p - 1 : k)) * D * D;
                MF_UNROLL for (int i = 0; i < D; ++i) {
                    T row[D];
                    MF_UNROLL for (int j = 0; j < D; ++j) row[j] = Sig[i][0] * G[0][j];
                    MF_UNROLL for (int l = 1; l < D; ++l) {
                        const T sil = (i >= l) ? Sig[i][l] : Sig[l][i];
                        MF_UNROLL for (int j = 0; j < D; ++j) row[j] += sil * G[l][j];
                    }
                    if (SRC != 1) { MF_UNROLL for (int j = 0; j < D; ++j) o[i * D + j] = -row[j]; }
                    else { MF_UNROLL for (int j = 0; j < D; ++j) o[j * D + i] = row[j]; }
                }
            }
            congruence_step_lower<T, D>(G, N, Sig);
        } else {
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Sig[i][j] = N[i][j];
        }
        store_sym<T, D>(odiag + (s * n + k) * D * D, Sig);
    }
}


// ---- reverse mode through the operators (banded_matrices registers a gradient for every op, block_tri_diag.py:22-31) -----------
// The block Cholesky  P_k = D_k - S_{k-1} P_{k-1}^-1 S_{k-1}^T,  L_k = chol(P_k),  W_k = S_k L_k^-T  is a LOCAL map (P_k, S_k) ->
// (L_k, W_k) behind a Riccati-type recursion in P_k.  Its adjoint therefore splits into a part that is local in time - the only
// place where the projection Phi of the dense Cholesky adjoint acts - and the adjoint of the recursion, which is a congruence
// recursion with the coupling of the block Takahashi recursion, G_k = W_k L_k^-1:
//     Sbar_k(loc) = Wbar_k L_k^-1,   Lbar_k(eff) = Lbar_k - tril(Sbar_k(loc)^T W_k),   C_k = sym(L_k^-T Phi(L_k^T Lbar_k(eff)) L_k^-1)
//     Dbar_k = Z_k,  Z_k = C_k + G_k^T Z_{k+1} G_k,   Sbar_k = Sbar_k(loc) - 2 Z_{k+1} G_k
// so the backward pass is one fully parallel kernel (below), the congruence scan of the marginals' adjoint (SRC 2: parallel in time
// for few series, a lane per series for many) and an axpy - no sequential sweep of its own.
template <typename T, int D>
__global__ void __launch_bounds__(64) btd_chol_grad_local_kernel(long B, long n, const T* __restrict__ ldiag, const T* __restrict__ lsub,
                                                                 const T* __restrict__ gl, const T* __restrict__ gw,
                                                                 T* __restrict__ oC, T* __restrict__ oG, T* __restrict__ oS) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * n) return;
    const long s = id / n, k = id % n;
    T L[D][D], Li[D][D], Lb[D][D];
    load_lower<T, D>(ldiag + id * D * D, L);
    LogAcc<T> la;
    la.init();
    bool bad = false;
    tri_inv_lower<T, D>(L, Li, la, bad);
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j < D; ++j) Lb[i][j] = (gl && j <= i) ? gl[id * D * D + i * D + j] : T(0);
    if (lsub && k < n - 1) {
        const long ks = s * (n - 1) + k;
        T W[D][D], X[D][D];
        load_mat<T, D, D>(lsub + ks * D * D, W);
        // G = W Li (lower-triangular Li: columns j use rows l >= j)
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T a = T(0);
                MF_UNROLL for (int l = j; l < D; ++l) a += W[i][l] * Li[l][j];
                X[i][j] = a;
            }
        store_mat<T, D, D>(oG + ks * D * D, X);
        if (gw) {
            T Wb[D][D];
            load_mat<T, D, D>(gw + ks * D * D, Wb);
            MF_UNROLL for (int i = 0; i < D; ++i)
                MF_UNROLL for (int j = 0; j < D; ++j) {
                    T a = T(0);
                    MF_UNROLL for (int l = j; l < D; ++l) a += Wb[i][l] * Li[l][j];
                    X[i][j] = a;                                         // Sbar(loc) = Wbar Li
                }
            MF_UNROLL for (int i = 0; i < D; ++i)
                MF_UNROLL for (int j = 0; j <= i; ++j) {
                    T a = T(0);
                    MF_UNROLL for (int l = 0; l < D; ++l) a += X[l][i] * W[l][j];
                    Lb[i][j] -= a;                                       // - tril(Sbar(loc)^T W)
                }
        } else {
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) X[i][j] = T(0);
        }
        store_mat<T, D, D>(oS + ks * D * D, X);
    }
    // Phi = tril(L^T Lbar), diagonal halved;  Y = Phi Li (lower);  X = Li^T Y;  C = (X + X^T) / 2
    T Phi[D][D];
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j <= i; ++j) {
            T a = T(0);
            MF_UNROLL for (int l = i; l < D; ++l) a += L[l][i] * Lb[l][j];
            Phi[i][j] = (i == j) ? T(0.5) * a : a;
        }
    T Y[D][D];
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j <= i; ++j) {
            T a = T(0);
            MF_UNROLL for (int l = j; l <= i; ++l) a += Phi[i][l] * Li[l][j];
            Y[i][j] = a;
        }
    T* o = oC + id * D * D;
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j <= i; ++j) {
            T a = T(0), b = T(0);                                        // X[i][j], X[j][i]
            MF_UNROLL for (int l = i; l < D; ++l) a += Li[l][i] * Y[l][j];
            MF_UNROLL for (int l = i; l < D; ++l) b += Li[l][j] * Y[l][i];       // (Y[l][i] = 0 for l < i)
            const T c = T(0.5) * (a + b);
            o[i * D + j] = c;
            o[j * D + i] = c;
        }
}
// out[i] += alpha * x[i]
template <typename T> __global__ void __launch_bounds__(256) axpy_kernel(long n, T alpha, const T* __restrict__ x, T* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += alpha * x[i];
}

// The block Takahashi recursion  Sigma_k = B_k + G_k^T Sigma_{k+1} G_k,  sub_k = -Sigma_{k+1} G_k  (B_k = L_k^-T L_k^-1, G_k = W_k L_k^-1)
// run backwards in reverse mode is the same congruence recursion run FORWARD with G instead of G^T:
//     A_0 = sym(Sigmabar_0),   A_{k+1} = sym(Sigmabar_{k+1}) - sym(subbar_k G_k^T) + G_k A_k G_k^T            (SRC 3)
// between two kernels that are local in time: `pre` forms G_k and the explicit terms, `post` turns the totals A_k into
//     Lbar_k = -2 tril(L_k^-T (L_k^-1 A_k L_k^-T)) - tril(G_k^T Wbar_k),   Wbar_k = (2 Sigma_{k+1} G_k A_k - Sigma_{k+1} subbar_k) L_k^-T.
template <typename T, int D>
__global__ void __launch_bounds__(64) btd_inv_grad_pre_kernel(long B, long n, const T* __restrict__ ldiag, const T* __restrict__ lsub,
                                                              const T* __restrict__ gd, const T* __restrict__ gs,
                                                              T* __restrict__ oQ, T* __restrict__ oG) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * n) return;
    const long s = id / n, k = id % n;
    auto sym_in = [&](long blk, T (&Q)[D][D]) {
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                const T v = gd ? T(0.5) * (gd[blk * D * D + i * D + j] + gd[blk * D * D + j * D + i]) : T(0);
                Q[i][j] = v;
            }
    };
    T Q[D][D];
    if (k == 0 || !lsub) {                        // (no coupling: every block is its own chain and its total is its own term)
        sym_in(id, Q);
        store_sym<T, D>(oQ + id * D * D, Q);
    }
    if (!(lsub && k < n - 1)) return;
    const long ks = s * (n - 1) + k;
    T L[D][D], Li[D][D], W[D][D], G[D][D];
    load_lower<T, D>(ldiag + id * D * D, L);
    LogAcc<T> la;
    la.init();
    bool bad = false;
    tri_inv_lower<T, D>(L, Li, la, bad);
    load_mat<T, D, D>(lsub + ks * D * D, W);
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j < D; ++j) {
            T a = T(0);
            MF_UNROLL for (int l = j; l < D; ++l) a += W[i][l] * Li[l][j];
            G[i][j] = a;
        }
    store_mat<T, D, D>(oG + ks * D * D, G);
    sym_in(id + 1, Q);
    if (gs) {
        T Sb[D][D];
        load_mat<T, D, D>(gs + ks * D * D, Sb);
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                T a = T(0);
                MF_UNROLL for (int l = 0; l < D; ++l) a += Sb[i][l] * G[j][l] + Sb[j][l] * G[i][l];
                Q[i][j] -= T(0.5) * a;                                   // - sym(subbar G^T)
            }
    }
    store_sym<T, D>(oQ + (id + 1) * D * D, Q);
}
template <typename T, int D>
__global__ void __launch_bounds__(64) btd_inv_grad_post_kernel(long B, long n, const T* __restrict__ ldiag, const T* __restrict__ lsub,
                                                               const T* __restrict__ sig, const T* __restrict__ tot,
                                                               const T* __restrict__ gs, const T* __restrict__ Gk,
                                                               T* __restrict__ g_ldiag, T* __restrict__ g_lsub) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * n) return;
    const long s = id / n, k = id % n;
    T L[D][D], Li[D][D], A[D][D];
    load_lower<T, D>(ldiag + id * D * D, L);
    LogAcc<T> la;
    la.init();
    bool bad = false;
    tri_inv_lower<T, D>(L, Li, la, bad);
    load_mat<T, D, D>(tot + id * D * D, A);                              // symmetric, stored full
    // M1 = Li A Li^T (symmetric);  Lbar = -2 tril(Li^T M1)
    T X[D][D], M1[D][D], Lb[D][D];
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j < D; ++j) {
            T a = T(0);
            MF_UNROLL for (int l = 0; l <= i; ++l) a += Li[i][l] * A[l][j];
            X[i][j] = a;
        }
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j < D; ++j) {
            T a = T(0);
            MF_UNROLL for (int l = 0; l <= j; ++l) a += X[i][l] * Li[j][l];
            M1[i][j] = a;
        }
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j <= i; ++j) {
            T a = T(0);
            MF_UNROLL for (int l = i; l < D; ++l) a += Li[l][i] * M1[l][j];
            Lb[i][j] = T(-2) * a;
        }
    if (lsub && k < n - 1) {
        const long ks = s * (n - 1) + k;
        T G[D][D], Sg[D][D], Gb[D][D];
        load_mat<T, D, D>(Gk + ks * D * D, G);
        load_mat<T, D, D>(sig + (id + 1) * D * D, Sg);                   // Sigma_{k+1}
        // X = 2 G A - subbar;  Gbar = Sigma_{k+1} X
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T a = T(0);
                MF_UNROLL for (int l = 0; l < D; ++l) a += G[i][l] * A[l][j];
                X[i][j] = T(2) * a - (gs ? gs[ks * D * D + i * D + j] : T(0));
            }
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T a = T(0);
                MF_UNROLL for (int l = 0; l < D; ++l) a += Sg[i][l] * X[l][j];
                Gb[i][j] = a;
            }
        // Wbar = Gbar Li^T;  Lbar -= tril(G^T Wbar)
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T a = T(0);
                MF_UNROLL for (int l = 0; l <= j; ++l) a += Gb[i][l] * Li[j][l];
                X[i][j] = a;
            }
        store_mat<T, D, D>(g_lsub + ks * D * D, X);
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                T a = T(0);
                MF_UNROLL for (int l = 0; l < D; ++l) a += G[l][i] * X[l][j];
                Lb[i][j] -= a;
            }
    }
    T* o = g_ldiag + id * D * D;
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) o[i * D + j] = (j <= i) ? Lb[i][j] : T(0);
}

// ---- upper_diagonal_lower (U D U^T) and the posterior chain in parallel in time -------------------------------------------------
// Level-0 emit of the reversed factorisation: chunk c covers positions [c len, ...) (position p = block n-1-p) and restarts
// the backward recursion  Delta_k = D_k - S_k^T Delta_{k+1}^-1 S_k  from the pivot at position c len - 1 (`up`).
template <typename T, int D>
__global__ void __launch_bounds__(64) par_udl_emit_kernel(long B, long n, long len, long P, const T* __restrict__ diag,
                                                          const T* __restrict__ sub, const T* __restrict__ up,
                                                          T* __restrict__ ut, T* __restrict__ chol_d,
                                                          T* __restrict__ chol_dinv, int chain, int* info) {
    // chain (posterior_state_space_model in StateSpaceModel's layout): ut receives the posterior transitions -U_k^T and
    // chol_dinv the factors chol(Delta_k^-1) right here, where chol(Delta_k) is in registers - the offsets kernel
    // (par_post_emit_kernel) then only runs the affine recursion; chol_d is optional in that mode
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * P) return;
    const long s = id / P, c = id % P;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T Lp[D][D], Lpi[D];
    LogAcc<T> la;
    la.init();
    bool bad = false;
    if (c > 0) {
        load_lower<T, D>(up + (s * P + c - 1) * D * D, Lp);
        chol_lower<T, D>(Lp, Lpi, la, bad);
        la.init();
    }
    for (long p = p0; p < p1; ++p) {
        const long k = n - 1 - p;
        T Dl[D][D];
        load_lower<T, D>(diag + (s * n + k) * D * D, Dl);
        if (p > 0) {
            T U[D][D];
            load_mat<T, D, D>(sub + (s * (n - 1) + k) * D * D, U);
            trsm_left_lower<T, D, D>(Lp, Lpi, U);          // L^-1 S
            syrk_tn_lower<T, D, D>(U, Dl, T(-1));          // Delta_k = D_k - S^T Delta_{k+1}^-1 S
            trsm_left_lower_t<T, D, D>(Lp, Lpi, U);        // U_k^T = Delta_{k+1}^-1 S
            if (chain) { MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) U[i][j] = -U[i][j]; }
            store_mat<T, D, D>(ut + (s * (n - 1) + k) * D * D, U);
        }
        chol_lower<T, D>(Dl, Lpi, la, bad);
        la.init();
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Lp[i][j] = Dl[i][j];
        if (chol_d) store_lower<T, D>(chol_d + (s * n + k) * D * D, Lp);
        if (chain) {
            // chol(Delta_k^-1) = chol(L^-T L^-1), at its place in the chain: block 0 of all series first, then [B, n-1]
            T Linv[D][D], Q[D][D], Qi[D];
            tri_inv_lower<T, D>(Lp, Linv, la, bad);
            trimulT_self_lower<T, D>(Linv, Q);
            la.init();
            chol_lower<T, D>(Q, Qi, la, bad);
            la.init();
            store_lower<T, D>(chol_dinv + (k == 0 ? s : B + s * (n - 1) + k - 1) * D * D, Q);
        }
    }
    if (bad && info) raise_info(info);
}

// posterior chain, backward affine recursion over positions:  x(p) = eta_k - U_k x(p-1),  U_k = (ut[k])^T, k = n-1-p
template <typename T, int D>
__global__ void __launch_bounds__(64) par_post_up0_kernel(long B, long n, long len, long P, const T* __restrict__ ut,
                                                          const T* __restrict__ eta, T* __restrict__ oM, T* __restrict__ oc,
                                                          int neg) {      // neg: ut holds -U_k^T (chain layout)
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * P) return;
    const long s = id / P, c = id % P;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T Pm[D][D], q[D];
    MF_UNROLL for (int i = 0; i < D; ++i) { q[i] = T(0); MF_UNROLL for (int j = 0; j < D; ++j) Pm[i][j] = T(0); }
    for (long p = p0; p < p1; ++p) {
        const long k = n - 1 - p;
        T e[D];
        load_vec<T, D>(eta + (s * n + k) * D, e);
        if (p > 0) {
            T Ut[D][D], uq[D];
            load_mat<T, D, D>(ut + (s * (n - 1) + k) * D * D, Ut);
            if (neg) { MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Ut[i][j] = -Ut[i][j]; }
            gemv_t<T, D, D>(Ut, q, uq);                    // U_k q
            MF_UNROLL for (int i = 0; i < D; ++i) q[i] = e[i] - uq[i];
            T nP[D][D];
            if (p == p0) {
                MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) nP[i][j] = -Ut[j][i];
            } else {
                MF_UNROLL for (int i = 0; i < D; ++i)
                    MF_UNROLL for (int j = 0; j < D; ++j) {
                        T a = T(0);
                        MF_UNROLL for (int l = 0; l < D; ++l) a += Ut[l][i] * Pm[l][j];
                        nP[i][j] = -a;
                    }
            }
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Pm[i][j] = nP[i][j];
        } else {
            MF_UNROLL for (int i = 0; i < D; ++i) q[i] = e[i];
        }
    }
    store_mat<T, D, D>(oM + id * D * D, Pm);
    store_vec<T, D>(oc + id * D, q);
}

// level 0: x_k, then m_k = Delta_k^-1 x_k and chol(Delta_k^-1) (kalman_filter.py:159-174)
template <typename T, int D>
__global__ void __launch_bounds__(64) par_post_emit_kernel(long B, long n, long len, long P, T* __restrict__ ut,
                                                           const T* __restrict__ chol_d, const T* __restrict__ eta,
                                                           const T* __restrict__ up, T* __restrict__ m_post,
                                                           T* __restrict__ chol_dinv, int chain, int* info) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * P) return;
    const long s = id / P, c = id % P;
    const long p0 = c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T xp[D];
    MF_UNROLL for (int i = 0; i < D; ++i) xp[i] = T(0);
    if (c > 0) load_vec<T, D>(up + (s * P + c - 1) * D, xp);
    bool bad = false;
    for (long p = p0; p < p1; ++p) {
        const long k = n - 1 - p;
        T x[D];
        load_vec<T, D>(eta + (s * n + k) * D, x);
        if (p > 0) {
            T Ut[D][D], ux[D];
            load_mat<T, D, D>(ut + (s * (n - 1) + k) * D * D, Ut);
            gemv_t<T, D, D>(Ut, xp, ux);
            // chain: the emit kernel of the factorisation already left -U_k^T (the posterior transition) in ut
            MF_UNROLL for (int i = 0; i < D; ++i) x[i] = chain ? x[i] + ux[i] : x[i] - ux[i];
        }
        MF_UNROLL for (int i = 0; i < D; ++i) xp[i] = x[i];
        const long ci = chain ? (k == 0 ? s : B + s * (n - 1) + k - 1) : s * n + k;
        if (chain) {
            // m_k = Delta_k^-1 x_k = C (C^T x) with C = chol(Delta_k^-1), written by the emit kernel of the factorisation
            T C[D][D], u[D], mk[D];
            load_lower<T, D>(chol_dinv + ci * D * D, C);
            trimulT_lower_vec<T, D>(C, x, u);
            trimul_lower_vec<T, D>(C, u, mk);
            store_vec<T, D>(m_post + ci * D, mk);
            continue;
        }
        T Lp[D][D], Lpi[D], Linv[D][D], Q[D][D], Qi[D];
        load_lower<T, D>(chol_d + (s * n + k) * D * D, Lp);
        LogAcc<T> lb;
        lb.init();
        tri_inv_lower<T, D>(Lp, Linv, lb, bad);
        MF_UNROLL for (int i = 0; i < D; ++i) Lpi[i] = Linv[i][i];
        trsv_lower<T, D>(Lp, Lpi, x);
        trsv_lower_t<T, D>(Lp, Lpi, x);                    // m_k = Delta_k^-1 x_k
        store_vec<T, D>(m_post + ci * D, x);
        trimulT_self_lower<T, D>(Linv, Q);
        lb.init();
        chol_lower<T, D>(Q, Qi, lb, bad);
        store_lower<T, D>(chol_dinv + ci * D * D, Q);
    }
    if (bad && info) raise_info(info);
}

}  // namespace mf
