// Reverse mode through SymmetricBlockTriDiagonal.cholesky and LowerTriangularBlockTriDiagonal.block_diagonal_of_inverse for
// 10 <= d <= 32 (banded_matrices registers gradients for cholesky_band / inverse_from_cholesky_band, block_tri_diag.py:22-31; the
// reference differentiates them at d = 30, T = 1001, tests/unit/test_ssm_gaussian_transformations.py:40-46).
//
// Two forms, both on 16 x 16 MFMA register tiles (mf_wave.hpp's toolkit: products as P^T Q, the triangular inversion on DPP, transposes
// through a wave-private LDS image, no workgroup barrier):
//   * PARALLEL IN TIME (second half of this file; needs adj_grad_ws bytes): each adjoint = terms that are local in time, one
//     wavefront per (series, block), around ONE congruence recursion X_k = N_k + G_k^T X_{k+1} G_k (cholesky; backwards) or
//     X_{k+1} = N_{k+1} + G_k X_k G_k^T (inverse blocks; forwards) with G_k = W_k L_k^-1, which runs as composed maps per chunk, a pass
//     over the chunk ends and a walk per chunk - the derivation and the d <= 9 kernels are mf_btd_par.hpp's.  d = 30, T = 1001, fp64:
//     backward 0.83 ms at B = 4 (0.7 x the forward), 16 ms at B = 256 (1.5 x).
//   * SEQUENTIAL (first half; no workspace, short chains): one wavefront per series walks the block recurrences - a dozen products
//     and one inversion per block: 13.7 us per block per adjoint at d = 30 in fp64 (27 ms for one chain of 1001 blocks), 3.7 us at
//     d <= 16.  Every product is arranged so that the left factor is available transposed - loaded transposed (load_mat_t: the same
//     cache lines) or formed transposed by a second product - because a register tile can only enter a product as P in P^T Q.
// Rounds 4-5 ran both adjoints for d > 9 as a Python loop over the T blocks, ~10 torch launches per block (10^4 launches at the
// reference's shape); the first kernel form of round 6 kept the matrices in LDS with scalar multiply-adds (42 us per block);
// profiles/r06_adjoints.txt has the sequence.  (d <= 9: the register kernels with the same scans, mf_btd_par.hpp.)
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mf_launch.hpp"
#include "mf_wave.hpp"
#include "mf_wave_ops.hpp"

namespace mf {
namespace wv {

// keep the lower triangle, scale the diagonal (0.5: the Phi of the Cholesky adjoint; 1: tril)
template <typename T, int NT> MF_DEV void mask_lower(Mat<T, NT>& m, T diag_scale, const Lane& ln) {
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        MF_UNROLL for (int tj = 0; tj < NT; ++tj)
            MF_UNROLL for (int e = 0; e < 4; ++e) {
                const int i = 16 * ti + Tr<T>::row(ln.q, e), j = 16 * tj + ln.r;
                const T v = m.t[ti][tj][e];
                m.t[ti][tj][e] = j < i ? v : (j == i ? diag_scale * v : T(0));
            }
}
template <typename T, int NT> MF_DEV void axpy(Mat<T, NT>& y, T alpha, const Mat<T, NT>& x) {
    MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j) y.t[i][j] += alpha * x.t[i][j];
}
// the lane indices, opaque to the optimiser: the per-lane element offsets of a load / store (16 per matrix, d at run time) are
// loop-invariant, LLVM hoists them out of the loop over the blocks and keeps ~150 address registers alive across it (mf_panel.hpp,
// the same pitfall); behind this they are recomputed where they are used
MF_DEV Lane opq(const Lane& ln) {
    Lane l = ln;
    asm volatile("" : "+v"(l.r), "+v"(l.q));
    return l;
}
// a lower-triangular factor (identity on the padding) and, for NT = 2, its off-diagonal tile transposed (tri_inv_mat's inputs)
template <typename T, int NT>
MF_DEV void load_factor(Mat<T, NT>& C, typename Tr<T>::v4& c10t, const T* __restrict__ g, int d, const Lane& ln_) {
    const Lane ln = opq(ln_);
    load_mat<T, NT, S_LOWER>(C, g, d, true, true, ln);
    c10t = typename Tr<T>::v4{0, 0, 0, 0};
    if constexpr (NT == 2) load_tile_t<T>(c10t, g, d, 1, 0, ln);
}
template <typename T, int NT> MF_DEV void load_or_zero(Mat<T, NT>& m, const T* __restrict__ g, int d, bool transposed, const Lane& ln_) {
    const Lane ln = opq(ln_);
    if (!g) m.zero();
    else if (transposed) load_mat_t<T, NT>(m, g, d, ln);
    else load_mat<T, NT, S_FULL>(m, g, d, false, false, ln);
}

// One load per lane, 128 bytes apart: every cache line of a d x d matrix (d <= 32: at most 64 lines) is asked for by ONE instruction.
// The step's own loads are issued where the values are needed (a step keeps up to a dozen matrices live: asking early costs
// registers the f64 kernels do not have); the NEXT block's matrices are touched at the top of the step instead and their lines
// are in the cache when the loads come.  The value is summed into a register that is consumed once, after the loop.
template <typename T> MF_DEV T touch(const T* __restrict__ g, long dd) {
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    const long e = (long)lane * (128 / (long)sizeof(T));
    return g ? g[e < dd ? e : 0] : T(0);
}

// The inputs of a step of the Cholesky adjoint that come from memory: block j = k - 1's factor, sub-diagonal block and incoming
// gradients.  PF (every instantiation but f64 at NT = 2, whose step already holds 300-370 of the 512 registers): block k - 2's are
// asked for at the top of step k and wait in registers for a whole step; otherwise they are loaded where they are used, from lines
// the step before touched.
template <typename T, int NT> struct CholIn {
    Mat<T, NT> L, W, Wbt, Lb;
    typename Tr<T>::v4 c10t;
};
template <typename T> struct CholPtrs {
    const T *ldiag, *lsub, *g_ldiag, *g_lsub;
    long n, dd;
    int d;
};
template <typename T, int NT, int PART> MF_DEV void load_part(CholIn<T, NT>& in, const CholPtrs<T>& p, long s, long j, const Lane& ln) {
    if constexpr (PART == 0 || PART < 0) {
        load_or_zero<T, NT>(in.W, p.lsub + (s * (p.n - 1) + j) * p.dd, p.d, false, ln);
        load_or_zero<T, NT>(in.Wbt, p.g_lsub ? p.g_lsub + (s * (p.n - 1) + j) * p.dd : nullptr, p.d, true, ln);      // g_lsub^T
    }
    if constexpr (PART == 1 || PART < 0) load_factor<T, NT>(in.L, in.c10t, p.ldiag + (s * p.n + j) * p.dd, p.d, ln);
    if constexpr (PART == 2 || PART < 0) load_or_zero<T, NT>(in.Lb, p.g_ldiag ? p.g_ldiag + (s * p.n + j) * p.dd : nullptr, p.d, false, ln);
}

// mf_btd_cholesky_grad for 10 <= d <= 32: (g_ldiag lower | NULL, g_lsub | NULL) -> (g_diag symmetric, g_sub).  The backward sweep
// of _autograd_ops._cholesky_backward_torch, last block first:
//     Pbar_k = sym(L_k^-T Phi(L_k^T Lbar_k) L_k^-1)   (Phi: lower triangle, diagonal halved);
//     Wbar = g_lsub_{k-1} - 2 Pbar_k W_{k-1};   Sbar_{k-1} = Wbar L_{k-1}^-1;   Lbar_{k-1} = tril(g_ldiag_{k-1} - Sbar^T W).
template <typename T, int NT, bool PF>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
wave_chol_grad_kernel(long n, int d, const T* __restrict__ ldiag, const T* __restrict__ lsub, const T* __restrict__ g_ldiag,
                      const T* __restrict__ g_lsub, T* __restrict__ g_diag, T* __restrict__ g_sub) {
    using v4 = typename Tr<T>::v4;
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    const Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x, dd = (long)d * d;
    const CholPtrs<T> ptr{ldiag, lsub, g_ldiag, g_lsub, n, dd, d};
    LogAcc<T> la;
    la.init();
    bool bad = false;
    Mat<T, NT> L, Li, Lbar;
    {
        v4 c10t;
        load_factor<T, NT>(L, c10t, ldiag + (s * n + n - 1) * dd, d, ln);
        load_or_zero<T, NT>(Lbar, g_ldiag ? g_ldiag + (s * n + n - 1) * dd : nullptr, d, false, ln);
        mask_lower<T, NT>(Lbar, T(1), ln);
        tri_inv_mat<T, NT>(L, c10t, Li, lds, ln, la, bad);
    }
    CholIn<T, NT> cur;
    if (PF && n > 1) load_part<T, NT, -1>(cur, ptr, s, n - 2, ln);
    T pf = T(0);
    for (long k = n - 1; k >= 0; --k) {
        const bool prev = k > 0;
        CholIn<T, NT> nx;
        if constexpr (PF) {
            if (k > 1) load_part<T, NT, -1>(nx, ptr, s, k - 2, ln);
        } else if (k > 1) {                                                   // block k - 2's lines (block k - 1's: a step ago)
            pf += (touch<T>(ldiag + (s * n + k - 2) * dd, dd) + touch<T>(lsub + (s * (n - 1) + k - 2) * dd, dd)) +
                  (touch<T>(g_lsub ? g_lsub + (s * (n - 1) + k - 2) * dd : nullptr, dd) +
                   touch<T>(g_ldiag ? g_ldiag + (s * n + k - 2) * dd : nullptr, dd));
        }
        Mat<T, NT> M, Y;
        tn<T, NT, S_LOWER, S_LOWER, S_FULL, OP_SET>(M, L, Lbar);              // L^T Lbar
        mask_lower<T, NT>(M, T(0.5), ln);                                     // Phi
        {
            Mat<T, NT> Ft;
            tn<T, NT, S_LOWER, S_LOWER, S_FULL, OP_SET>(Ft, M, Li);           // Phi^T L^-1 = (L^-T Phi)^T
            tn<T, NT, S_FULL, S_LOWER, S_FULL, OP_SET>(Y, Ft, Li);            // L^-T Phi L^-1
        }
        transpose<T, NT, S_FULL>(M, Y, lds, ln);
        MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j) Y.t[i][j] = T(0.5) * (Y.t[i][j] + M.t[i][j]);
        store_mat<T, NT, false>(g_diag + (s * n + k) * dd, Y, d, lds, opq(ln));    // Pbar_k
        if (prev) {
            if constexpr (!PF) load_part<T, NT, 0>(cur, ptr, s, k - 1, ln);
            tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(M, cur.W, Y);           // W^T Pbar = (Pbar W)^T
            axpy<T, NT>(cur.Wbt, T(-2), M);                                   // Wbar^T
            if constexpr (!PF) load_part<T, NT, 1>(cur, ptr, s, k - 1, ln);
            L = cur.L;
            tri_inv_mat<T, NT>(L, cur.c10t, Li, lds, ln, la, bad);            // L_{k-1}^-1
            tn<T, NT, S_FULL, S_LOWER, S_FULL, OP_SET>(M, cur.Wbt, Li);       // Sbar = Wbar L_{k-1}^-1
            store_mat<T, NT, false>(g_sub + (s * (n - 1) + k - 1) * dd, M, d, lds, opq(ln));
            tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(Y, M, cur.W);           // Sbar^T W
            if constexpr (!PF) load_part<T, NT, 2>(cur, ptr, s, k - 1, ln);
            Lbar = cur.Lb;
            axpy<T, NT>(Lbar, T(-1), Y);
            mask_lower<T, NT>(Lbar, T(1), ln);
        }
        if constexpr (PF) {
            if (k > 1) cur = nx;
        }
    }
    if (pf == T(1.234567e-30) && threadIdx.x == 0) g_diag[s * n * dd] = pf;    // (keeps the touches alive; never true in practice)
    (void)bad;
}

// The inputs of a step of the inverse's adjoint: block k's factor and sub-diagonal block, the incoming gradient of the
// sub-diagonal block of the inverse, Sigma_{k+1} and the incoming gradient of Sigma_{k+1}.  PF as above.
template <typename T, int NT> struct InvIn {
    Mat<T, NT> L, W, Sb, Sg, Zn;
    typename Tr<T>::v4 c10t;
};
template <typename T> struct InvPtrs {
    const T *ldiag, *lsub, *sigma, *g_diag, *g_sub;
    long n, dd;
    int d;
};
template <typename T, int NT, int PART>
MF_DEV void load_part(InvIn<T, NT>& in, const InvPtrs<T>& p, long s, long k, bool nxt, const Lane& ln) {
    if constexpr (PART == 0 || PART < 0) load_factor<T, NT>(in.L, in.c10t, p.ldiag + (s * p.n + k) * p.dd, p.d, ln);
    if (!nxt) return;
    if constexpr (PART == 1 || PART < 0) {
        load_or_zero<T, NT>(in.W, p.lsub + (s * (p.n - 1) + k) * p.dd, p.d, false, ln);
        load_or_zero<T, NT>(in.Sb, p.g_sub ? p.g_sub + (s * (p.n - 1) + k) * p.dd : nullptr, p.d, false, ln);
    }
    if constexpr (PART == 2 || PART < 0) load_or_zero<T, NT>(in.Sg, p.sigma + (s * p.n + k + 1) * p.dd, p.d, false, ln);
    if constexpr (PART == 3 || PART < 0)
        load_or_zero<T, NT>(in.Zn, p.g_diag ? p.g_diag + (s * p.n + k + 1) * p.dd : nullptr, p.d, false, ln);
}

// mf_btd_diag_of_inverse_grad for 10 <= d <= 32.  Forward (block Takahashi): Li_k = L_k^-1, G_k = W_k Li_k,
// Sigma_k = Li_k^T Li_k + G_k^T Sigma_{k+1} G_k, Sub_k = -Sigma_{k+1} G_k.  Reverse mode, first block first, with the accumulated
// Z_k = dSigma_k (g_diag_k + what block k - 1 sent), Zs = Z + Z^T, H = G Zs - Subbar_k:
//     Gbar_k = Sigma_{k+1} H;   Z_{k+1} = g_diag_{k+1} + (G_k Z_k - Subbar_k) G_k^T;
//     Wbar_k = Gbar_k Li_k^T;   Libar_k = Li_k Zs + W_k^T Gbar_k;   Lbar_k = -tril(Li_k^T Libar_k Li_k^T).
// W^T and Subbar^T come through the LDS image (a transpose is ~30 LDS instructions; a second load of the same matrix is 16 global
// ones and, prefetched, a matrix of registers for a whole step).
template <typename T, int NT, bool PF>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
wave_inv_grad_kernel(long n, int d, const T* __restrict__ ldiag, const T* __restrict__ lsub, const T* __restrict__ sigma,
                     const T* __restrict__ g_diag, const T* __restrict__ g_sub, T* __restrict__ g_ldiag, T* __restrict__ g_lsub) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    const Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x, dd = (long)d * d;
    const InvPtrs<T> ptr{ldiag, lsub, sigma, g_diag, g_sub, n, dd, d};
    LogAcc<T> la;
    la.init();
    bool bad = false;
    Mat<T, NT> Z;
    load_or_zero<T, NT>(Z, g_diag ? g_diag + (s * n) * dd : nullptr, d, false, ln);
    InvIn<T, NT> cur;
    if constexpr (PF) load_part<T, NT, -1>(cur, ptr, s, 0, n > 1, ln);
    T pf = T(0);
    for (long k = 0; k < n; ++k) {
        const bool nxt = k + 1 < n;
        InvIn<T, NT> nx;
        if constexpr (PF) {
            if (nxt) load_part<T, NT, -1>(nx, ptr, s, k + 1, k + 2 < n, ln);
        } else if (nxt) {                                                           // block k + 1's lines
            pf += touch<T>(ldiag + (s * n + k + 1) * dd, dd);
            if (k + 2 < n)
                pf += (touch<T>(lsub + (s * (n - 1) + k + 1) * dd, dd) + touch<T>(g_sub ? g_sub + (s * (n - 1) + k + 1) * dd : nullptr, dd)) +
                      (touch<T>(sigma + (s * n + k + 2) * dd, dd) + touch<T>(g_diag ? g_diag + (s * n + k + 2) * dd : nullptr, dd));
        }
        Mat<T, NT> Li, LiT, Libar;
        if constexpr (!PF) load_part<T, NT, 0>(cur, ptr, s, k, nxt, ln);
        tri_inv_mat<T, NT>(cur.L, cur.c10t, Li, lds, ln, la, bad);
        LiT.zero();
        transpose<T, NT, S_LOWER>(LiT, Li, lds, ln);
        {
            Mat<T, NT> Zs;
            transpose<T, NT, S_FULL>(Zs, Z, lds, ln);
            axpy<T, NT>(Zs, T(1), Z);
            tn<T, NT, S_UPPER, S_FULL, S_FULL, OP_SET>(Libar, LiT, Zs);             // Li Zs
            if (nxt) {
                Mat<T, NT> Gt, H, Gbar, X;
                if constexpr (!PF) load_part<T, NT, 1>(cur, ptr, s, k, nxt, ln);
                transpose<T, NT, S_FULL>(X, cur.W, lds, ln);                        // W^T
                tn<T, NT, S_LOWER, S_FULL, S_FULL, OP_SET>(Gt, Li, X);              // G^T = Li^T W^T
                tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(H, Gt, Zs);               // G Zs
                axpy<T, NT>(H, T(-1), cur.Sb);                                      // H = G Zs - Subbar
                if constexpr (!PF) load_part<T, NT, 2>(cur, ptr, s, k, nxt, ln);
                tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(Gbar, cur.Sg, H);         // Gbar = Sigma H  (Sigma symmetric)
                tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(Zs, H, cur.Sg);           // Gbar^T = H^T Sigma
                tn<T, NT, S_FULL, S_FULL, S_FULL, OP_ADD>(Libar, cur.W, Gbar);      // Libar += W^T Gbar
                tn<T, NT, S_FULL, S_UPPER, S_FULL, OP_SET>(H, Zs, LiT);             // Wbar = Gbar Li^T
                store_mat<T, NT, false>(g_lsub + (s * (n - 1) + k) * dd, H, d, lds, opq(ln));
                tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(H, Z, Gt);                // (G Z)^T = Z^T G^T
                transpose<T, NT, S_FULL>(X, cur.Sb, lds, ln);
                axpy<T, NT>(H, T(-1), X);                                           // (G Z - Subbar)^T
                if constexpr (!PF) load_part<T, NT, 3>(cur, ptr, s, k, nxt, ln);
                Z = cur.Zn;
                tn<T, NT, S_FULL, S_FULL, S_FULL, OP_ADD>(Z, H, Gt);                // Z_{k+1} = g_diag_{k+1} + (G Z - Subbar) G^T
            }
        }
        {
            Mat<T, NT> Ut;
            tn<T, NT, S_FULL, S_LOWER, S_FULL, OP_SET>(Ut, Libar, Li);              // (Li^T Libar)^T
            tn<T, NT, S_FULL, S_UPPER, S_FULL, OP_NEG>(Libar, Ut, LiT);             // -(Li^T Libar) Li^T
            mask_lower<T, NT>(Libar, T(1), ln);
            store_mat<T, NT, false>(g_ldiag + (s * n + k) * dd, Libar, d, lds, opq(ln));
        }
        if constexpr (PF) {
            if (nxt) cur = nx;
        }
    }
    if (pf == T(1.234567e-30) && threadIdx.x == 0) g_ldiag[s * n * dd] = pf;      // (keeps the touches alive; never true in practice)
    (void)bad;
}


// =====================================================================================================================================
// Parallel in time.  The block Cholesky  P_k = D_k - S_{k-1} P_{k-1}^-1 S_{k-1}^T, L_k = chol(P_k), W_k = S_k L_k^-T  is a LOCAL map
// (P_k, S_k) -> (L_k, W_k) behind a Riccati-type recursion in P_k: its adjoint splits into a part that is local in time - the only
// place where the projection Phi of the dense Cholesky adjoint acts - and the adjoint of the recursion, a CONGRUENCE recursion with
// the coupling of the block Takahashi recursion, G_k = W_k L_k^-1 (mf_btd_par.hpp has the derivation and the d <= 9 kernels):
//     Sbar_k(loc) = Wbar_k L_k^-1,   Lbar_k(eff) = Lbar_k - tril(Sbar_k(loc)^T W_k),   C_k = sym(L_k^-T Phi(L_k^T Lbar_k(eff)) L_k^-1)
//     Dbar_k = Z_k,   Z_k = C_k + G_k^T Z_{k+1} G_k,   Sbar_k = Sbar_k(loc) - 2 Z_{k+1} G_k.
// So: one wavefront per (series, block) for the local terms, the congruence recursion as composed maps per chunk + a pass over the
// chunk ends + a walk per chunk (the scheme of wave_inv_up_kernel / wave_inv_boundary_kernel / wave_inverse_blocks_kernel,
// mf_wave_ops.hpp, with C and G read instead of formed from a factor), an axpy.  (The sequential kernels above: 10.7 ms per launch
// for ONE series of 1001 blocks at d = 30 - a wavefront walking the chain; they stay for short chains and without a workspace.)
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_chol_grad_local_kernel(long B, long n, int d, const T* __restrict__ ldiag,
                                                                 const T* __restrict__ lsub, const T* __restrict__ gl,
                                                                 const T* __restrict__ gw, T* __restrict__ oC, T* __restrict__ oG,
                                                                 T* __restrict__ oS) {
    using v4 = typename Tr<T>::v4;
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    const Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long id = blockIdx.x, s = id / n, k = id % n, dd = (long)d * d;
    LogAcc<T> la;
    la.init();
    bool bad = false;
    Mat<T, NT> L, Li, Lb;
    v4 c10t;
    load_factor<T, NT>(L, c10t, ldiag + id * dd, d, ln);
    load_or_zero<T, NT>(Lb, gl ? gl + id * dd : nullptr, d, false, ln);
    Mat<T, NT> Wt, W, Wbt;
    const bool has_w = lsub && k + 1 < n;
    const long ks = s * (n - 1) + k;
    if (has_w) {
        load_or_zero<T, NT>(Wt, lsub + ks * dd, d, true, ln);
        if (gw) {
            load_or_zero<T, NT>(W, lsub + ks * dd, d, false, ln);
            load_or_zero<T, NT>(Wbt, gw + ks * dd, d, true, ln);
        }
    }
    tri_inv_mat<T, NT>(L, c10t, Li, lds, ln, la, bad);
    mask_lower<T, NT>(Lb, T(1), ln);
    if (has_w) {
        Mat<T, NT> X;
        tn<T, NT, S_FULL, S_LOWER, S_FULL, OP_SET>(X, Wt, Li);                  // G = W L^-1
        store_mat<T, NT, false>(oG + ks * dd, X, d, lds, ln);
        if (gw) {
            tn<T, NT, S_FULL, S_LOWER, S_FULL, OP_SET>(X, Wbt, Li);             // Sbar(loc) = Wbar L^-1
            store_mat<T, NT, false>(oS + ks * dd, X, d, lds, ln);
            Mat<T, NT> Tm;
            tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(Tm, X, W);                // Sbar(loc)^T W
            mask_lower<T, NT>(Tm, T(1), ln);
            axpy<T, NT>(Lb, T(-1), Tm);
        } else {
            X.zero();
            store_mat<T, NT, false>(oS + ks * dd, X, d, lds, ln);
        }
    }
    Mat<T, NT> M, Y;
    tn<T, NT, S_LOWER, S_LOWER, S_FULL, OP_SET>(M, L, Lb);                      // L^T Lbar(eff)
    mask_lower<T, NT>(M, T(0.5), ln);                                           // Phi
    {
        Mat<T, NT> Ft;
        tn<T, NT, S_LOWER, S_LOWER, S_FULL, OP_SET>(Ft, M, Li);
        tn<T, NT, S_FULL, S_LOWER, S_FULL, OP_SET>(Y, Ft, Li);                  // L^-T Phi L^-1
    }
    transpose<T, NT, S_FULL>(M, Y, lds, ln);
    MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j) Y.t[i][j] = T(0.5) * (Y.t[i][j] + M.t[i][j]);
    store_mat<T, NT, false>(oC + id * dd, Y, d, lds, ln);
    (void)bad;
}

// The congruence recursion run backwards, X_k = N_k + G_k^T X_{k+1} G_k (X symmetric), with N [B, n, d, d] and G [B, n - 1, d, d] read
// from memory (a.diag, a.sub); a.o1 <- X_k, a.o2 <- -X_{k+1} G_k (or NULL).  Chunks count from the LAST block, as in the kernels these
// are modelled on; the pass over the chunk ends IS wave_inv_boundary_kernel.
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_cong_up_kernel(FactArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x / a.P, c = blockIdx.x % a.P, n = a.n;
    const long k_hi = n - 1 - c * a.L, k_lo = n - (c + 1) * a.L > 0 ? n - (c + 1) * a.L : 0;
    int d = a.d;
    const long dd = long(d) * d;
    Mat<T, NT> MT, N;
    identity_mat<T, NT>(MT, ln);
    N.zero();
    for (long k = k_hi; k >= k_lo; --k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        Mat<T, NT> G, C, X;
        G.zero();
        if (k + 1 < n) load_mat<T, NT, S_FULL>(G, a.sub + (s * (n - 1) + k) * dd, d, false, false, ln);
        load_mat<T, NT, S_FULL>(C, a.diag + (s * n + k) * dd, d, false, false, ln);
        phase();
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, G, MT);                           // (M G)^T = G^T M^T
        MT = X;
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, N, G);                            // N G
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(N, G, X);                            // G^T N G
        axpy<T, NT>(N, T(1), C);
    }
    const long id = s * a.P + c;
    store_mat<T, NT, false>(a.rDv + id * dd, MT, d, lds, ln);
    store_mat<T, NT, false>(a.rGU + id * dd, N, d, lds, ln);
}
template <typename T, int NT, bool PART>
__global__ void __launch_bounds__(64) wave_cong_walk_kernel(FactArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = PART ? blockIdx.x / a.P : blockIdx.x, c = PART ? blockIdx.x % a.P : 0, n = a.n;
    const long k_hi = PART ? n - 1 - c * a.L : n - 1;
    const long k_lo = PART ? (n - (c + 1) * a.L > 0 ? n - (c + 1) * a.L : 0) : 0;
    int d = a.d;
    const long dd = long(d) * d;
    Mat<T, NT> Sig, Gk, Ck;
    Sig.zero();
    if (PART && c > 0) load_mat<T, NT, S_FULL>(Sig, a.bSig + (s * a.P + c) * dd, d, false, false, ln);
    Gk.zero();
    if (k_hi + 1 < n) load_mat<T, NT, S_FULL>(Gk, a.sub + (s * (n - 1) + k_hi) * dd, d, false, false, ln);
    load_mat<T, NT, S_FULL>(Ck, a.diag + (s * n + k_hi) * dd, d, false, false, ln);
    for (long k = k_hi; k >= k_lo; --k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        const bool more = k > k_lo, coupled = k + 1 < n;
        Mat<T, NT> Gn, Cn;
        if (more) {
            load_mat<T, NT, S_FULL>(Gn, a.sub + (s * (n - 1) + k - 1) * dd, d, false, false, ln);
            load_mat<T, NT, S_FULL>(Cn, a.diag + (s * n + k - 1) * dd, d, false, false, ln);
        }
        phase();
        Mat<T, NT> Out = Ck;
        if (coupled) {
            Mat<T, NT> NSG;
            tn<T, NT, S_FULL, S_FULL, S_FULL, OP_NEG>(NSG, Sig, Gk);                   // -X_{k+1} G   (X symmetric)
            if (a.o2) store_mat<T, NT, false>(a.o2 + (s * (n - 1) + k) * dd, NSG, d, lds, ln);
            tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SUB>(Out, Gk, NSG);                   // + G^T X G
        }
        store_mat<T, NT, false>(a.o1 + (s * n + k) * dd, Out, d, lds, ln);
        Sig = Out;
        if (more) {
            Gk = Gn;
            Ck = Cn;
        }
    }
}
// ---- block_diagonal_of_inverse, reverse mode, parallel in time --------------------------------------------------------------------
// The block Takahashi recursion run backwards in reverse mode is the same congruence recursion run FORWARD with G instead of G^T:
//     A_0 = sym(Sigmabar_0),   A_{k+1} = sym(Sigmabar_{k+1}) - sym(subbar_k G_k^T) + G_k A_k G_k^T
// between two kernels that are local in time: `pre` forms G_k and the explicit terms Q, `post` turns the totals A_k into
//     Lbar_k = -2 tril(L_k^-T (L_k^-1 A_k L_k^-T)) - tril(G_k^T Wbar_k),   Wbar_k = (2 Sigma_{k+1} G_k A_k - Sigma_{k+1} subbar_k) L_k^-T
// (mf_btd_par.hpp: btd_inv_grad_pre_kernel / _post_kernel, d <= 9).
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_inv_grad_pre_kernel(long B, long n, int d, const T* __restrict__ ldiag, const T* __restrict__ lsub,
                                                              const T* __restrict__ gd, const T* __restrict__ gs, T* __restrict__ oQ,
                                                              T* __restrict__ oG) {
    using v4 = typename Tr<T>::v4;
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    const Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long id = blockIdx.x, s = id / n, k = id % n, dd = (long)d * d;
    auto sym_in = [&](long blk, Mat<T, NT>& Q) {
        Mat<T, NT> Qt;
        load_or_zero<T, NT>(Q, gd ? gd + blk * dd : nullptr, d, false, ln);
        load_or_zero<T, NT>(Qt, gd ? gd + blk * dd : nullptr, d, true, ln);
        MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j) Q.t[i][j] = T(0.5) * (Q.t[i][j] + Qt.t[i][j]);
    };
    if (k == 0 || !lsub) {
        Mat<T, NT> Q;
        sym_in(id, Q);
        store_mat<T, NT, false>(oQ + id * dd, Q, d, lds, ln);
    }
    if (!(lsub && k + 1 < n)) return;
    const long ks = s * (n - 1) + k;
    LogAcc<T> la;
    la.init();
    bool bad = false;
    Mat<T, NT> L, Li, Wt, G, Gt, Q;
    v4 c10t;
    load_factor<T, NT>(L, c10t, ldiag + id * dd, d, ln);
    load_or_zero<T, NT>(Wt, lsub + ks * dd, d, true, ln);
    tri_inv_mat<T, NT>(L, c10t, Li, lds, ln, la, bad);
    tn<T, NT, S_FULL, S_LOWER, S_FULL, OP_SET>(G, Wt, Li);                      // G = W L^-1
    store_mat<T, NT, false>(oG + ks * dd, G, d, lds, ln);
    sym_in(id + 1, Q);
    if (gs) {
        Mat<T, NT> Sbt, X, Xt;
        tn<T, NT, S_LOWER, S_FULL, S_FULL, OP_SET>(Gt, Li, Wt);                 // G^T
        load_or_zero<T, NT>(Sbt, gs + ks * dd, d, true, ln);
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, Sbt, Gt);                  // subbar G^T
        transpose<T, NT, S_FULL>(Xt, X, lds, ln);
        MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j) Q.t[i][j] -= T(0.5) * (X.t[i][j] + Xt.t[i][j]);
    }
    store_mat<T, NT, false>(oQ + (id + 1) * dd, Q, d, lds, ln);
    (void)bad;
}
// The congruence recursion run forwards, X_{k+1} = N_{k+1} + G_k X_k G_k^T, X_0 = N_0 (X symmetric), N [B, n, d, d] = a.diag and
// G [B, n - 1, d, d] = a.sub read from memory; a.o1 <- X_k.  Chunk c carries position c L to (c + 1) L (a.rDv: the chunk's M = prod G,
// a.rGU: its N, a.bSig: X at position c L) - the scheme of wave_marg_up_kernel / _boundary_kernel / wave_marginals_kernel.
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_cong_fwd_up_kernel(FactArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x / a.P, c = blockIdx.x % a.P, nt = a.n - 1;
    const long t_lo = c * a.L, t_hi = (c + 1) * a.L < nt ? (c + 1) * a.L : nt;
    int d = a.d;
    const long dd = long(d) * d;
    Mat<T, NT> M, N;
    identity_mat<T, NT>(M, ln);
    N.zero();
    for (long k = t_lo; k < t_hi; ++k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        Mat<T, NT> GT, Qn, X;
        load_mat_t<T, NT>(GT, a.sub + (s * nt + k) * dd, d, ln);
        load_mat<T, NT, S_FULL>(Qn, a.diag + (s * a.n + k + 1) * dd, d, false, false, ln);
        phase();
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, GT, M);                           // G M
        M = X;
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, N, GT);                           // N G^T
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(N, GT, X);                           // G N G^T
        axpy<T, NT>(N, T(1), Qn);
    }
    const long id = s * a.P + c;
    store_mat<T, NT, false>(a.rDv + id * dd, M, d, lds, ln);
    store_mat<T, NT, false>(a.rGU + id * dd, N, d, lds, ln);
}
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_cong_fwd_boundary_kernel(FactArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x;
    int d = a.d;
    const long dd = long(d) * d;
    Mat<T, NT> X;
    load_mat<T, NT, S_FULL>(X, a.diag + (s * a.n) * dd, d, false, false, ln);           // X_0 = N_0
    for (long c = 0; c + 1 < a.P; ++c) {
        const long id = s * a.P + c;
        Mat<T, NT> MT, N, Y;
        load_mat_t<T, NT>(MT, a.rDv + id * dd, d, ln);
        load_mat<T, NT, S_FULL>(N, a.rGU + id * dd, d, false, false, ln);
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(Y, X, MT);                           // X M^T
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, MT, Y);                           // M X M^T
        axpy<T, NT>(X, T(1), N);
        store_mat<T, NT, false>(a.bSig + (id + 1) * dd, X, d, lds, ln);
    }
}
template <typename T, int NT, bool PART>
__global__ void __launch_bounds__(64) wave_cong_fwd_walk_kernel(FactArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = PART ? blockIdx.x / a.P : blockIdx.x, c = PART ? blockIdx.x % a.P : 0, nt = a.n - 1;
    const long t_lo = PART ? c * a.L : 0, t_hi = PART ? ((c + 1) * a.L < nt ? (c + 1) * a.L : nt) : nt;
    int d = a.d;
    const long dd = long(d) * d;
    Mat<T, NT> X, GT, Qn;
    if (!PART || c == 0) {
        load_mat<T, NT, S_FULL>(X, a.diag + (s * a.n) * dd, d, false, false, ln);
        store_mat<T, NT, false>(a.o1 + (s * a.n) * dd, X, d, lds, ln);
    } else {
        load_mat<T, NT, S_FULL>(X, a.bSig + (s * a.P + c) * dd, d, false, false, ln);
    }
    if (t_hi > t_lo) {
        load_mat_t<T, NT>(GT, a.sub + (s * nt + t_lo) * dd, d, ln);
        load_mat<T, NT, S_FULL>(Qn, a.diag + (s * a.n + t_lo + 1) * dd, d, false, false, ln);
    }
    for (long k = t_lo; k < t_hi; ++k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        Mat<T, NT> GTn, Qnn, Y;
        const long kn = k + 1 < t_hi ? k + 1 : k;
        load_mat_t<T, NT>(GTn, a.sub + (s * nt + kn) * dd, d, ln);
        load_mat<T, NT, S_FULL>(Qnn, a.diag + (s * a.n + kn + 1) * dd, d, false, false, ln);
        phase();
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(Y, X, GT);                           // X G^T
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, GT, Y);                           // G X G^T
        axpy<T, NT>(X, T(1), Qn);
        store_mat<T, NT, false>(a.o1 + (s * a.n + k + 1) * dd, X, d, lds, ln);
        GT = GTn;
        Qn = Qnn;
    }
}
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_inv_grad_post_kernel(long B, long n, int d, const T* __restrict__ ldiag, const T* __restrict__ lsub,
                                                               const T* __restrict__ sig, const T* __restrict__ tot,
                                                               const T* __restrict__ gs, const T* __restrict__ Gk, T* __restrict__ g_ldiag,
                                                               T* __restrict__ g_lsub) {
    using v4 = typename Tr<T>::v4;
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    const Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long id = blockIdx.x, s = id / n, k = id % n, dd = (long)d * d;
    LogAcc<T> la;
    la.init();
    bool bad = false;
    Mat<T, NT> L, Li, LiT, A, Lb;
    v4 c10t;
    load_factor<T, NT>(L, c10t, ldiag + id * dd, d, ln);
    load_or_zero<T, NT>(A, tot + id * dd, d, false, ln);                        // symmetric, stored full
    tri_inv_mat<T, NT>(L, c10t, Li, lds, ln, la, bad);
    LiT.zero();
    transpose<T, NT, S_LOWER>(LiT, Li, lds, ln);
    {
        Mat<T, NT> U, M1;
        tn<T, NT, S_FULL, S_UPPER, S_FULL, OP_SET>(U, A, LiT);                  // A L^-T   (A symmetric)
        tn<T, NT, S_UPPER, S_FULL, S_FULL, OP_SET>(M1, LiT, U);                 // L^-1 A L^-T
        tn<T, NT, S_LOWER, S_FULL, S_FULL, OP_SET>(Lb, Li, M1);                 // L^-T (.)
        MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j) Lb.t[i][j] = T(-2) * Lb.t[i][j];
    }
    if (lsub && k + 1 < n) {
        const long ks = s * (n - 1) + k;
        Mat<T, NT> G, Gt, Sg, X, Kt, Wb;
        load_or_zero<T, NT>(G, Gk + ks * dd, d, false, ln);
        load_or_zero<T, NT>(Gt, Gk + ks * dd, d, true, ln);
        load_or_zero<T, NT>(Sg, sig + (id + 1) * dd, d, false, ln);             // Sigma_{k+1} (symmetric)
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, Gt, A);                    // G A
        {
            Mat<T, NT> Sb;
            load_or_zero<T, NT>(Sb, gs ? gs + ks * dd : nullptr, d, false, ln);
            MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j) X.t[i][j] = T(2) * X.t[i][j] - Sb.t[i][j];
        }
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(Kt, X, Sg);                   // (Sigma X)^T = X^T Sigma
        tn<T, NT, S_FULL, S_UPPER, S_FULL, OP_SET>(Wb, Kt, LiT);                // Wbar = Sigma X L^-T
        store_mat<T, NT, false>(g_lsub + ks * dd, Wb, d, lds, ln);
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SUB>(Lb, G, Wb);                   // - G^T Wbar
    }
    mask_lower<T, NT>(Lb, T(1), ln);
    store_mat<T, NT, false>(g_ldiag + id * dd, Lb, d, lds, ln);
    (void)bad;
}

template <typename T> __global__ void __launch_bounds__(256) adj_axpy_kernel(long cnt, T alpha, const T* __restrict__ x, T* __restrict__ y) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < cnt) y[i] += alpha * x[i];
}

}  // namespace wv

bool adj_covers(int d) { return d >= 10 && d <= 32; }

void wave_udl_partition(long B, long n, int d, int elem_size, long& P, long& L);      // mf_wave_inst.hip

// workspace of the parallel-in-time forms: three [B, n, d, d] arrays (local terms, couplings, scan by-products) + the chunk maps and
// the states at the chunk ends of the congruence scans
size_t adj_grad_ws(long B, long n, int d, int elem_size) {
    if (!adj_covers(d) || B < 1 || n < 1) return 0;
    long P = 1, L = n;
    wave_udl_partition(B, n, d < 16 ? 16 : d, elem_size, P, L);
    const size_t need = (3 * size_t(B) * n + 3 * size_t(B) * P) * size_t(d) * d * elem_size + 256;
    // (three arrays of the size of the factor: beyond 8 GiB - thousands of series, which fill the chip with a wavefront each - the
    // sequential form, which needs none, takes the call; the token says "covered")
    return need <= (size_t(8) << 30) ? need : 16;
}
namespace {
constexpr long ADJ_PAR_MIN_BLOCKS = 32;          // shorter chains: the sequential kernels
template <typename T> struct AdjWs {
    T *C, *G, *Zs, *rM, *rN, *bS;
    long P, L;
};
template <typename T> bool adj_carve(void* ws, size_t ws_bytes, long B, long n, int d, AdjWs<T>& w) {
    static const bool seq = getenv("MF_ADJ_SEQUENTIAL") != nullptr;       // (A/B switch)
    const size_t need = adj_grad_ws(B, n, d, (int)sizeof(T));
    if (seq || !ws || n < ADJ_PAR_MIN_BLOCKS || need <= 16 || ws_bytes < need) return false;
    wave_udl_partition(B, n, d < 16 ? 16 : d, (int)sizeof(T), w.P, w.L);
    const size_t blk = size_t(B) * n * d * d, red = size_t(B) * w.P * d * d;
    T* p = static_cast<T*>(ws);
    w.C = p; w.G = p + blk; w.Zs = p + 2 * blk;
    w.rM = p + 3 * blk; w.rN = w.rM + red; w.bS = w.rN + red;
    return true;
}
// X_k = N_k + G_k^T X_{k+1} G_k for every block (N = w.C, G = w.G), X -> o1, -X_{k+1} G_k -> o2
template <typename T, int NT> void cong_back(long B, long n, int d, const AdjWs<T>& w, T* o1, T* o2, hipStream_t st) {
    wv::FactArgs<T> a{B, n, d, w.C, w.G, o1, o2, nullptr, nullptr, nullptr, nullptr};
    a.P = w.P; a.L = w.L; a.rDv = w.rM; a.rGU = w.rN; a.bSig = w.bS;
    const dim3 chunks((unsigned)(B * w.P)), series((unsigned)B), block(64);
    if (w.P > 1) {
        hipLaunchKernelGGL((wv::wave_cong_up_kernel<T, NT>), chunks, block, 0, st, a);
        hipLaunchKernelGGL((wv::wave_inv_boundary_kernel<T, NT>), series, block, 0, st, a);
        hipLaunchKernelGGL((wv::wave_cong_walk_kernel<T, NT, true>), chunks, block, 0, st, a);
    } else {
        hipLaunchKernelGGL((wv::wave_cong_walk_kernel<T, NT, false>), series, block, 0, st, a);
    }
}
template <typename T, int NT>
void inv_grad_par(long B, long n, int d, const T* ldiag, const T* lsub, const T* sigma, const T* g_diag, const T* g_sub, T* g_ldiag,
                  T* g_lsub, const AdjWs<T>& w, hipStream_t st) {
    const dim3 blocks((unsigned)(B * n)), chunks((unsigned)(B * w.P)), series((unsigned)B), block(64);
    T *Q = w.C, *G = w.G, *A = w.Zs;
    hipLaunchKernelGGL((wv::wave_inv_grad_pre_kernel<T, NT>), blocks, block, 0, st, B, n, d, ldiag, lsub, g_diag, g_sub, Q, G);
    wv::FactArgs<T> a{B, n, d, Q, G, A, nullptr, nullptr, nullptr, nullptr, nullptr};
    // (the forward recursion runs over the n - 1 transitions: the chunks of the partition are chunks of transitions)
    a.P = w.P; a.L = w.L; a.rDv = w.rM; a.rGU = w.rN; a.bSig = w.bS;
    if (w.P > 1) {
        hipLaunchKernelGGL((wv::wave_cong_fwd_up_kernel<T, NT>), chunks, block, 0, st, a);
        hipLaunchKernelGGL((wv::wave_cong_fwd_boundary_kernel<T, NT>), series, block, 0, st, a);
        hipLaunchKernelGGL((wv::wave_cong_fwd_walk_kernel<T, NT, true>), chunks, block, 0, st, a);
    } else {
        hipLaunchKernelGGL((wv::wave_cong_fwd_walk_kernel<T, NT, false>), series, block, 0, st, a);
    }
    hipLaunchKernelGGL((wv::wave_inv_grad_post_kernel<T, NT>), blocks, block, 0, st, B, n, d, ldiag, lsub, sigma, static_cast<const T*>(A),
                       g_sub, static_cast<const T*>(G), g_ldiag, g_lsub);
}
template <typename T, int NT>
void chol_grad_par(long B, long n, int d, const T* ldiag, const T* lsub, const T* g_ldiag, const T* g_lsub, T* g_diag, T* g_sub,
                   const AdjWs<T>& w, hipStream_t st) {
    hipLaunchKernelGGL((wv::wave_chol_grad_local_kernel<T, NT>), dim3((unsigned)(B * n)), dim3(64), 0, st, B, n, d, ldiag, lsub, g_ldiag,
                       g_lsub, w.C, w.G, g_sub);
    cong_back<T, NT>(B, n, d, w, g_diag, w.Zs, st);
    const long cnt = B * (n - 1) * (long)d * d;
    hipLaunchKernelGGL((wv::adj_axpy_kernel<T>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, cnt, T(2), static_cast<const T*>(w.Zs),
                       g_sub);
}
}  // namespace

template <typename T>
int adj_cholesky_grad(long B, long n, int d, const T* ldiag, const T* lsub, const T* g_ldiag, const T* g_lsub, T* g_diag, T* g_sub,
                      void* ws, size_t ws_bytes, hipStream_t st) {
    if (!adj_covers(d)) return -100;
    if (n > 1 && !lsub) return -100;
    {
        AdjWs<T> w;
        if (n > 1 && adj_carve<T>(ws, ws_bytes, B, n, d, w)) {
            if (d <= 16) chol_grad_par<T, 1>(B, n, d, ldiag, lsub, g_ldiag, g_lsub, g_diag, g_sub, w, st);
            else chol_grad_par<T, 2>(B, n, d, ldiag, lsub, g_ldiag, g_lsub, g_diag, g_sub, w, st);
            return hipGetLastError() == hipSuccess ? 0 : -1000;
        }
    }
    const bool ph = getenv("MF_ADJ_PREFETCH") ? atoi(getenv("MF_ADJ_PREFETCH")) != 0 : !(sizeof(T) == 8 && d > 16);   // (the variable: A/B only)
#define MF_ADJ_LAUNCH(NT_, PH_)                                                                                                        \
    hipLaunchKernelGGL((wv::wave_chol_grad_kernel<T, NT_, PH_>), dim3((unsigned)B), dim3(64), 0, st, n, d, ldiag, lsub, g_ldiag, g_lsub, \
                       g_diag, g_sub)
    if (d <= 16) { if (ph) MF_ADJ_LAUNCH(1, true); else MF_ADJ_LAUNCH(1, false); }
    else { if (ph) MF_ADJ_LAUNCH(2, true); else MF_ADJ_LAUNCH(2, false); }
#undef MF_ADJ_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
template <typename T>
int adj_diag_of_inverse_grad(long B, long n, int d, const T* ldiag, const T* lsub, const T* sigma, const T* g_diag, const T* g_sub,
                             T* g_ldiag, T* g_lsub, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!adj_covers(d)) return -100;
    if (n > 1 && !lsub) return -100;
    {
        AdjWs<T> w;
        if (n > 1 && adj_carve<T>(ws, ws_bytes, B, n, d, w)) {
            if (d <= 16) inv_grad_par<T, 1>(B, n, d, ldiag, lsub, sigma, g_diag, g_sub, g_ldiag, g_lsub, w, st);
            else inv_grad_par<T, 2>(B, n, d, ldiag, lsub, sigma, g_diag, g_sub, g_ldiag, g_lsub, w, st);
            return hipGetLastError() == hipSuccess ? 0 : -1000;
        }
    }
    const bool ph = getenv("MF_ADJ_PREFETCH") ? atoi(getenv("MF_ADJ_PREFETCH")) != 0 : !(sizeof(T) == 8 && d > 16);
#define MF_ADJ_LAUNCH(NT_, PH_)                                                                                                         \
    hipLaunchKernelGGL((wv::wave_inv_grad_kernel<T, NT_, PH_>), dim3((unsigned)B), dim3(64), 0, st, n, d, ldiag, lsub, sigma, g_diag, g_sub, \
                       g_ldiag, g_lsub)
    if (d <= 16) { if (ph) MF_ADJ_LAUNCH(1, true); else MF_ADJ_LAUNCH(1, false); }
    else { if (ph) MF_ADJ_LAUNCH(2, true); else MF_ADJ_LAUNCH(2, false); }
#undef MF_ADJ_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
template int adj_cholesky_grad<double>(long, long, int, const double*, const double*, const double*, const double*, double*, double*,
                                       void*, size_t, hipStream_t);
template int adj_cholesky_grad<float>(long, long, int, const float*, const float*, const float*, const float*, float*, float*, void*,
                                      size_t, hipStream_t);
template int adj_diag_of_inverse_grad<double>(long, long, int, const double*, const double*, const double*, const double*,
                                              const double*, double*, double*, void*, size_t, hipStream_t);
template int adj_diag_of_inverse_grad<float>(long, long, int, const float*, const float*, const float*, const float*, const float*,
                                             float*, float*, void*, size_t, hipStream_t);

}  // namespace mf
