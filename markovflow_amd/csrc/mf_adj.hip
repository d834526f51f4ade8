// Reverse mode through SymmetricBlockTriDiagonal.cholesky and LowerTriangularBlockTriDiagonal.block_diagonal_of_inverse for
// 10 <= d <= 32 (banded_matrices registers gradients for cholesky_band / inverse_from_cholesky_band, block_tri_diag.py:22-31; the
// reference differentiates them at d = 30, T = 1001, tests/unit/test_ssm_gaussian_transformations.py:40-46).
//
// Both adjoints are recurrences along the chain whose step is a dozen d x d products and one triangular inversion - sequential in
// time whatever the formulation (the local adjoint of a dense Cholesky is a general linear map on d^2 numbers: it does not compose
// as a scan of d x d matrices).  Rounds 4-5 ran them for d > 9 as a Python loop over the T blocks, ~10 torch launches per block
// (10^4 launches at the reference's shape).  Here: ONE launch per adjoint, one WAVEFRONT per series walking its chain with every
// matrix of the step in registers as 16 x 16 MFMA tiles (mf_wave.hpp's toolkit: products in the P^T Q form, the triangular
// inversion on DPP, transposes through a wave-private LDS image, no barrier).  Every product below is arranged so that the left
// factor is available transposed - either loaded transposed from memory (load_mat_t: the same cache lines) or formed transposed by
// a second product - because a register tile can only enter a product as P in P^T Q.  The first version of this file kept the
// matrices in LDS with scalar multiply-adds in a 256-thread workgroup: 42 us per block per adjoint at d = 30 (LDS latency and
// bandwidth, ~15 barriers per block) against 14 us here, 3.7 us at d <= 16; profiles/r06_adjoints.txt has the sequence.  What is
// left at f64, d > 16 is half matrix-core time (~310 v_mfma_f64_16x16x4 of 64 cycles per block on ONE SIMD - the series are the only
// parallelism) and half the dependent chain of a step (inversion, transposes, loads).
// (d <= 9 has the register kernels with a scan in time, mf_btd_par.hpp.)
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mf_launch.hpp"
#include "mf_wave.hpp"

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

}  // namespace wv

bool adj_covers(int d) { return d >= 10 && d <= 32; }

template <typename T>
int adj_cholesky_grad(long B, long n, int d, const T* ldiag, const T* lsub, const T* g_ldiag, const T* g_lsub, T* g_diag, T* g_sub,
                      hipStream_t st) {
    if (!adj_covers(d)) return -100;
    if (n > 1 && !lsub) return -100;
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
                             T* g_ldiag, T* g_lsub, hipStream_t st) {
    if (!adj_covers(d)) return -100;
    if (n > 1 && !lsub) return -100;
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
                                       hipStream_t);
template int adj_cholesky_grad<float>(long, long, int, const float*, const float*, const float*, const float*, float*, float*,
                                      hipStream_t);
template int adj_diag_of_inverse_grad<double>(long, long, int, const double*, const double*, const double*, const double*,
                                              const double*, double*, double*, hipStream_t);
template int adj_diag_of_inverse_grad<float>(long, long, int, const float*, const float*, const float*, const float*, const float*,
                                             float*, float*, hipStream_t);

}  // namespace mf
