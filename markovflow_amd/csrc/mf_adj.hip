// Reverse mode through SymmetricBlockTriDiagonal.cholesky and LowerTriangularBlockTriDiagonal.block_diagonal_of_inverse for
// 10 <= d <= 32 (banded_matrices registers gradients for cholesky_band / inverse_from_cholesky_band, block_tri_diag.py:22-31; the
// reference differentiates them at d = 30, T = 1001, tests/unit/test_ssm_gaussian_transformations.py:40-46).
//
// Both adjoints are recurrences along the chain whose step is a handful of d x d products and triangular solves - sequential in
// time whatever the formulation (the local adjoint of a dense Cholesky is a general linear map on d^2 numbers: it does not compose
// as a scan of d x d matrices).  Rounds 4-5 ran them for d > 9 as a Python loop over the T blocks, ~10 torch launches per block
// (10^4 launches at the reference's shape).  Here: ONE launch, one 256-thread workgroup per series walking its chain with every
// matrix of the step in LDS (32 x 33 images), four outputs per thread in the products, the triangular solves as products with the
// explicit inverse (one forward substitution per block, a thread per column).  No matrix cores: at these sizes a step is ~15
// barriers and a few hundred multiply-adds per thread, and the series are independent - the kernel's job is to remove the
// launches (d <= 9 has the register kernels with a scan in time,
// mf_btd_par.hpp).
#include <hip/hip_runtime.h>

#include "mf_launch.hpp"

namespace mf {
namespace adj {

constexpr int MAXD = 32, LD = MAXD + 1, MSZ = MAXD * LD, NTH = 256;

// C(i, j) = alpha sum_k A(i, k) B(k, j) + beta C(i, j); element (i, j) of X at X[i * ri + j * rj] (transposes are strides).
// Thread (i, g) = (tid / 8, tid % 8) forms the four outputs C(i, 4 g .. 4 g + 3): 32 x 8 threads cover d <= 32; one read of A and four of
// B per step of k, four independent accumulators, the loop unrolled so that the LDS reads of later steps are in flight (the first
// version - one output per loop, `for k` not unrolled - waited out an LDS round trip per multiply-add: 80 us per block at d = 30).
template <typename T>
__device__ __forceinline__ void mm(int d, T* C, const T* A, int ai, int ak, const T* B, int bk, int bj, T alpha, T beta) {
    const int i = threadIdx.x >> 3, j0 = 4 * (threadIdx.x & 7);
    if (i < d && j0 < d) {
        T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        const T* Ar = A + i * ai;
        const T* Bc = B + j0 * bj;
#pragma unroll 6
        for (int k = 0; k < d; ++k) {
            const T av = Ar[k * ak];
            const T* bp = Bc + k * bk;
            a0 += av * bp[0];           // (columns beyond d inside the 32 x 33 image: formed, never stored)
            a1 += av * bp[bj];
            a2 += av * bp[2 * bj];
            a3 += av * bp[3 * bj];
        }
        T* c = C + i * LD + j0;
        const T r[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (j0 + q < d) c[q] = alpha * r[q] + (beta == T(0) ? T(0) : beta * c[q]);
    }
}
// Li <- L^-1 (L lower): forward substitution on the identity, a thread per column (its column of Li is written and re-read by the
// same thread: LDS operations of a wavefront are in order)
template <typename T> __device__ __forceinline__ void tri_inverse(int d, const T* L, T* Li) {
    const int j = threadIdx.x;
    if (j < d) {
        for (int i = 0; i < j; ++i) Li[i * LD + j] = T(0);
        for (int i = j; i < d; ++i) {
            T v0 = i == j ? T(1) : T(0), v1 = T(0);
            int k = j;
#pragma unroll 4
            for (; k + 1 < i; k += 2) {
                v0 -= L[i * LD + k] * Li[k * LD + j];
                v1 -= L[i * LD + k + 1] * Li[(k + 1) * LD + j];
            }
            if (k < i) v0 -= L[i * LD + k] * Li[k * LD + j];
            Li[i * LD + j] = (v0 + v1) / L[i * LD + i];
        }
    }
}
template <typename T> __device__ __forceinline__ void load(int d, T* dst, const T* __restrict__ g, bool lower) {
    for (int e = threadIdx.x; e < d * d; e += NTH) {
        const int i = e / d, j = e - i * d;
        dst[i * LD + j] = (g && (!lower || j <= i)) ? g[e] : T(0);
    }
}
template <typename T> __device__ __forceinline__ void store(int d, T* __restrict__ g, const T* src, bool lower) {
    for (int e = threadIdx.x; e < d * d; e += NTH) {
        const int i = e / d, j = e - i * d;
        g[e] = (!lower || j <= i) ? src[i * LD + j] : T(0);
    }
}

// Adjoint of L = chol(P), P symmetric:  Pbar = sym(L^-T Phi(L^T Lbar) L^-1), Phi = lower triangle with the diagonal halved.
// In: L, Li = L^-1, Lbar (lower).  Out: Pbar in X.  tmp: scratch.  Ends with a barrier.
template <typename T> __device__ __forceinline__ void chol_adjoint(int d, const T* L, const T* Li, const T* Lbar, T* X, T* tmp) {
    mm<T>(d, tmp, L, 1, LD, Lbar, LD, 1, T(1), T(0));                     // L^T Lbar
    __syncthreads();
    for (int e = threadIdx.x; e < d * d; e += NTH) {
        const int i = e / d, j = e - i * d;
        X[i * LD + j] = j < i ? tmp[i * LD + j] : (j == i ? T(0.5) * tmp[i * LD + j] : T(0));
    }
    __syncthreads();
    mm<T>(d, tmp, Li, 1, LD, X, LD, 1, T(1), T(0));                       // Li^T Phi
    __syncthreads();
    mm<T>(d, X, tmp, LD, 1, Li, LD, 1, T(1), T(0));                       // ... Li
    __syncthreads();
    for (int e = threadIdx.x; e < d * d; e += NTH) {
        const int i = e / d, j = e - i * d;
        tmp[i * LD + j] = T(0.5) * (X[i * LD + j] + X[j * LD + i]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < d * d; e += NTH) {
        const int i = e / d, j = e - i * d;
        X[i * LD + j] = tmp[i * LD + j];
    }
    __syncthreads();
}

// mf_btd_cholesky_grad for 10 <= d <= 32: (g_ldiag lower | NULL, g_lsub | NULL) -> (g_diag symmetric, g_sub).  The backward sweep
// of _autograd_ops._cholesky_backward_torch:  Pbar_k = adj(L_k, Lbar_k);  Wbar = g_lsub_{k-1} - 2 Pbar_k W;  Sbar = Wbar L_{k-1}^-1;
// Lbar_{k-1} -= tril(Sbar^T W).  The triangular solves are products with the explicit inverse, formed once per block.
template <typename T>
__global__ void __launch_bounds__(NTH) chol_grad_kernel(long n, int d, const T* __restrict__ ldiag, const T* __restrict__ lsub,
                                                       const T* __restrict__ g_ldiag, const T* __restrict__ g_lsub,
                                                       T* __restrict__ g_diag, T* __restrict__ g_sub) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* base = reinterpret_cast<T*>(smem_raw);
    T *Lbar = base + 2 * MSZ, *X = base + 3 * MSZ, *tmp = base + 4 * MSZ, *W = base + 5 * MSZ, *corr = base + 6 * MSZ, *t2 = base + 7 * MSZ;
    T* Lbuf[2] = {base, base + 8 * MSZ};          // L_k / L_{k-1} and their inverses, swapped from block to block
    T* Libuf[2] = {base + MSZ, base + 9 * MSZ};
    const long s = blockIdx.x, dd = (long)d * d;
    for (int e = threadIdx.x; e < MSZ; e += NTH) corr[e] = T(0);
    load<T>(d, Lbuf[0], ldiag + (s * n + n - 1) * dd, true);
    __syncthreads();
    tri_inverse<T>(d, Lbuf[0], Libuf[0]);
    __syncthreads();
    int cur = 0;
    for (long k = n - 1; k >= 0; --k) {
        T *L = Lbuf[cur], *Li = Libuf[cur], *Lp = Lbuf[cur ^ 1], *Lip = Libuf[cur ^ 1];
        load<T>(d, Lbar, g_ldiag ? g_ldiag + (s * n + k) * dd : nullptr, true);
        const bool prev = k > 0;
        if (prev) load<T>(d, Lp, ldiag + (s * n + k - 1) * dd, true);
        __syncthreads();
        for (int e = threadIdx.x; e < d * d; e += NTH) {
            const int i = e / d, j = e - i * d;
            if (j <= i) Lbar[i * LD + j] -= corr[i * LD + j];
        }
        if (prev) tri_inverse<T>(d, Lp, Lip);
        __syncthreads();
        chol_adjoint<T>(d, L, Li, Lbar, X, tmp);
        store<T>(d, g_diag + (s * n + k) * dd, X, false);
        if (lsub && prev) {
            load<T>(d, W, lsub + (s * (n - 1) + k - 1) * dd, false);
            load<T>(d, tmp, g_lsub ? g_lsub + (s * (n - 1) + k - 1) * dd : nullptr, false);
            __syncthreads();
            mm<T>(d, tmp, X, LD, 1, W, LD, 1, T(-2), T(1));               // Wbar = g_lsub - 2 Pbar W
            __syncthreads();
            mm<T>(d, t2, tmp, LD, 1, Lip, LD, 1, T(1), T(0));             // Sbar = Wbar L_{k-1}^-1
            __syncthreads();
            store<T>(d, g_sub + (s * (n - 1) + k - 1) * dd, t2, false);
            mm<T>(d, corr, t2, 1, LD, W, LD, 1, T(1), T(0));              // Sbar^T W (its lower triangle is what is used)
        }
        __syncthreads();
        cur ^= 1;
    }
}

// mf_btd_diag_of_inverse_grad for 10 <= d <= 32.  Forward (block Takahashi): Li_k = L_k^-1, base_k = Li_k^T Li_k, G_k = W_k Li_k,
// Sigma_k = base_k + G_k^T Sigma_{k+1} G_k, Sub_k = -Sigma_{k+1} G_k.  Reverse mode, forward in time with the accumulated
// Z_k = dSigma_k (g_diag_k + what block k - 1 sent):
//     Gbar_k = Sigma_{k+1} (G_k (Z_k + Z_k^T) - Subbar_k);   Z_{k+1} = g_diag_{k+1} + (G_k Z_k - Subbar_k) G_k^T;
//     Wbar_k = Gbar_k Li_k^T;   Libar_k = W_k^T Gbar_k + Li_k (Z_k + Z_k^T);   Lbar_k = -tril(Li_k^T Libar_k Li_k^T).
template <typename T>
__global__ void __launch_bounds__(NTH) inv_grad_kernel(long n, int d, const T* __restrict__ ldiag, const T* __restrict__ lsub,
                                                      const T* __restrict__ sigma, const T* __restrict__ g_diag,
                                                      const T* __restrict__ g_sub, T* __restrict__ g_ldiag, T* __restrict__ g_lsub) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* base = reinterpret_cast<T*>(smem_raw);
    T *Li = base, *Z = base + MSZ, *G = base + 2 * MSZ, *Gbar = base + 3 * MSZ, *W = base + 4 * MSZ, *Sg = base + 5 * MSZ,
      *t1 = base + 6 * MSZ, *t2 = base + 7 * MSZ, *Sb = base + 8 * MSZ;
    const long s = blockIdx.x, dd = (long)d * d;
    load<T>(d, Z, g_diag ? g_diag + (s * n) * dd : nullptr, false);
    __syncthreads();
    for (long k = 0; k < n; ++k) {
        const bool nxt = lsub && k + 1 < n;
        load<T>(d, t1, ldiag + (s * n + k) * dd, true);
        if (nxt) {
            load<T>(d, W, lsub + (s * (n - 1) + k) * dd, false);
            load<T>(d, Sg, sigma + (s * n + k + 1) * dd, false);
            load<T>(d, Sb, g_sub ? g_sub + (s * (n - 1) + k) * dd : nullptr, false);
        }
        for (int e = threadIdx.x; e < d * d; e += NTH) {          // Zs = Z + Z^T -> t2
            const int i = e / d, j = e - i * d;
            t2[i * LD + j] = Z[i * LD + j] + Z[j * LD + i];
        }
        __syncthreads();
        tri_inverse<T>(d, t1, Li);
        __syncthreads();
        mm<T>(d, t1, Li, LD, 1, t2, LD, 1, T(1), T(0));                            // Libar = Li Zs  [+ W^T Gbar below]
        if (nxt) {
            mm<T>(d, G, W, LD, 1, Li, LD, 1, T(1), T(0));                          // G = W Li
            __syncthreads();
            mm<T>(d, Gbar, G, LD, 1, t2, LD, 1, T(1), T(0));                       // G Zs
            __syncthreads();
            for (int e = threadIdx.x; e < d * d; e += NTH) {
                const int i = e / d, j = e - i * d;
                Gbar[i * LD + j] -= Sb[i * LD + j];
            }
            __syncthreads();
            mm<T>(d, t2, Sg, LD, 1, Gbar, LD, 1, T(1), T(0));                      // t2 = Gbar (final) = Sigma_{k+1} (G Zs - Subbar)
            __syncthreads();
            mm<T>(d, Gbar, t2, LD, 1, Li, 1, LD, T(1), T(0));                      // Wbar = Gbar Li^T
            mm<T>(d, t1, W, 1, LD, t2, LD, 1, T(1), T(1));                         // Libar += W^T Gbar
            __syncthreads();
            store<T>(d, g_lsub + (s * (n - 1) + k) * dd, Gbar, false);
            mm<T>(d, t2, G, LD, 1, Z, LD, 1, T(1), T(0));                          // G Z
            __syncthreads();
            for (int e = threadIdx.x; e < d * d; e += NTH) {
                const int i = e / d, j = e - i * d;
                t2[i * LD + j] -= Sb[i * LD + j];
            }
            load<T>(d, Z, g_diag ? g_diag + (s * n + k + 1) * dd : nullptr, false);
            __syncthreads();
            mm<T>(d, Z, t2, LD, 1, G, 1, LD, T(1), T(1));                          // Z_{k+1} = g_diag_{k+1} + (G Z - Subbar) G^T
        }
        __syncthreads();
        mm<T>(d, t2, Li, 1, LD, t1, LD, 1, T(1), T(0));                            // Li^T Libar
        __syncthreads();
        mm<T>(d, G, t2, LD, 1, Li, 1, LD, T(-1), T(0));                            // - ... Li^T
        __syncthreads();
        store<T>(d, g_ldiag + (s * n + k) * dd, G, true);
        __syncthreads();
    }
}

template <typename K> bool attr(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}
}  // namespace adj

bool adj_covers(int d) { return d >= 10 && d <= adj::MAXD; }

template <typename T>
int adj_cholesky_grad(long B, long n, int d, const T* ldiag, const T* lsub, const T* g_ldiag, const T* g_lsub, T* g_diag, T* g_sub,
                      hipStream_t st) {
    if (!adj_covers(d)) return -100;
    constexpr int bytes = 10 * adj::MSZ * (int)sizeof(T);
    static const bool ok = adj::attr(&adj::chol_grad_kernel<T>, bytes);
    if (!ok) return -1000;
    hipLaunchKernelGGL((adj::chol_grad_kernel<T>), dim3((unsigned)B), dim3(adj::NTH), bytes, st, n, d, ldiag, n > 1 ? lsub : nullptr,
                       g_ldiag, g_lsub, g_diag, g_sub);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
template <typename T>
int adj_diag_of_inverse_grad(long B, long n, int d, const T* ldiag, const T* lsub, const T* sigma, const T* g_diag, const T* g_sub,
                             T* g_ldiag, T* g_lsub, hipStream_t st) {
    if (!adj_covers(d)) return -100;
    constexpr int bytes = 9 * adj::MSZ * (int)sizeof(T);
    static const bool ok = adj::attr(&adj::inv_grad_kernel<T>, bytes);
    if (!ok) return -1000;
    hipLaunchKernelGGL((adj::inv_grad_kernel<T>), dim3((unsigned)B), dim3(adj::NTH), bytes, st, n, d, ldiag, n > 1 ? lsub : nullptr,
                       sigma, g_diag, g_sub, g_ldiag, g_lsub);
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
