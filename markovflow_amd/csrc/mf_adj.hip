// Reverse mode through SymmetricBlockTriDiagonal.cholesky and LowerTriangularBlockTriDiagonal.block_diagonal_of_inverse for
// 10 <= d <= 32 (banded_matrices registers gradients for cholesky_band / inverse_from_cholesky_band, block_tri_diag.py:22-31; the
// reference differentiates them at d = 30, T = 1001, tests/unit/test_ssm_gaussian_transformations.py:40-46).
//
// Both adjoints are recurrences along the chain whose step is a handful of d x d products and triangular solves - sequential in
// time whatever the formulation (the local adjoint of a dense Cholesky is a general linear map on d^2 numbers: it does not compose
// as a scan of d x d matrices).  Rounds 4-5 ran them for d > 9 as a Python loop over the T blocks, ~10 torch launches per block
// (10^4 launches at the reference's shape).  Here: ONE launch, one 256-thread workgroup per series walking its chain with every
// matrix of the step in LDS (row stride d + 1), a thread per output element in the products, a thread per column / row in the
// triangular solves.  No matrix cores: at these sizes a step is ~10 barriers and ~10^3 multiply-adds per thread, and the series are
// independent - the kernel's job is to remove the launches, not to race (d <= 9 has the register kernels with a scan in time,
// mf_btd_par.hpp).
#include <hip/hip_runtime.h>

#include "mf_launch.hpp"

namespace mf {
namespace adj {

constexpr int MAXD = 32, LD = MAXD + 1, MSZ = MAXD * LD, NTH = 256;

template <typename T> struct Ws {
    T* m[8];        // eight d x d matrices in LDS
};

// C(i, j) = alpha sum_k A(i, k) B(k, j) + beta C(i, j); element (i, j) of X at X[i * ri + j * rj] (transposes are strides)
template <typename T>
__device__ __forceinline__ void mm(int d, T* C, const T* A, int ai, int ak, const T* B, int bk, int bj, T alpha, T beta) {
    for (int e = threadIdx.x; e < d * d; e += NTH) {
        const int i = e / d, j = e - i * d;
        T acc = T(0);
        for (int k = 0; k < d; ++k) acc += A[i * ai + k * ak] * B[k * bk + j * bj];
        C[i * LD + j] = alpha * acc + (beta == T(0) ? T(0) : beta * C[i * LD + j]);
    }
}
// X <- L^-T X  (L lower): back substitution down the columns, a thread per column
template <typename T> __device__ __forceinline__ void solve_lt_left(int d, const T* L, T* X) {
    const int j = threadIdx.x;
    if (j < d) {
        for (int i = d - 1; i >= 0; --i) {
            T v = X[i * LD + j];
            for (int k = i + 1; k < d; ++k) v -= L[k * LD + i] * X[k * LD + j];
            X[i * LD + j] = v / L[i * LD + i];
        }
    }
}
// X <- X L^-1  (L lower): x L = r per row, from the last column down, a thread per row
template <typename T> __device__ __forceinline__ void solve_l_right(int d, const T* L, T* X) {
    const int i = threadIdx.x;
    if (i < d) {
        for (int j = d - 1; j >= 0; --j) {
            T v = X[i * LD + j];
            for (int k = j + 1; k < d; ++k) v -= X[i * LD + k] * L[k * LD + j];
            X[i * LD + j] = v / L[j * LD + j];
        }
    }
}
// X <- L^-1 X  (L lower): forward substitution, a thread per column
template <typename T> __device__ __forceinline__ void solve_l_left(int d, const T* L, T* X) {
    const int j = threadIdx.x;
    if (j < d) {
        for (int i = 0; i < d; ++i) {
            T v = X[i * LD + j];
            for (int k = 0; k < i; ++k) v -= L[i * LD + k] * X[k * LD + j];
            X[i * LD + j] = v / L[i * LD + i];
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
// In: L, Lbar (lower).  Out: Pbar in X.  tmp: scratch.  Barriers inside; ends with one.
template <typename T> __device__ __forceinline__ void chol_adjoint(int d, const T* L, const T* Lbar, T* X, T* tmp) {
    mm<T>(d, tmp, L, 1, LD, Lbar, LD, 1, T(1), T(0));                     // L^T Lbar
    __syncthreads();
    for (int e = threadIdx.x; e < d * d; e += NTH) {
        const int i = e / d, j = e - i * d;
        X[i * LD + j] = j < i ? tmp[i * LD + j] : (j == i ? T(0.5) * tmp[i * LD + j] : T(0));
    }
    __syncthreads();
    solve_lt_left<T>(d, L, X);
    __syncthreads();
    solve_l_right<T>(d, L, X);
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
// Lbar_{k-1} -= tril(Sbar^T W).
template <typename T>
__global__ void __launch_bounds__(NTH) chol_grad_kernel(long n, int d, const T* __restrict__ ldiag, const T* __restrict__ lsub,
                                                       const T* __restrict__ g_ldiag, const T* __restrict__ g_lsub,
                                                       T* __restrict__ g_diag, T* __restrict__ g_sub) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* base = reinterpret_cast<T*>(smem_raw);
    T *L = base, *Lbar = base + MSZ, *X = base + 2 * MSZ, *tmp = base + 3 * MSZ, *W = base + 4 * MSZ, *Lp = base + 5 * MSZ,
      *corr = base + 6 * MSZ;
    const long s = blockIdx.x, dd = (long)d * d;
    for (int e = threadIdx.x; e < MSZ; e += NTH) corr[e] = T(0);
    __syncthreads();
    for (long k = n - 1; k >= 0; --k) {
        load<T>(d, L, ldiag + (s * n + k) * dd, true);
        load<T>(d, Lbar, g_ldiag ? g_ldiag + (s * n + k) * dd : nullptr, true);
        __syncthreads();
        for (int e = threadIdx.x; e < d * d; e += NTH) {
            const int i = e / d, j = e - i * d;
            if (j <= i) Lbar[i * LD + j] -= corr[i * LD + j];
        }
        __syncthreads();
        chol_adjoint<T>(d, L, Lbar, X, tmp);
        store<T>(d, g_diag + (s * n + k) * dd, X, false);
        if (lsub && k > 0) {
            load<T>(d, W, lsub + (s * (n - 1) + k - 1) * dd, false);
            load<T>(d, tmp, g_lsub ? g_lsub + (s * (n - 1) + k - 1) * dd : nullptr, false);
            load<T>(d, Lp, ldiag + (s * n + k - 1) * dd, true);
            __syncthreads();
            mm<T>(d, tmp, X, LD, 1, W, LD, 1, T(-2), T(1));               // Wbar = g_lsub - 2 Pbar W
            __syncthreads();
            solve_l_right<T>(d, Lp, tmp);                                 // Sbar = Wbar L_{k-1}^-1
            __syncthreads();
            store<T>(d, g_sub + (s * (n - 1) + k - 1) * dd, tmp, false);
            mm<T>(d, corr, tmp, 1, LD, W, LD, 1, T(1), T(0));             // Sbar^T W (its lower triangle is what is used)
            __syncthreads();
        } else {
            __syncthreads();
        }
    }
}

// mf_btd_diag_of_inverse_grad for 10 <= d <= 32.  Forward (block Takahashi): Li_k = L_k^-1, base_k = Li_k^T Li_k, G_k = W_k Li_k,
// Sigma_k = base_k + G_k^T Sigma_{k+1} G_k, Sub_k = -Sigma_{k+1} G_k.  Reverse mode, forward in time with the accumulated
// Z_k = dSigma_k (g_diag_k + what block k - 1 sent):
//     Gbar_k = Sigma_{k+1} G_k (Z_k + Z_k^T) - Sigma_{k+1} Subbar_k;   Z_{k+1} = g_diag_{k+1} + G_k Z_k G_k^T - Subbar_k G_k^T;
//     Wbar_k = Gbar_k Li_k^T;   Libar_k = W_k^T Gbar_k + Li_k (Z_k + Z_k^T);   Lbar_k = -tril(Li_k^T Libar_k Li_k^T).
template <typename T>
__global__ void __launch_bounds__(NTH) inv_grad_kernel(long n, int d, const T* __restrict__ ldiag, const T* __restrict__ lsub,
                                                      const T* __restrict__ sigma, const T* __restrict__ g_diag,
                                                      const T* __restrict__ g_sub, T* __restrict__ g_ldiag, T* __restrict__ g_lsub) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* base = reinterpret_cast<T*>(smem_raw);
    T *Li = base, *Z = base + MSZ, *G = base + 2 * MSZ, *Gbar = base + 3 * MSZ, *W = base + 4 * MSZ, *Sg = base + 5 * MSZ,
      *t1 = base + 6 * MSZ, *t2 = base + 7 * MSZ;
    const long s = blockIdx.x, dd = (long)d * d;
    load<T>(d, Z, g_diag ? g_diag + (s * n) * dd : nullptr, false);
    __syncthreads();
    for (long k = 0; k < n; ++k) {
        const bool nxt = lsub && k + 1 < n;
        // Li = L_k^-1: forward substitution on the identity
        load<T>(d, t1, ldiag + (s * n + k) * dd, true);
        for (int e = threadIdx.x; e < d * d; e += NTH) {
            const int i = e / d, j = e - i * d;
            Li[i * LD + j] = i == j ? T(1) : T(0);
        }
        __syncthreads();
        solve_l_left<T>(d, t1, Li);
        __syncthreads();
        // Zs = Z + Z^T -> t2
        for (int e = threadIdx.x; e < d * d; e += NTH) {
            const int i = e / d, j = e - i * d;
            t2[i * LD + j] = Z[i * LD + j] + Z[j * LD + i];
        }
        __syncthreads();
        // Libar (-> t1) = Li Zs  [+ W^T Gbar below]
        mm<T>(d, t1, Li, LD, 1, t2, LD, 1, T(1), T(0));
        if (nxt) {
            load<T>(d, W, lsub + (s * (n - 1) + k) * dd, false);
            load<T>(d, Sg, sigma + (s * n + k + 1) * dd, false);
            __syncthreads();
            mm<T>(d, G, W, LD, 1, Li, LD, 1, T(1), T(0));                          // G = W Li
            __syncthreads();
            // Gbar = Sigma_{k+1} (G Zs - Subbar)
            mm<T>(d, Gbar, G, LD, 1, t2, LD, 1, T(1), T(0));                       // G Zs
            __syncthreads();
            if (g_sub) {
                for (int e = threadIdx.x; e < d * d; e += NTH) {
                    const int i = e / d, j = e - i * d;
                    Gbar[i * LD + j] -= g_sub[(s * (n - 1) + k) * dd + e];
                }
                __syncthreads();
            }
            mm<T>(d, t2, Sg, LD, 1, Gbar, LD, 1, T(1), T(0));                      // t2 = Gbar (final)
            __syncthreads();
            // Wbar = Gbar Li^T -> out;  Libar += W^T Gbar
            mm<T>(d, Gbar, t2, LD, 1, Li, 1, LD, T(1), T(0));
            mm<T>(d, t1, W, 1, LD, t2, LD, 1, T(1), T(1));
            __syncthreads();
            store<T>(d, g_lsub + (s * (n - 1) + k) * dd, Gbar, false);
            // Z_{k+1} = g_diag_{k+1} + G Z G^T - Subbar G^T
            mm<T>(d, Gbar, G, LD, 1, Z, LD, 1, T(1), T(0));                        // G Z
            __syncthreads();
            if (g_sub) {
                for (int e = threadIdx.x; e < d * d; e += NTH) {
                    const int i = e / d, j = e - i * d;
                    Gbar[i * LD + j] -= g_sub[(s * (n - 1) + k) * dd + e];
                }
                __syncthreads();
            }
            load<T>(d, Z, g_diag ? g_diag + (s * n + k + 1) * dd : nullptr, false);
            __syncthreads();
            mm<T>(d, Z, Gbar, LD, 1, G, 1, LD, T(1), T(1));                        // += (G Z - Subbar) G^T
        }
        __syncthreads();
        // Lbar = -tril(Li^T Libar Li^T)
        mm<T>(d, t2, Li, 1, LD, t1, LD, 1, T(1), T(0));
        __syncthreads();
        mm<T>(d, G, t2, LD, 1, Li, 1, LD, T(-1), T(0));
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
    constexpr int bytes = 7 * adj::MSZ * (int)sizeof(T);
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
    constexpr int bytes = 8 * adj::MSZ * (int)sizeof(T);
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
