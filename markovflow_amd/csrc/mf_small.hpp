// Register-resident small-matrix primitives for one lane = one (series, time-chunk) sub-problem.
//
// Everything here is fully unrolled over the compile-time state dimension D so that every matrix
// lives in VGPRs with static indexing (runtime-indexed arrays would go to scratch on gfx950).
// Symmetric matrices are kept in full D x D arrays but only the LOWER triangle (i >= j) is ever
// computed or read, so the compiler drops the upper half.
#pragma once
#include <hip/hip_runtime.h>

#define MF_DEV __device__ __forceinline__
// primitives that are plain arithmetic: also compiled for the host, where tests/host_sim builds the step functions of the
// streamed posterior kernels (mf_post_math.hpp) and runs them lane by lane against the oracle WITHOUT a GPU
#define MF_HD __host__ __device__ __forceinline__

namespace mf {
// The caller's `info` word (non-positive pivot): one int in DEVICE memory, LAPACK-style.  0: every pivot positive.  Otherwise the
// word is MF_INFO_TOP - f with f the FLAT INDEX (series x blocks per series + block) of a block whose elimination met a
// non-positive pivot - the kernels combine with an atomic max, so the SMALLEST such index over everything that raised survives -
// or 1 when the raising kernel cannot name the block (reduction levels, composite kernels).  Decoded by mf_info_flat_index().
// The host learns of it through a stream-ordered copy queued behind the kernel (markovflow_amd/_lib.py), never through a store of
// the kernel's own across the bus - that one could land after the stream's synchronisation had returned.
constexpr int MF_INFO_TOP = 0x7fffffff;
__device__ __forceinline__ void raise_info(int* info) { atomicMax(info, 1); }
__device__ __forceinline__ void raise_pivot(int* info, long flat) {
    long f = flat < 0 ? 0 : flat;
    if (f > (long)MF_INFO_TOP - 2) f = (long)MF_INFO_TOP - 2;
    atomicMax(info, MF_INFO_TOP - (int)f);
}
}   // namespace mf
#define MF_UNROLL _Pragma("unroll")

namespace mf {

template <typename T> MF_HD T t_sqrt(T x);
template <> MF_HD float t_sqrt<float>(float x) { return __builtin_sqrtf(x); }
template <> MF_HD double t_sqrt<double>(double x) { return __builtin_sqrt(x); }

// 1/x and 1/sqrt(x) from the hardware seed (v_rcp / v_rsq) plus one Newton step: ~1 ulp, a handful of
// dependent instructions.  The IEEE-exact expansions of `1/x` and `sqrt` are ~3x longer and sit on the
// critical path of every Cholesky pivot.
template <typename T> MF_HD T t_rcp(T x);
template <> MF_HD float t_rcp<float>(float x) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return 1.0f / x;
#else
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(r, __builtin_fmaf(-x, r, 1.0f), r);
#endif
}
template <> MF_HD double t_rcp<double>(double x) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return 1.0 / x;
#else
    const double r = __builtin_amdgcn_rcp(x);
    const double r1 = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    return __builtin_fma(r1, __builtin_fma(-x, r1, 1.0), r1);
#endif
}
template <typename T> MF_HD T t_rsqrt(T x);
template <> MF_HD float t_rsqrt<float>(float x) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return 1.0f / __builtin_sqrtf(x);
#else
    const float r = __builtin_amdgcn_rsqf(x);
    const float h = 0.5f * r;
    return __builtin_fmaf(h, __builtin_fmaf(-x * r, r, 1.0f), r);
#endif
}
template <> MF_HD double t_rsqrt<double>(double x) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return 1.0 / __builtin_sqrt(x);
#else
    const double r = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-x * r, r, 1.0);
    const double r1 = __builtin_fma(0.5 * r, e, r);
    const double e1 = __builtin_fma(-x * r1, r1, 1.0);
    return __builtin_fma(0.5 * r1, e1, r1);
#endif
}

// Running log-determinant accumulator (sum of log|x_i|).
// double: kept as (mantissa product, exponent sum) so the per-step cost is a few multiplies and one
//         frexp instead of 2*D software log evaluations; one log at the very end.
// float : v_log_f32 is a single instruction, so simply sum log2|x|.
template <typename T> struct LogAcc;
template <> struct LogAcc<double> {
    double mant;
    int expo;
    MF_HD void init() { mant = 1.0; expo = 0; }
    MF_HD void mul(double x) { mant *= x; }
    MF_HD void renorm() {
        int e;
        mant = frexp(mant, &e);
        expo += e;
    }
    MF_HD double value() const {
        const double m = mant < 0.0 ? -mant : mant;
        return log(m) + double(expo) * 0.6931471805599453094;
    }
};
template <> struct LogAcc<float> {
    float acc;
    MF_HD void init() { acc = 0.f; }
#if defined(__HIP_DEVICE_COMPILE__)
    MF_HD void mul(float x) { acc += __log2f(__builtin_fabsf(x)); }
#else
    MF_HD void mul(float x) { acc += log2f(__builtin_fabsf(x)); }
#endif
    MF_HD void renorm() {}
    MF_HD float value() const { return acc * 0.6931471805599453094f; }
};

// ---- loads / stores of contiguous blocks -------------------------------------------------------
template <typename T, int N> MF_HD void load_vec(const T* __restrict__ p, T (&v)[N]) {
    MF_UNROLL for (int i = 0; i < N; ++i) v[i] = p[i];
}
template <typename T, int N> MF_HD void store_vec(T* __restrict__ p, const T (&v)[N]) {
    MF_UNROLL for (int i = 0; i < N; ++i) p[i] = v[i];
}
template <typename T, int R, int C> MF_HD void load_mat(const T* __restrict__ p, T (&m)[R][C]) {
    MF_UNROLL for (int i = 0; i < R; ++i) MF_UNROLL for (int j = 0; j < C; ++j) m[i][j] = p[i * C + j];
}
template <typename T, int D> MF_HD void load_lower(const T* __restrict__ p, T (&m)[D][D]) {
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) m[i][j] = p[i * D + j];
}
template <typename T, int R, int C> MF_HD void store_mat(T* __restrict__ p, const T (&m)[R][C]) {
    MF_UNROLL for (int i = 0; i < R; ++i) MF_UNROLL for (int j = 0; j < C; ++j) p[i * C + j] = m[i][j];
}
// store a symmetric matrix held in its lower triangle as a full dense block
template <typename T, int D> MF_HD void store_sym(T* __restrict__ p, const T (&m)[D][D]) {
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j < D; ++j) p[i * D + j] = (i >= j) ? m[i][j] : m[j][i];
}
// store a lower-triangular matrix as a dense block with an explicit zero upper triangle
template <typename T, int D> MF_HD void store_lower(T* __restrict__ p, const T (&m)[D][D]) {
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j < D; ++j) p[i * D + j] = (i >= j) ? m[i][j] : T(0);
}

// ---- triangular / symmetric kernels --------------------------------------------------------------
// Loop order convention: the INNERMOST loop always runs over independent outputs (outer-product /
// right-looking forms), never along a dot product.  With one wavefront per SIMD there is no other wave
// to hide the latency of a dependent fp64 FMA, so instruction-level parallelism has to be in program
// order (under register pressure the scheduler stays close to it).

// Ci = C^-1 for lower-triangular C (only the lower triangle of C is read).  `la` picks up prod diag(C).
template <typename T, int D>
MF_HD void tri_inv_lower(const T (&C)[D][D], T (&Ci)[D][D], LogAcc<T>& la, bool& bad) {
    MF_UNROLL for (int i = 0; i < D; ++i) {
        const T c = C[i][i];
        bad |= !(c != T(0));
        la.mul(c);
        Ci[i][i] = t_rcp<T>(c);
    }
    // row i of C^-1:  Ci[i][j] = -Ci[i][i] * sum_{k=j}^{i-1} C[i][k] Ci[k][j]
    MF_UNROLL for (int i = 1; i < D; ++i) {
        T acc[D];
        MF_UNROLL for (int j = 0; j < i; ++j) acc[j] = T(0);
        MF_UNROLL for (int k = 0; k < i; ++k)
            MF_UNROLL for (int j = 0; j <= k; ++j) acc[j] += C[i][k] * Ci[k][j];
        MF_UNROLL for (int j = 0; j < i; ++j) Ci[i][j] = -acc[j] * Ci[i][i];
    }
}

// out = Lo * A   (Lo lower triangular)
template <typename T, int D, int N>
MF_HD void trimul_lower(const T (&Lo)[D][D], const T (&A)[D][N], T (&out)[D][N]) {
    MF_UNROLL for (int i = 0; i < D; ++i) {
        MF_UNROLL for (int j = 0; j < N; ++j) out[i][j] = Lo[i][0] * A[0][j];
        MF_UNROLL for (int k = 1; k <= i; ++k)
            MF_UNROLL for (int j = 0; j < N; ++j) out[i][j] += Lo[i][k] * A[k][j];
    }
}
// A <- Lo * A in place (row i only needs rows k <= i: bottom-up)
template <typename T, int D, int N>
MF_HD void trimul_lower_inplace(const T (&Lo)[D][D], T (&A)[D][N]) {
    MF_UNROLL for (int i = D - 1; i >= 0; --i) {
        MF_UNROLL for (int j = 0; j < N; ++j) A[i][j] *= Lo[i][i];
        MF_UNROLL for (int k = 0; k < i; ++k)
            MF_UNROLL for (int j = 0; j < N; ++j) A[i][j] += Lo[i][k] * A[k][j];
    }
}
template <typename T, int D>
MF_HD void trimul_lower_vec(const T (&Lo)[D][D], const T (&a)[D], T (&out)[D]) {
    MF_UNROLL for (int i = 0; i < D; ++i) out[i] = Lo[i][0] * a[0];
    MF_UNROLL for (int k = 1; k < D; ++k)
        MF_UNROLL for (int i = k; i < D; ++i) out[i] += Lo[i][k] * a[k];
}
// out = Lo^T * A
template <typename T, int D, int N>
MF_HD void trimulT_lower(const T (&Lo)[D][D], const T (&A)[D][N], T (&out)[D][N]) {
    MF_UNROLL for (int i = 0; i < D; ++i) {
        MF_UNROLL for (int j = 0; j < N; ++j) out[i][j] = Lo[i][i] * A[i][j];
        MF_UNROLL for (int k = i + 1; k < D; ++k)
            MF_UNROLL for (int j = 0; j < N; ++j) out[i][j] += Lo[k][i] * A[k][j];
    }
}
template <typename T, int D>
MF_HD void trimulT_lower_vec(const T (&Lo)[D][D], const T (&a)[D], T (&out)[D]) {
    MF_UNROLL for (int i = 0; i < D; ++i) out[i] = Lo[i][i] * a[i];
    MF_UNROLL for (int k = 1; k < D; ++k)
        MF_UNROLL for (int i = 0; i < k; ++i) out[i] += Lo[k][i] * a[k];
}
// Y <- -(Lo^T Y), in place (row i of the result only needs rows k >= i of Y)
template <typename T, int D, int N>
MF_HD void neg_trimulT_lower_inplace(const T (&Lo)[D][D], T (&Y)[D][N]) {
    MF_UNROLL for (int i = 0; i < D; ++i) {
        MF_UNROLL for (int j = 0; j < N; ++j) Y[i][j] *= -Lo[i][i];
        MF_UNROLL for (int k = i + 1; k < D; ++k)
            MF_UNROLL for (int j = 0; j < N; ++j) Y[i][j] -= Lo[k][i] * Y[k][j];
    }
}
// S(lower) = Lo^T Lo
template <typename T, int D> MF_HD void trimulT_self_lower(const T (&Lo)[D][D], T (&S)[D][D]) {
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) S[i][j] = Lo[D - 1][i] * Lo[D - 1][j];
    MF_UNROLL for (int k = D - 2; k >= 0; --k)
        MF_UNROLL for (int i = 0; i <= k; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) S[i][j] += Lo[k][i] * Lo[k][j];
}
// S(lower) += sign * B^T B
template <typename T, int D, int N>
MF_HD void syrk_tn_lower(const T (&B)[N][D], T (&S)[D][D], T sign) {
    MF_UNROLL for (int k = 0; k < N; ++k)
        MF_UNROLL for (int i = 0; i < D; ++i) {
            const T bi = sign * B[k][i];
            MF_UNROLL for (int j = 0; j <= i; ++j) S[i][j] += bi * B[k][j];
        }
}
// S(lower) += sign * W W^T
template <typename T, int D, int N>
MF_HD void syrk_nt_lower(const T (&W)[D][N], T (&S)[D][D], T sign) {
    MF_UNROLL for (int k = 0; k < N; ++k)
        MF_UNROLL for (int i = 0; i < D; ++i) {
            const T wi = sign * W[i][k];
            MF_UNROLL for (int j = 0; j <= i; ++j) S[i][j] += wi * W[j][k];
        }
}
// out = B^T v
template <typename T, int D, int N>
MF_HD void gemv_t(const T (&B)[N][D], const T (&v)[N], T (&out)[D]) {
    MF_UNROLL for (int i = 0; i < D; ++i) out[i] = B[0][i] * v[0];
    MF_UNROLL for (int k = 1; k < N; ++k)
        MF_UNROLL for (int i = 0; i < D; ++i) out[i] += B[k][i] * v[k];
}
// out = W v
template <typename T, int D, int N>
MF_HD void gemv_n(const T (&W)[D][N], const T (&v)[N], T (&out)[D]) {
    MF_UNROLL for (int i = 0; i < D; ++i) out[i] = W[i][0] * v[0];
    MF_UNROLL for (int k = 1; k < N; ++k)
        MF_UNROLL for (int i = 0; i < D; ++i) out[i] += W[i][k] * v[k];
}

// In-place lower Cholesky (right-looking) of the symmetric matrix held in the lower triangle of S.  On exit
// the lower triangle holds L and Li[i] = 1 / L[i][i].  `la` picks up prod diag(L); `bad` is set on a
// non-positive pivot (the result is then NaN, LAPACK info > 0 style).
template <typename T, int D>
MF_HD void chol_lower(T (&S)[D][D], T (&Li)[D], LogAcc<T>& la, bool& bad) {
    MF_UNROLL for (int j = 0; j < D; ++j) {
        const T s = S[j][j];
        bad |= !(s > T(0));
        const T inv = t_rsqrt<T>(s);
        const T l = s * inv;
        S[j][j] = l;
        Li[j] = inv;
        la.mul(l);
        MF_UNROLL for (int i = j + 1; i < D; ++i) S[i][j] *= inv;
        MF_UNROLL for (int i = j + 1; i < D; ++i)
            MF_UNROLL for (int k = j + 1; k <= i; ++k) S[i][k] -= S[i][j] * S[k][j];
    }
}

// z <- L^-1 z
template <typename T, int D>
MF_HD void trsv_lower(const T (&L)[D][D], const T (&Li)[D], T (&z)[D]) {
    MF_UNROLL for (int k = 0; k < D; ++k) {
        z[k] *= Li[k];
        MF_UNROLL for (int i = k + 1; i < D; ++i) z[i] -= L[i][k] * z[k];
    }
}
// z <- L^-T z
template <typename T, int D>
MF_HD void trsv_lower_t(const T (&L)[D][D], const T (&Li)[D], T (&z)[D]) {
    MF_UNROLL for (int k = D - 1; k >= 0; --k) {
        z[k] *= Li[k];
        MF_UNROLL for (int i = 0; i < k; ++i) z[i] -= L[k][i] * z[k];
    }
}
// X <- L^-1 X   (N columns)
template <typename T, int D, int N>
MF_HD void trsm_left_lower(const T (&L)[D][D], const T (&Li)[D], T (&X)[D][N]) {
    MF_UNROLL for (int k = 0; k < D; ++k) {
        MF_UNROLL for (int c = 0; c < N; ++c) X[k][c] *= Li[k];
        MF_UNROLL for (int i = k + 1; i < D; ++i)
            MF_UNROLL for (int c = 0; c < N; ++c) X[i][c] -= L[i][k] * X[k][c];
    }
}
// X <- L^-T X
template <typename T, int D, int N>
MF_HD void trsm_left_lower_t(const T (&L)[D][D], const T (&Li)[D], T (&X)[D][N]) {
    MF_UNROLL for (int k = D - 1; k >= 0; --k) {
        MF_UNROLL for (int c = 0; c < N; ++c) X[k][c] *= Li[k];
        MF_UNROLL for (int i = 0; i < k; ++i)
            MF_UNROLL for (int c = 0; c < N; ++c) X[i][c] -= L[k][i] * X[k][c];
    }
}
// Y <- Y L^-T   (each of the R rows y solves  L y^T = b^T)
template <typename T, int D, int R>
MF_HD void trsm_right_lower_t(const T (&L)[D][D], const T (&Li)[D], T (&Y)[R][D]) {
    MF_UNROLL for (int k = 0; k < D; ++k) {
        MF_UNROLL for (int r = 0; r < R; ++r) Y[r][k] *= Li[k];
        MF_UNROLL for (int j = k + 1; j < D; ++j)
            MF_UNROLL for (int r = 0; r < R; ++r) Y[r][j] -= Y[r][k] * L[j][k];
    }
}
// Y <- Y L^-1   (each row y solves  L^T y^T = b^T)
template <typename T, int D, int R>
MF_HD void trsm_right_lower(const T (&L)[D][D], const T (&Li)[D], T (&Y)[R][D]) {
    MF_UNROLL for (int k = D - 1; k >= 0; --k) {
        MF_UNROLL for (int r = 0; r < R; ++r) Y[r][k] *= Li[k];
        MF_UNROLL for (int j = 0; j < k; ++j)
            MF_UNROLL for (int r = 0; r < R; ++r) Y[r][j] -= Y[r][k] * L[k][j];
    }
}
// X <- -(W X), column by column in place
template <typename T, int D>
MF_HD void neg_mul_inplace(const T (&W)[D][D], T (&X)[D][D]) {
    MF_UNROLL for (int c = 0; c < D; ++c) {
        T col[D];
        MF_UNROLL for (int i = 0; i < D; ++i) col[i] = W[i][0] * X[0][c];
        MF_UNROLL for (int k = 1; k < D; ++k)
            MF_UNROLL for (int i = 0; i < D; ++i) col[i] += W[i][k] * X[k][c];
        MF_UNROLL for (int i = 0; i < D; ++i) X[i][c] = -col[i];
    }
}

template <typename T, int D> MF_HD T dot_self(const T (&z)[D]) {
    T s = T(0);
    MF_UNROLL for (int i = 0; i < D; ++i) s += z[i] * z[i];
    return s;
}

}  // namespace mf
