// SDE kernel -> state space model on the device: for a concatenation (block-diagonal state) of
// Matern-1/2, 3/2, 5/2 components, the transition matrices A_k = exp(F dt_k) in closed form and the Cholesky factors of
// the process covariances Q_k = Pinf - A_k Pinf A_k^T (+ jitter I), written straight into the [B, T-1, d, d] tensors
// StateSpaceModel takes.  Replaces, for these kernels, the TensorFlow graph of
//   markovflow/kernels/matern.py:66-86 (Matern12), :299-324,:343-356 (Matern32), :434-460,:485-501 (Matern52),
//   markovflow/kernels/sde_kernel.py:421-446 (transition_statistics: Q = Pinf - A Pinf A^T + jitter),
//   markovflow/kernels/sde_kernel.py:592-610,644-658 (ConcatKernel: block-diagonal A and Pinf),
//   markovflow/state_space_model.py:634-656 (cholesky_or_zero: an all-zero covariance passes through as zero).
// One lane per (series, transition); every component is at most 3 x 3, so everything is register resident.
#include "../../include/markovflow_amd.h"

// No fused-multiply-add contraction in this file: Q = Pinf - A Pinf A^T must come out EXACTLY zero for a zero time gap
// (the all-zero pass-through of cholesky_or_zero), which a contracted `P - fma(...)` breaks by one rounding error.
#pragma STDC FP_CONTRACT OFF

#include <hip/hip_runtime.h>

namespace {

constexpr int MAXC = 16;   // components per kernel
struct Spec {
    int ncomp, d;
    int order[MAXC];       // 1, 3 or 5  (Matern-order/2)
    int off[MAXC];         // first state index of the component
};

template <typename T> __device__ __forceinline__ T t_exp(T x);
template <> __device__ __forceinline__ float t_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ double t_exp<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T t_sqrt_(T x);
template <> __device__ __forceinline__ float t_sqrt_<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt_<double>(double x) { return sqrt(x); }

// Forward-mode derivative carrier with two tangents (d/d lam, d/d var): the gradient kernel below instantiates the SAME closed
// forms (Comp::build, comp_chol) with it, so the derivative cannot drift from the value code.
template <typename T> struct Dual2 {
    T v, a, b;
    __device__ __forceinline__ Dual2() {}
    __device__ __forceinline__ Dual2(T x) : v(x), a(T(0)), b(T(0)) {}
    __device__ __forceinline__ Dual2(T x, T da, T db) : v(x), a(da), b(db) {}
};
template <typename T> __device__ __forceinline__ Dual2<T> operator+(Dual2<T> x, Dual2<T> y) { return {x.v + y.v, x.a + y.a, x.b + y.b}; }
template <typename T> __device__ __forceinline__ Dual2<T> operator-(Dual2<T> x, Dual2<T> y) { return {x.v - y.v, x.a - y.a, x.b - y.b}; }
template <typename T> __device__ __forceinline__ Dual2<T> operator-(Dual2<T> x) { return {-x.v, -x.a, -x.b}; }
template <typename T> __device__ __forceinline__ Dual2<T> operator*(Dual2<T> x, Dual2<T> y) {
    return {x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b};
}
template <typename T> __device__ __forceinline__ Dual2<T> operator/(Dual2<T> x, Dual2<T> y) {
    const T r = T(1) / y.v, q = x.v * r;          // ONE division (an fp64 division is a dozen quarter-rate instructions)
    return {q, (x.a - q * y.a) * r, (x.b - q * y.b) * r};
}
template <typename T> __device__ __forceinline__ Dual2<T>& operator+=(Dual2<T>& x, Dual2<T> y) { x = x + y; return x; }
template <typename T> __device__ __forceinline__ Dual2<T>& operator-=(Dual2<T>& x, Dual2<T> y) { x = x - y; return x; }
template <typename T> __device__ __forceinline__ bool operator==(Dual2<T> x, Dual2<T> y) { return x.v == y.v; }
template <> __device__ __forceinline__ Dual2<float> t_exp<Dual2<float>>(Dual2<float> x) { const float e = expf(x.v); return {e, e * x.a, e * x.b}; }
template <> __device__ __forceinline__ Dual2<double> t_exp<Dual2<double>>(Dual2<double> x) { const double e = exp(x.v); return {e, e * x.a, e * x.b}; }
template <> __device__ __forceinline__ Dual2<float> t_sqrt_<Dual2<float>>(Dual2<float> x) {
    const float r = sqrtf(x.v), h = 0.5f / r;
    return {r, h * x.a, h * x.b};
}
template <> __device__ __forceinline__ Dual2<double> t_sqrt_<Dual2<double>>(Dual2<double> x) {
    const double r = sqrt(x.v), h = 0.5 / r;
    return {r, h * x.a, h * x.b};
}

// A (K x K) and Pinf (K x K) of one Matern component; lam = sqrt(order) / lengthscale
template <typename T, int K> struct Comp {
    T A[K][K], P[K][K];
    __device__ __forceinline__ void build(T lam, T var, T dt) {
        const T e = t_exp<T>(-lam * dt);
        for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) P[i][j] = T(0);
        if (K == 1) {
            A[0][0] = e;
            P[0][0] = var;
        } else if (K == 2) {
            // F = [[0, 1], [-lam^2, -2 lam]],  A = e^{-lam dt} (I + (F + lam I) dt)        (matern.py:316-320)
            A[0][0] = e * (T(1) + lam * dt);
            A[0][1] = e * dt;
            A[1][0] = -e * lam * lam * dt;
            A[1][1] = e * (T(1) - lam * dt);
            P[0][0] = var;
            P[1][1] = var * lam * lam;
        } else {
            // F = [[0,1,0],[0,0,1],[-lam^3,-3 lam^2,-3 lam]],  N = F + lam I is nilpotent: A = e^{-lam dt}(I + N dt + N^2 dt^2/2)
            const T l2 = lam * lam, l3 = l2 * lam;
            const T N[3][3] = {{lam, T(1), T(0)}, {T(0), lam, T(1)}, {-l3, -T(3) * l2, -T(2) * lam}};
            T N2[3][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    T a = T(0);
                    for (int l = 0; l < 3; ++l) a += N[i][l] * N[l][j];
                    N2[i][j] = a;
                }
            const T h = T(0.5) * dt * dt;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) A[i][j] = e * ((i == j ? T(1) : T(0)) + N[i][j] * dt + N2[i][j] * h);
            const T l23 = l2 / T(3);                                                        // matern.py:494-500
            P[0][0] = var;
            P[0][2] = -var * l23;
            P[2][0] = -var * l23;
            P[1][1] = var * l23;
            P[2][2] = var * l2 * l2;
        }
    }
};

// Q = Pinf - A Pinf A^T + jitter (symmetrised) and its lower Cholesky factor L (an exactly zero Q passes through as a zero factor:
// `zero`)
template <typename T, int K>
__device__ __forceinline__ void comp_chol(const Comp<T, K>& c, T jitter, T (&Q)[K][K], T (&L)[K][K], bool& zero, bool want_chol) {
    T AP[K][K];
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) {
            T a = T(0);
            for (int l = 0; l < K; ++l) a += c.A[i][l] * c.P[l][j];
            AP[i][j] = a;
        }
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) {
            T a = T(0);
            for (int l = 0; l < K; ++l) a += AP[i][l] * c.A[j][l];
            Q[i][j] = c.P[i][j] - a + (i == j ? jitter : T(0));
        }
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < i; ++j) { const T m = T(0.5) * (Q[i][j] + Q[j][i]); Q[i][j] = m; Q[j][i] = m; }
    zero = true;
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) { zero &= (Q[i][j] == T(0)); L[i][j] = T(0); }
    if (!want_chol || zero) return;
    for (int j = 0; j < K; ++j) {
        T s = Q[j][j];
        for (int l = 0; l < j; ++l) s -= L[j][l] * L[j][l];
        const T ljj = t_sqrt_<T>(s);
        L[j][j] = ljj;
        for (int i = j + 1; i < K; ++i) {
            T v = Q[i][j];
            for (int l = 0; l < j; ++l) v -= L[i][l] * L[j][l];
            L[i][j] = v / ljj;
        }
    }
}

// the component's blocks of ONE of A / chol(Q) / Q (which = 0 / 1 / 2) into a d x d image `blk`
template <typename T, int K>
__device__ __forceinline__ void emit(const Comp<T, K>& c, T jitter, int d, int off, int which, T* __restrict__ blk) {
    if (which == 0) {
        for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) blk[(off + i) * d + off + j] = c.A[i][j];
        return;
    }
    T Q[K][K], L[K][K];
    bool zero;
    comp_chol<T, K>(c, jitter, Q, L, zero, which == 1);
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) blk[(off + i) * d + off + j] = which == 2 ? Q[i][j] : L[i][j];
}

// One wavefront per 64 consecutive (series, transition) pairs: every lane builds its d x d block in an LDS slice (odd
// stride: conflict-free), then the wave writes the 64 blocks - contiguous in the [B, n, d, d] tensor - with coalesced
// stores.  Per-lane scattered 8-byte stores reached 0.6 TB/s; this form writes at several TB/s.
template <typename T>
__global__ void __launch_bounds__(64) matern_transitions_kernel(long B, long n, Spec sp, const T* __restrict__ lam,
                                                                const T* __restrict__ var, long hstride,
                                                                const T* __restrict__ dt, T jitter, T* __restrict__ A,
                                                                T* __restrict__ cholQ, T* __restrict__ Q) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* buf = reinterpret_cast<T*>(smem_raw);
    const long base = (long)blockIdx.x * 64, total = B * n;
    const long id = base + threadIdx.x;
    const bool valid = id < total;
    const int d = sp.d, dd = d * d, stride = dd | 1;
    const long s = valid ? id / n : 0;
    const T delta = valid ? dt[id] : T(1);
    T* mine = buf + threadIdx.x * stride;
    T* outs[3] = {A, cholQ, Q};
    for (int which = 0; which < 3; ++which) {
        T* dst = outs[which];
        if (!dst) continue;
        for (int e = 0; e < dd; ++e) mine[e] = T(0);
        for (int c = 0; c < sp.ncomp; ++c) {
            const T l = lam[s * hstride + c], v = var[s * hstride + c];
            if (sp.order[c] == 1) { Comp<T, 1> k; k.build(l, v, delta); emit<T, 1>(k, jitter, d, sp.off[c], which, mine); }
            else if (sp.order[c] == 3) { Comp<T, 2> k; k.build(l, v, delta); emit<T, 2>(k, jitter, d, sp.off[c], which, mine); }
            else { Comp<T, 3> k; k.build(l, v, delta); emit<T, 3>(k, jitter, d, sp.off[c], which, mine); }
        }
        __syncthreads();
        long nvalid = total - base;
        if (nvalid > 64) nvalid = 64;
        const long count = nvalid * dd;
        T* gdst = dst + base * dd;
        for (long e = threadIdx.x; e < count; e += 64) gdst[e] = buf[(e / dd) * stride + (e % dd)];
        __syncthreads();
    }
}

template <typename T>
int run(int64_t B, int64_t n, int ncomp, const int* orders, const T* lam, const T* var, int per_series, const T* dt, T jitter,
        T* A, T* cholQ, T* Q, void* stream) {
    if (B < 0) return -1;
    if (n < 0) return -2;
    if (ncomp < 1 || ncomp > MAXC) return -3;
    if (!orders) return -4;
    Spec sp;
    sp.ncomp = ncomp;
    int off = 0;
    for (int c = 0; c < ncomp; ++c) {
        if (orders[c] != 1 && orders[c] != 3 && orders[c] != 5) return -4;
        sp.order[c] = orders[c];
        sp.off[c] = off;
        off += (orders[c] + 1) / 2;
    }
    sp.d = off;
    if (B == 0 || n == 0) return 0;
    if (!lam) return -5;
    if (!var) return -6;
    if (!dt) return -8;
    if (!A) return -10;
    const long total = B * n;
    const size_t lds = size_t(64) * size_t((sp.d * sp.d) | 1) * sizeof(T);
    // one wave's staging image: past the 64 KB a kernel gets by default the limit is raised explicitly (gfx950: 160 KB per
    // workgroup); past that the state dimension is not supported by this kernel
    if (lds > size_t(160) * 1024) return -100;
    if (lds > size_t(64) * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&matern_transitions_kernel<T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return -100;
    hipLaunchKernelGGL((matern_transitions_kernel<T>), dim3((unsigned)((total + 63) / 64)), dim3(64), lds,
                       static_cast<hipStream_t>(stream), (long)B, (long)n, sp, lam, var, per_series ? (long)ncomp : 0L, dt,
                       jitter, A, cholQ, Q);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// ---- reverse mode of the generator: d/d(lam, var) of  sum_k <gA_k, A_k> + <gC_k, chol Q_k> ----------------------------------------------
// (the hyper-parameter gradient of GaussianProcessRegression.log_likelihood: the reference differentiates matern.py / sde_kernel.py
// :421-446 and the banded Cholesky through TensorFlow; here gA, gC come from the Fisher-identity backward of the log-likelihood.)
// One lane per (series, transition): the component's closed forms in forward mode (Dual2: tangents d/d lam, d/d var), contracted
// with the incoming gradients' diagonal blocks; out [B, n, ncomp, 2] - summed over the transitions by the caller (deterministic).
// (gA / gC point at the component's own block: dense tensors with row stride ld, or - packed - the K x K block row-major and the
// lower triangle of the factor's block row-major)
template <typename T, int K>
__device__ __forceinline__ void grad_component(T lam, T var, T dt, T jitter, int ld, bool packed, const T* __restrict__ gA,
                                               const T* __restrict__ gC, T& g_lam, T& g_var) {
    using Du = Dual2<T>;
    Comp<Du, K> c;
    c.build(Du(lam, T(1), T(0)), Du(var, T(0), T(1)), Du(dt));
    g_lam = T(0);
    g_var = T(0);
    if (gA) {
        for (int i = 0; i < K; ++i)
            for (int j = 0; j < K; ++j) {
                const T g = gA[i * (packed ? K : ld) + j];
                g_lam += g * c.A[i][j].a;
                g_var += g * c.A[i][j].b;
            }
    }
    if (gC) {
        Du Q[K][K], L[K][K];
        bool zero;
        comp_chol<Du, K>(c, Du(jitter), Q, L, zero, true);
        if (!zero) {
            for (int i = 0; i < K; ++i)
                for (int j = 0; j <= i; ++j) {
                    const T g = packed ? gC[i * (i + 1) / 2 + j] : gC[i * ld + j];
                    g_lam += g * L[i][j].a;
                    g_var += g * L[i][j].b;
                }
        }
    }
}
// packed = 0: gA, gC are the dense [B, n, d, d] tensors.  packed = 1: gA is ONE record per transition of `rec` elements -
// [the K x K block of g_A of every component | the lower triangle of the block of g_cholQ of every component], what
// mf_gpr_matern_loglik_grad writes (csrc/mf_gpr_grad.hpp) - and gC is ignored.
template <typename T>
__global__ void __launch_bounds__(64) matern_transitions_grad_kernel(long B, long n, Spec sp, const T* __restrict__ lam,
                                                                     const T* __restrict__ var, long hstride,
                                                                     const T* __restrict__ dt, T jitter, const T* __restrict__ gA,
                                                                     const T* __restrict__ gC, int packed, int rec,
                                                                     T* __restrict__ out) {
    const long id = (long)blockIdx.x * 64 + threadIdx.x;
    if (id >= B * n) return;
    const long s = id / n;
    const int d = sp.d;
    const T delta = dt[id];
    int na = 0;
    for (int c = 0; c < sp.ncomp; ++c) { const int k = (sp.order[c] + 1) / 2; na += k * k; }
    int pa = 0, pc = (int)(((na * sizeof(T) + 15) / 16) * 16 / sizeof(T));      // the factor's part starts on a 16-byte unit
    for (int c = 0; c < sp.ncomp; ++c) {
        const T l = lam[s * hstride + c], v = var[s * hstride + c];
        const int k = (sp.order[c] + 1) / 2, off = sp.off[c];
        const T* ga = packed ? gA + id * rec + pa : (gA ? gA + id * d * d + off * d + off : nullptr);
        const T* gc = packed ? gA + id * rec + pc : (gC ? gC + id * d * d + off * d + off : nullptr);
        pa += k * k;
        pc += k * (k + 1) / 2;
        T gl, gv;
        if (sp.order[c] == 1) grad_component<T, 1>(l, v, delta, jitter, d, packed != 0, ga, gc, gl, gv);
        else if (sp.order[c] == 3) grad_component<T, 2>(l, v, delta, jitter, d, packed != 0, ga, gc, gl, gv);
        else grad_component<T, 3>(l, v, delta, jitter, d, packed != 0, ga, gc, gl, gv);
        out[(id * sp.ncomp + c) * 2] = gl;
        out[(id * sp.ncomp + c) * 2 + 1] = gv;
    }
}

// The stationary prior's factor chol(Pinf + jitter), block by block: out[s, c, :] = <g_cholP0 block c, d chol / d (lam, var)>.
// (Comp::build with the transition zeroed: Q = Pinf + jitter.)
template <typename T, int K>
__device__ __forceinline__ void prior_component(T lam, T var, T jitter, int ld, const T* __restrict__ gC, T& g_lam, T& g_var) {
    using Du = Dual2<T>;
    Comp<Du, K> c;
    c.build(Du(lam, T(1), T(0)), Du(var, T(0), T(1)), Du(T(1)));
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) c.A[i][j] = Du(T(0));
    Du Q[K][K], L[K][K];
    bool zero;
    comp_chol<Du, K>(c, Du(jitter), Q, L, zero, true);
    g_lam = T(0);
    g_var = T(0);
    if (zero) return;
    for (int i = 0; i < K; ++i)
        for (int j = 0; j <= i; ++j) {
            const T g = gC[i * ld + j];
            g_lam += g * L[i][j].a;
            g_var += g * L[i][j].b;
        }
}
template <typename T>
__global__ void __launch_bounds__(64) matern_prior_grad_kernel(long B, Spec sp, const T* __restrict__ lam, const T* __restrict__ var,
                                                               long hstride, T jitter, const T* __restrict__ gC0, T* __restrict__ out) {
    const long s = (long)blockIdx.x * 64 + threadIdx.x;
    if (s >= B) return;
    const int d = sp.d;
    for (int c = 0; c < sp.ncomp; ++c) {
        const T l = lam[s * hstride + c], v = var[s * hstride + c];
        const int off = sp.off[c];
        const T* gc = gC0 + s * d * d + off * d + off;
        T gl, gv;
        if (sp.order[c] == 1) prior_component<T, 1>(l, v, jitter, d, gc, gl, gv);
        else if (sp.order[c] == 3) prior_component<T, 2>(l, v, jitter, d, gc, gl, gv);
        else prior_component<T, 3>(l, v, jitter, d, gc, gl, gv);
        out[(s * sp.ncomp + c) * 2] = gl;
        out[(s * sp.ncomp + c) * 2 + 1] = gv;
    }
}
template <typename T>
int run_prior_grad(int64_t B, int ncomp, const int* orders, const T* lam, const T* var, int per_series, T jitter, const T* gC0, T* out,
                   void* stream) {
    if (B < 0) return -1;
    if (ncomp < 1 || ncomp > MAXC) return -2;
    if (!orders) return -3;
    Spec sp;
    sp.ncomp = ncomp;
    int off = 0;
    for (int c = 0; c < ncomp; ++c) {
        if (orders[c] != 1 && orders[c] != 3 && orders[c] != 5) return -3;
        sp.order[c] = orders[c];
        sp.off[c] = off;
        off += (orders[c] + 1) / 2;
    }
    sp.d = off;
    if (B == 0) return 0;
    if (!lam) return -4;
    if (!var) return -5;
    if (!gC0) return -8;
    if (!out) return -9;
    hipLaunchKernelGGL((matern_prior_grad_kernel<T>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream),
                       (long)B, sp, lam, var, per_series ? (long)ncomp : 0L, jitter, gC0, out);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T>
int run_grad(int64_t B, int64_t n, int ncomp, const int* orders, const T* lam, const T* var, int per_series, const T* dt, T jitter,
             const T* gA, const T* gC, int packed, T* out, void* stream) {
    if (B < 0) return -1;
    if (n < 0) return -2;
    if (ncomp < 1 || ncomp > MAXC) return -3;
    if (!orders) return -4;
    Spec sp;
    sp.ncomp = ncomp;
    int off = 0;
    for (int c = 0; c < ncomp; ++c) {
        if (orders[c] != 1 && orders[c] != 3 && orders[c] != 5) return -4;
        sp.order[c] = orders[c];
        sp.off[c] = off;
        off += (orders[c] + 1) / 2;
    }
    sp.d = off;
    if (B == 0 || n == 0) return 0;
    if (!lam) return -5;
    if (!var) return -6;
    if (!dt) return -8;
    if (!out) return -12;
    if (packed && !gA) return -10;
    int ra = 0, rc = 0;               // elements of a packed record: blocks of g_A | lower triangles of g_cholQ, each padded to 16 bytes
    for (int c = 0; c < ncomp; ++c) { const int k = (orders[c] + 1) / 2; ra += k * k; rc += k * (k + 1) / 2; }
    const int rec = (int)((((ra * sizeof(T) + 15) / 16) * 16 + ((rc * sizeof(T) + 15) / 16) * 16) / sizeof(T));
    const long total = B * n;
    hipLaunchKernelGGL((matern_transitions_grad_kernel<T>), dim3((unsigned)((total + 63) / 64)), dim3(64), 0,
                       static_cast<hipStream_t>(stream), (long)B, (long)n, sp, lam, var, per_series ? (long)ncomp : 0L, dt, jitter,
                       gA, gC, packed, rec, out);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// R^-1 = (L L^T)^-1 from the Cholesky factor of the observation covariance (KalmanFilter._r_inv, kalman_filter.py:341-348: a
// tf.linalg.cholesky_solve against the identity).  m <= 32: one wavefront, thread j solves column j of L^-1 by forward
// substitution, then the threads share the m^2 entries of L^-T L^-1.  ONE launch instead of the eleven small torch / rocBLAS
// kernels of an identity + two triangular solves (4.6 us each: 6 % of an evaluation at BASELINE config 4).
template <typename T>
__global__ void __launch_bounds__(64) obs_precision_kernel(int m, const T* __restrict__ chol, T* __restrict__ out, int* info) {
    __shared__ T L[32 * 33], Li[32 * 33];
    const int t = threadIdx.x;
    for (int e = t; e < m * m; e += 64) L[(e / m) * 33 + (e % m)] = chol[e];
    __syncthreads();
    if (t < m) {
        // column t of L^-1: x_t = 1 / L_tt, x_i = -(sum_{k<i} L_ik x_k) / L_ii for i > t
        for (int i = 0; i < m; ++i) {
            T acc = (i == t) ? T(1) : T(0);
            for (int k = t; k < i; ++k) acc -= L[i * 33 + k] * Li[k * 33 + t];
            Li[i * 33 + t] = i < t ? T(0) : acc / L[i * 33 + i];
        }
        if (!(L[t * 33 + t] != T(0)) && info) __hip_atomic_store(info, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    for (int e = t; e < m * m; e += 64) {
        const int i = e / m, j = e % m;
        T acc = T(0);
        for (int k = (i > j ? i : j); k < m; ++k) acc += Li[k * 33 + i] * Li[k * 33 + j];
        out[e] = acc;
    }
}

template <typename T> int obs_precision(int m, const T* chol, T* out, int* info, void* stream) {
    if (m < 1 || m > 32) return -1;
    if (!chol) return -2;
    if (!out) return -3;
    hipLaunchKernelGGL((obs_precision_kernel<T>), dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), m, chol, out, info);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

}  // namespace

extern "C" {

int mf_obs_precision_from_chol_f64(int m, const double* chol, double* out, int* info, void* stream) {
    return obs_precision<double>(m, chol, out, info, stream);
}
int mf_obs_precision_from_chol_f32(int m, const float* chol, float* out, int* info, void* stream) {
    return obs_precision<float>(m, chol, out, info, stream);
}

int mf_sde_matern_transitions_f64(int64_t B, int64_t n, int ncomp, const int* orders, const double* lam, const double* var,
                                  int per_series, const double* dt, double jitter, double* A, double* cholQ, double* Q,
                                  void* stream) {
    return run<double>(B, n, ncomp, orders, lam, var, per_series, dt, jitter, A, cholQ, Q, stream);
}
int mf_sde_matern_transitions_f32(int64_t B, int64_t n, int ncomp, const int* orders, const float* lam, const float* var,
                                  int per_series, const float* dt, float jitter, float* A, float* cholQ, float* Q,
                                  void* stream) {
    return run<float>(B, n, ncomp, orders, lam, var, per_series, dt, jitter, A, cholQ, Q, stream);
}

int mf_sde_matern_transitions_grad_f64(int64_t B, int64_t n, int ncomp, const int* orders, const double* lam, const double* var,
                                       int per_series, const double* dt, double jitter, const double* g_A, const double* g_cholQ,
                                       double* out, void* stream) {
    return run_grad<double>(B, n, ncomp, orders, lam, var, per_series, dt, jitter, g_A, g_cholQ, 0, out, stream);
}
int mf_sde_matern_transitions_grad_f32(int64_t B, int64_t n, int ncomp, const int* orders, const float* lam, const float* var,
                                       int per_series, const float* dt, float jitter, const float* g_A, const float* g_cholQ,
                                       float* out, void* stream) {
    return run_grad<float>(B, n, ncomp, orders, lam, var, per_series, dt, jitter, g_A, g_cholQ, 0, out, stream);
}
int mf_sde_matern_transitions_grad_packed_f64(int64_t B, int64_t n, int ncomp, const int* orders, const double* lam,
                                              const double* var, int per_series, const double* dt, double jitter,
                                              const double* g_packed, double* out, void* stream) {
    return run_grad<double>(B, n, ncomp, orders, lam, var, per_series, dt, jitter, g_packed, nullptr, 1, out, stream);
}
int mf_sde_matern_transitions_grad_packed_f32(int64_t B, int64_t n, int ncomp, const int* orders, const float* lam,
                                              const float* var, int per_series, const float* dt, float jitter,
                                              const float* g_packed, float* out, void* stream) {
    return run_grad<float>(B, n, ncomp, orders, lam, var, per_series, dt, jitter, g_packed, nullptr, 1, out, stream);
}
int mf_sde_matern_prior_chol_grad_f64(int64_t B, int ncomp, const int* orders, const double* lam, const double* var, int per_series,
                                      double jitter, const double* g_cholP0, double* out, void* stream) {
    return run_prior_grad<double>(B, ncomp, orders, lam, var, per_series, jitter, g_cholP0, out, stream);
}
int mf_sde_matern_prior_chol_grad_f32(int64_t B, int ncomp, const int* orders, const float* lam, const float* var, int per_series,
                                      float jitter, const float* g_cholP0, float* out, void* stream) {
    return run_prior_grad<float>(B, ncomp, orders, lam, var, per_series, jitter, g_cholP0, out, stream);
}

}  // extern "C"
