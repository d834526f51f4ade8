// GPR log-likelihood with the kernel -> state-space-model step FUSED into the Kalman sweep:
// the transitions A_k = exp(F dt_k) and chol(Pinf - A_k Pinf A_k^T) of a Sum of Matern components are generated in
// registers from (dt_k, hyper-parameters) inside the level-0 kernel, so a step reads 16 bytes (dt_k, y_k) instead of the
// (2 d^2 + 3 d + 1) s bytes of materialised tensors - the path is then bound by arithmetic, not by HBM.
// Same partitioned elimination, same step (kf_lds_step) and same reduction levels as mf_kf_loglik; one output (m = 1),
// zero state offsets, emission H = [1 0 0 | 1 0 0 ...] (Sum.generate_emission_model, kernels/sde_kernel.py:670-688).
// Closed forms: kernels/matern.py:66-86 (order 1), :299-356 (order 3), :434-501 (order 5); Q: sde_kernel.py:421-446.
#pragma once
#include "mf_kf_lds.hpp"

#include <type_traits>

namespace mf {

constexpr int gpr_size(int order) { return (order + 1) / 2; }

// A (dense D x D, block diagonal) and the lower Cholesky factor of Q (or of Pinf + jitter for the first block) of a
// concatenation of up to two Matern components of compile-time orders O0, O1 (O1 = 0: one component).
template <typename T, int O0, int O1> struct GprGen {
    static constexpr int K0 = gpr_size(O0), K1 = O1 ? gpr_size(O1) : 0, D = K0 + K1;
    T lam[2], var[2], jitter;

    template <int O, int K> MF_DEV void comp(T l, T v, T dt, T (&A)[K][K], T (&P)[K][K]) const {
        const T e = exp(-l * dt);
        MF_UNROLL for (int i = 0; i < K; ++i) MF_UNROLL for (int j = 0; j < K; ++j) P[i][j] = T(0);
        if constexpr (O == 1) {
            A[0][0] = e;
            P[0][0] = v;
        } else if constexpr (O == 3) {
            A[0][0] = e * (T(1) + l * dt);
            A[0][1] = e * dt;
            A[1][0] = -e * l * l * dt;
            A[1][1] = e * (T(1) - l * dt);
            P[0][0] = v;
            P[1][1] = v * l * l;
        } else {
            const T l2 = l * l, l3 = l2 * l, h = T(0.5) * dt * dt;
            const T N[3][3] = {{l, T(1), T(0)}, {T(0), l, T(1)}, {-l3, -T(3) * l2, -T(2) * l}};
            MF_UNROLL for (int i = 0; i < 3; ++i)
                MF_UNROLL for (int j = 0; j < 3; ++j) {
                    T n2 = T(0);
                    MF_UNROLL for (int q = 0; q < 3; ++q) n2 += N[i][q] * N[q][j];
                    A[i][j] = e * ((i == j ? T(1) : T(0)) + N[i][j] * dt + n2 * h);
                }
            const T l23 = l2 / T(3);
            P[0][0] = v;
            P[0][2] = -v * l23;
            P[2][0] = -v * l23;
            P[1][1] = v * l23;
            P[2][2] = v * l2 * l2;
        }
    }
    // chol(Q) of one component into the (off, off) block of C; Q = P - A P A^T + jitter (prior: Q = P + jitter)
    template <int K> MF_DEV void chol_block(const T (&A)[K][K], const T (&P)[K][K], bool prior, int off, T (&C)[D][D]) const {
        T Q[K][K];
        MF_UNROLL for (int i = 0; i < K; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                T a = T(0);
                if (!prior) {
                    MF_UNROLL for (int p = 0; p < K; ++p)
                        MF_UNROLL for (int q = 0; q < K; ++q) a += A[i][p] * P[p][q] * A[j][q];
                }
                Q[i][j] = P[i][j] - a + (i == j ? jitter : T(0));
            }
        MF_UNROLL for (int j = 0; j < K; ++j) {
            T s = Q[j][j];
            MF_UNROLL for (int p = 0; p < j; ++p) s -= C[off + j][off + p] * C[off + j][off + p];
            const T inv = t_rsqrt<T>(s);
            C[off + j][off + j] = s * inv;
            MF_UNROLL for (int i = j + 1; i < K; ++i) {
                T v = Q[i][j];
                MF_UNROLL for (int p = 0; p < j; ++p) v -= C[off + i][off + p] * C[off + j][off + p];
                C[off + i][off + j] = v * inv;
            }
        }
    }
    // transition over dt: dense A (as Bm) and C = chol Q;  prior: C = chol(Pinf + jitter), A untouched
    MF_DEV void make(T dt, bool prior, T (&Am)[D][D], T (&C)[D][D]) const {
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) { C[i][j] = T(0); if (!prior) Am[i][j] = T(0); }
        {
            T A[K0][K0], P[K0][K0];
            comp<O0, K0>(lam[0], var[0], dt, A, P);
            chol_block<K0>(A, P, prior, 0, C);
            if (!prior) { MF_UNROLL for (int i = 0; i < K0; ++i) MF_UNROLL for (int j = 0; j < K0; ++j) Am[i][j] = A[i][j]; }
        }
        if constexpr (O1 != 0) {
            T A[K1][K1], P[K1][K1];
            comp<O1, K1>(lam[1], var[1], dt, A, P);
            chol_block<K1>(A, P, prior, K0, C);
            if (!prior) { MF_UNROLL for (int i = 0; i < K1; ++i) MF_UNROLL for (int j = 0; j < K1; ++j) Am[K0 + i][K0 + j] = A[i][j]; }
        }
    }
};

#ifndef MF_NOPUMP_DEFINED
#define MF_NOPUMP_DEFINED
struct NoPump {        // (mf_post_math.hpp has the same stand-in for the host simulation)
    template <int K> MF_HD void small() const {}
    template <int K> MF_HD void big() const {}
};
#endif

template <typename T> struct GprArgs {
    long B, Tn;
    const T* lam; const T* var; long hstride;      // [ncomp] (hstride 0) or [B, ncomp]
    const T* t; const T* y;                          // [B, T], [B, T]
    const T* rinv;                                   // [1]: observation precision
    T jitter;
    long P, L;                                       // chunks per series, transitions per chunk
    int* info;
};

// Level 0: one lane per (series, chunk); chunk c owns transitions [c L, min((c+1) L, T-1)), chunk 0 also block 0.
template <typename T, int O0, int O1, bool SPIKE>
__global__ void __launch_bounds__(64) gpr_chunk_kernel(GprArgs<T> a, RedSys<T> out) {
    using Gen = GprGen<T, O0, O1>;
    constexpr int D = Gen::D;
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= a.B * a.P) return;
    const long s = id / a.P, c = id % a.P;
    const long nt = a.Tn - 1, tau0 = c * a.L;
    long len = nt - tau0;
    if (len > a.L) len = a.L;
    if (len < 0) len = 0;
    // per-lane hyper-parameters live in LDS (5 scalars) and are re-read every step: the sweep itself needs nearly all of
    // the 512 registers of a lane
    __shared__ T hyp[5][64];
    hyp[0][threadIdx.x] = a.lam[s * a.hstride];
    hyp[1][threadIdx.x] = a.var[s * a.hstride];
    hyp[2][threadIdx.x] = O1 ? a.lam[s * a.hstride + 1] : T(0);
    hyp[3][threadIdx.x] = O1 ? a.var[s * a.hstride + 1] : T(0);
    hyp[4][threadIdx.x] = a.rinv[0];
    auto load_gen = [&]() {
        Gen g;
        g.lam[0] = hyp[0][threadIdx.x]; g.var[0] = hyp[1][threadIdx.x];
        g.lam[1] = hyp[2][threadIdx.x]; g.var[1] = hyp[3][threadIdx.x];
        g.jitter = a.jitter;
        return g;
    };
    T hk[D], Rsh[1], zero[D];
    MF_UNROLL for (int i = 0; i < D; ++i) { hk[i] = (i == 0 || (O1 && i == Gen::K0)) ? T(1) : T(0); zero[i] = T(0); }
    Rsh[0] = a.rinv[0];
    (void)Rsh;
    const T* ts = a.t + s * a.Tn;
    const T* ys = a.y + s * a.Tn;

    Elim<T, D, SPIKE> E;
    E.init();
    LogAcc<T> laC;
    laC.init();
    T acc_yry = T(0), acc_ww = T(0);
    if (c == 0) {   // block 0: the stationary prior
        T C[D][D], Ci[D][D], dummy[D][D];
        load_gen().make(T(0), true, dummy, C);
        tri_inv_lower<T, D>(C, Ci, laC, E.bad);
        laC.renorm();
        trimulT_self_lower<T, D>(Ci, E.Phi);
        T y0[1] = {ys[0]};
        acc_yry += Obs<T, D, 1>::apply(hk, y0, Rsh, 1, E.Phi, E.t);
    }
    // time points and observations one step ahead of their use (16 bytes per step: nothing else to stream)
    T t_prev = len > 0 ? ts[tau0] : T(0), t_next = len > 0 ? ts[tau0 + 1] : T(0), y_next = len > 0 ? ys[tau0 + 1] : T(0);
    const NoPump pump;
    auto step = [&](auto first_tag, long j) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const T dt = t_next - t_prev;
        T yk[1] = {y_next};
        t_prev = t_next;
        if (j + 1 < len) { t_next = ts[tau0 + j + 2]; y_next = ys[tau0 + j + 2]; }
        T C[D][D], Bm[D][D], Rk[1];
        load_gen().make(dt, false, Bm, C);
        Rk[0] = hyp[4][threadIdx.x];
        kf_lds_step<T, D, 1, SPIKE, FIRST>(E, laC, acc_yry, acc_ww, C, zero, hk, yk, Rk, Bm, pump, true, c > 0);
    };
    if (len > 0) step(std::integral_constant<bool, true>{}, 0);
    for (long j = 1; j < len; ++j) step(std::integral_constant<bool, false>{}, j);
    const T scalar = T(-0.5) * (acc_yry + acc_ww) + T(0.5) * E.quad - laC.value() - E.laL.value();
    store_chunk<T, D, SPIKE>(out, id, E, scalar);
    if (E.bad && a.info) raise_info(a.info);
}

}  // namespace mf
