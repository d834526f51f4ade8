// GaussianProcessRegression.log_likelihood (models/gaussian_process_regression.py:150-160 of the reference) with the kernel ->
// state-space-model step fused into the ROW log-likelihood kernel (mf_row.hpp): 7 <= d <= 15, any concatenation of Matern-1/2,
// -3/2, -5/2 components - a Sum kernel (one output, H = [1 0 0 | 1 0 0 | ...], kernels/sde_kernel.py:660-690) or
// IndependentMultiOutput (one output per component, H[o] = e_{first state of component o}, sde_kernel.py:826-880; up to four
// outputs).  BASELINE config 4 is the second form: three Matern-5/2 components, three outputs, d = 9.
//
// kf_row_kernel reads a transition's (A_k, chol Q_k, b_k, H_k) - 1.6 kB per step at d = 9; here every lane generates its own row
// of chol Q_k and its own column of A_k from dt_k and its component's two hyper-parameters (closed forms of kernels/matern.py:
// 66-86, :299-356, :434-501; Q = P_inf - A P_inf A^T + jitter, sde_kernel.py:421-446), so a step reads 8 (1 + m) bytes.  The lanes
// of one component compute the same 3 x 3 block (~200 instructions per step, one exp and three rsqrt among them, against the
// ~1 000 of the elimination).
#pragma once
#include "mf_row.hpp"

namespace mf {
namespace row {

constexpr int GPR_MAX_COMP = 15;
template <typename T> struct GprRowArgs {
    long B, Tn;
    int ncomp, multi;                 // multi: one output per component (IndependentMultiOutput), else a Sum kernel (m = 1)
    int order[GPR_MAX_COMP];          // 1, 3, 5 per component
    const T* lam; const T* var; long hstride;      // [ncomp] (hstride 0) or [B, ncomp]
    const T* t; const T* y;           // [B, T], [B, T, M]
    const T* rinv;                    // [M, M]
    T jitter;
    long P;
    int* info;
};

// one component's transition over dt: own row i of chol Q (crow, lower) and own column i of A (acol); prior: chol(P_inf + jitter).
// A = exp(-l dt) (I + N dt + N^2 dt^2 / 2) with N = F + l I nilpotent; for Matern-5/2, N = [[l,1,0],[0,l,1],[-l^3,-3l^2,-2l]] and
// N^2 = [[l^2,2l,1],[-l^3,-2l^2,-l],[l^4,2l^3,l^2]] written out, and Q = P - (A P) A^T with the five non-zeros of P_inf
// (v, v l^2 / 3 twice with a minus sign in the corners, v l^4) - ~60 multiply-adds instead of the generic triple loops' ~250.
template <typename T> MF_DEV void gpr_component(int order, T l, T v, T dt, T jitter, bool prior, int i, T (&crow)[3], T (&acol)[3]) {
    T A[3][3], Q[3][3];                         // Q: lower triangle; rows / columns beyond the component's size: identity
    MF_UNROLL for (int a = 0; a < 3; ++a) MF_UNROLL for (int b = 0; b < 3; ++b) { A[a][b] = T(0); Q[a][b] = a == b ? T(1) : T(0); }
    const T e = prior ? T(0) : exp(-l * dt);
    if (order == 1) {
        A[0][0] = e;
        Q[0][0] = v - e * e * v + jitter;
    } else if (order == 3) {
        const T l2 = l * l, p1 = v * l2;
        A[0][0] = e * (T(1) + l * dt);
        A[0][1] = e * dt;
        A[1][0] = -e * l2 * dt;
        A[1][1] = e * (T(1) - l * dt);
        Q[0][0] = v - (A[0][0] * A[0][0] * v + A[0][1] * A[0][1] * p1) + jitter;
        Q[1][0] = -(A[1][0] * A[0][0] * v + A[1][1] * A[0][1] * p1);
        Q[1][1] = p1 - (A[1][0] * A[1][0] * v + A[1][1] * A[1][1] * p1) + jitter;
    } else {
        const T l2 = l * l, l3 = l2 * l, l4 = l2 * l2, h = T(0.5) * dt * dt;
        A[0][0] = e * (T(1) + l * dt + l2 * h);
        A[0][1] = e * (dt + T(2) * l * h);
        A[0][2] = e * h;
        A[1][0] = -e * l3 * h;
        A[1][1] = e * (T(1) + l * dt - T(2) * l2 * h);
        A[1][2] = e * (dt - l * h);
        A[2][0] = e * (l4 * h - l3 * dt);
        A[2][1] = e * (T(2) * l3 * h - T(3) * l2 * dt);
        A[2][2] = e * (T(1) - T(2) * l * dt + l2 * h);
        const T kap = v * l2 / T(3), p22 = v * l4;
        T Mx[3][3];                              // A P
        MF_UNROLL for (int a = 0; a < 3; ++a) {
            Mx[a][0] = A[a][0] * v - A[a][2] * kap;
            Mx[a][1] = A[a][1] * kap;
            Mx[a][2] = A[a][2] * p22 - A[a][0] * kap;
        }
        const T P[3][3] = {{v, T(0), -kap}, {T(0), kap, T(0)}, {-kap, T(0), p22}};
        MF_UNROLL for (int a = 0; a < 3; ++a)
            MF_UNROLL for (int b = 0; b <= a; ++b)
                Q[a][b] = P[a][b] - (Mx[a][0] * A[b][0] + Mx[a][1] * A[b][1] + Mx[a][2] * A[b][2]) + (a == b ? jitter : T(0));
    }
    T C[3][3];
    MF_UNROLL for (int a = 0; a < 3; ++a) MF_UNROLL for (int b = 0; b < 3; ++b) C[a][b] = T(0);
    MF_UNROLL for (int b = 0; b < 3; ++b) {
        T s = Q[b][b];
        MF_UNROLL for (int p = 0; p < b; ++p) s -= C[b][p] * C[b][p];
        const T inv = t_rsqrt<T>(s);
        C[b][b] = s * inv;
        MF_UNROLL for (int a = b + 1; a < 3; ++a) {
            T w = Q[a][b];
            MF_UNROLL for (int p = 0; p < b; ++p) w -= C[a][p] * C[b][p];
            C[a][b] = w * inv;
        }
    }
    MF_UNROLL for (int b = 0; b < 3; ++b) {
        crow[b] = i == 0 ? C[0][b] : (i == 1 ? C[1][b] : C[2][b]);
        acol[b] = i == 0 ? A[b][0] : (i == 1 ? A[b][1] : A[b][2]);
    }
}

// Level 0 of the log-likelihood, conventions of kf_row_kernel (chunk c owns blocks [c T / P, (c+1) T / P); 64 threads = 4 chunks).
// One wavefront per SIMD fewer than kf_row_kernel: the generator's 3 x 3 temporaries sit on top of the elimination state (at three
// waves / 168 registers the d = 9 kernel spilt 230 B per lane inside the loop), and with no loads to hide the third wave buys little:
// level-0 kernel at config 4's size 0.455 -> 0.377 ms together with the written-out closed forms of gpr_component.
constexpr int gpr_row_waves(int d) { return d <= 9 ? 2 : row_waves_per_simd(d); }        // (d >= 10: already two / one)
template <typename T, int D, int M>
__global__ void __launch_bounds__(64, gpr_row_waves(D)) gpr_row_kernel(GprRowArgs<T> a, RedSys<T> out) {
    static_assert(D + 1 <= 16 && M <= D, "one row of 16 lanes per chunk");
    const int lane = threadIdx.x;
    const int r = lane & 15;
    const int rcl = r < D ? r : D - 1;
    const long total = a.B * a.P;
    const long id_raw = (long)blockIdx.x * 4 + (lane >> 4);
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;
    const long s = id / a.P, c = id % a.P;
    const long k0 = (c * a.Tn) / a.P, k1 = ((c + 1) * a.Tn) / a.P;

    // this lane's component: offset of its block, own index inside it, hyper-parameters
    int comp = 0, off = 0, ord = a.order[0];
    {
        int o = 0;
        for (int q = 0; q < a.ncomp; ++q) {
            const int kq = (a.order[q] + 1) / 2;
            if (rcl >= o) { comp = q; off = o; ord = a.order[q]; }
            o += kq;
        }
    }
    const int ii = rcl - off;
    const T lam = a.lam[s * a.hstride + comp], var = a.var[s * a.hstride + comp];
    const bool live = r < D;
    const T* ts = a.t + s * a.Tn;
    const T* ys = a.y + s * a.Tn * M;

    RowChunk<T, D, M> E;
    E.init(r);
    RiRegs<T, M> rshared;
    sfor<M * M>([&](auto e) { rshared.v[decltype(e)::value] = to_uniform(a.rinv[decltype(e)::value]); });

    // emission row of output o in this lane's column: the first state of a component observes (all of them for a Sum kernel,
    // component o alone for independent outputs)
    T hsel[M];
    sfor<M>([&](auto o) {
        constexpr int oo = decltype(o)::value;
        hsel[oo] = (live && ii == 0 && (!a.multi || comp == oo)) ? T(1) : T(0);
    });

    auto gen = [&](T dt, bool prior, T (&Crow)[D], T& cdiag, T (&Aa)[D], T (&Ha)[M]) {
        T crow[3], acol[3];
        gpr_component<T>(ord, lam, var, dt, a.jitter, prior, ii, crow, acol);
        sfor<D>([&](auto k) {
            constexpr int kk = decltype(k)::value;
            const int j = kk - off;
            Crow[kk] = j == 0 ? crow[0] : (j == 1 ? crow[1] : (j == 2 ? crow[2] : T(0)));
            const T av = j == 0 ? acol[0] : (j == 1 ? acol[1] : (j == 2 ? acol[2] : T(0)));
            Aa[kk] = live ? av : T(0);
        });
        // (entries of the block's unused rows / columns are exact zeros: A and chol Q are zero there, the padding rows of the
        // 3 x 3 working block are never selected because ii < K)
        cdiag = ii == 0 ? crow[0] : (ii == 1 ? crow[1] : crow[2]);
        sfor<M>([&](auto o) { Ha[decltype(o)::value] = hsel[decltype(o)::value]; });
    };
    auto load_y = [&](long k) { return r < M ? ys[k * M + r] : T(0); };

    // ---- first block of the chunk ----
    {
        T Crow[D], Aa[D], Ha[M], cdiag;
        const bool first = k0 == 0;
        const T dt = first ? T(0) : ts[k0] - ts[k0 - 1];
        gen(dt, first, Crow, cdiag, Aa, Ha);
        E.start(Crow, cdiag, Aa, T(0), Ha, load_y(k0), rshared);
    }
    // ---- interior blocks ----
    T t_prev = ts[k0];
    for (long k = k0 + 1; k < k1; ++k) {
        asm volatile("s_nop 4");
        T Crow[D], Aa[D], Ha[M], cdiag;
        const T t_cur = ts[k];
        gen(t_cur - t_prev, false, Crow, cdiag, Aa, Ha);
        t_prev = t_cur;
        E.step(Crow, cdiag, Aa, T(0), Ha, load_y(k), rshared);
    }
    // ---- the chunk's reduced block (as kf_row_kernel) ----
    using P = Dpp<T>;
    T logc = E.laC.value(), qd = E.quad, w2 = E.ww, yr = E.yry;
    fence1(logc); fence1(qd); fence1(w2); fence1(yr);
    T lc = T(0);
    sfor<D>([&](auto i) { lc += P::template bcast<decltype(i)::value>(logc); });
    const T scalar = T(-0.5) * (P::template bcast<D>(yr) + P::template bcast<D>(w2)) + T(0.5) * P::template bcast<D>(qd) - lc -
                     T(0.5) * E.laL.value();
    if (valid) {
        if (r < D) {
            T* dv = out.Dv + id * D * D + r * D;
            T* gu = out.GU + id * D * D + r * D;
            T* f = out.F + id * D * D + r;
            sfor<D>([&](auto j) {
                constexpr int jj = decltype(j)::value;
                dv[jj] = E.Phi[jj];
                gu[jj] = E.GU[jj];
                f[jj * D] = E.Xa[jj];
            });
            out.gU[id * D + r] = E.gU;
            if (r == 0) out.sc[id] = scalar;
        } else if (r == D) {
            T* tv = out.tv + id * D;
            sfor<D>([&](auto j) { tv[decltype(j)::value] = E.Phi[decltype(j)::value]; });
        }
        if (E.bad && a.info) raise_info(a.info);
    }
}

}  // namespace row
}  // namespace mf
