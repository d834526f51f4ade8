// K0 and its reduction level for state dimensions whose elimination state does not fit a lane's 512 registers
// (d >= 7 in fp64, d = 9 in fp32): the SPIKE part of the state - X (d x d, coupling to the chunk's left separator) and
// GU (the separator's accumulated pivot contribution) - lives in LDS instead of registers.
//
// Same math, same chunk convention and same outputs as kf_chunk_kernel / red_chunk_kernel (mf_kernels.hpp).  What changes
// is where things live and the order of a step:
//   * X and GU are touched in three streaming passes per step (solve the spike columns, fold them into GU, propagate
//     them through W), each moving one column / row through registers: ~3 d^2 LDS reads and 2 d^2 writes per step;
//   * the next block's own pivot D_n = Q^-1 + H^T R^-1 H is formed directly in Phi once the factor it held is dead, and
//     H^T R^-1 H is accumulated pair by pair from global memory - no d x d temporaries besides ONE (A -> B -> Y -> W).
// Measured before (B=512, T=1000, d=9, m=3, fp64): kf_chunk_kernel 3.7 KB of scratch per lane, 230 us per block step.
// LDS layout: element e of lane l at word e * 64 + l (consecutive lanes = consecutive words: conflict-free).
#pragma once
#include "mf_kernels.hpp"

namespace mf {

template <typename T, int D> struct LdsSpike {
    static constexpr int NX = D * D, NG = D * (D + 1) / 2;
    static constexpr int BYTES = (NX + NG) * 64 * (int)sizeof(T);
    T* x;
    T* g;
    MF_DEV void bind(T* smem, int lane) { x = smem + lane; g = smem + NX * 64 + lane; }
    MF_DEV T X(int i, int j) const { return x[(i * D + j) * 64]; }
    MF_DEV void setX(int i, int j, T v) const { x[(i * D + j) * 64] = v; }
    MF_DEV T G(int i, int j) const { return g[(i * (i + 1) / 2 + j) * 64]; }
    MF_DEV void setG(int i, int j, T v) const { g[(i * (i + 1) / 2 + j) * 64] = v; }
    MF_DEV void zero() const {
        MF_UNROLL for (int e = 0; e < NX; ++e) x[e * 64] = T(0);
        MF_UNROLL for (int e = 0; e < NG; ++e) g[e * 64] = T(0);
    }
};

template <typename T, int D> struct ElimX {
    T Phi[D][D];   // lower: partial pivot of the current block, then its Cholesky factor, then the next block's pivot
    T Li[D];
    T t[D];
    T gU[D];
    T quad;
    LogAcc<T> laL;
    bool bad;
    LdsSpike<T, D> sp;

    MF_DEV void init(T* smem, int lane) {
        MF_UNROLL for (int i = 0; i < D; ++i) {
            t[i] = T(0); gU[i] = T(0); Li[i] = T(0);
            MF_UNROLL for (int j = 0; j < D; ++j) Phi[i][j] = T(0);
        }
        quad = T(0);
        laL.init();
        bad = false;
        sp.bind(smem, lane);
        sp.zero();
    }
    MF_DEV void eliminate_main() {
        chol_lower<T, D>(Phi, Li, laL, bad);
        laL.renorm();
        trsv_lower<T, D>(Phi, Li, t);
        quad += dot_self<T, D>(t);
    }
    // V = L^-1 X in place, gU -= V^T z, GU -= V^T V.  The column / row loops are deliberately NOT unrolled: unrolled, the
    // scheduler hoists the LDS reads of several columns at once and the live set is back over the register file.
    MF_DEV void eliminate_spike() {
#pragma unroll 1
        for (int c = 0; c < D; ++c) {
            T* xc = sp.x + c * 64;
            T v[D];
            MF_UNROLL for (int i = 0; i < D; ++i) v[i] = xc[i * D * 64];
            trsv_lower<T, D>(Phi, Li, v);
            T s = T(0);
            MF_UNROLL for (int i = 0; i < D; ++i) s += v[i] * t[i];
            MF_UNROLL for (int q = 0; q < D; ++q) gU[q] -= (q == c) ? s : T(0);
            MF_UNROLL for (int i = 0; i < D; ++i) xc[i * D * 64] = v[i];
        }
        // GU rows in two groups so that at most ~D^2/4 accumulators are live; group 0 only needs the first R0 columns of V
        constexpr int R0 = (2 * D + 2) / 3;
        fold_rows<0, R0>();
        fold_rows<R0, D>();
    }
    template <int A0, int A1> MF_DEV void fold_rows() {
        if constexpr (A0 < A1) {
            T acc[A1 - A0][A1];
            MF_UNROLL for (int a = A0; a < A1; ++a) MF_UNROLL for (int b = 0; b <= a; ++b) acc[a - A0][b] = sp.G(a, b);
#pragma unroll 1
            for (int k = 0; k < D; ++k) {
                const T* xr = sp.x + k * D * 64;
                T v[A1];
                MF_UNROLL for (int b = 0; b < A1; ++b) v[b] = xr[b * 64];
                MF_UNROLL for (int a = A0; a < A1; ++a) MF_UNROLL for (int b = 0; b <= a; ++b) acc[a - A0][b] -= v[a] * v[b];
            }
            MF_UNROLL for (int a = A0; a < A1; ++a) MF_UNROLL for (int b = 0; b <= a; ++b) sp.setG(a, b, acc[a - A0][b]);
        }
    }
    // X <- -(W V)
    MF_DEV void propagate_spike(const T (&W)[D][D]) {
#pragma unroll 1
        for (int c = 0; c < D; ++c) {
            T* xc = sp.x + c * 64;
            T v[D], col[D];
            MF_UNROLL for (int k = 0; k < D; ++k) v[k] = xc[k * D * 64];
            MF_UNROLL for (int i = 0; i < D; ++i) col[i] = W[i][0] * v[0];
            MF_UNROLL for (int k = 1; k < D; ++k) MF_UNROLL for (int i = 0; i < D; ++i) col[i] += W[i][k] * v[k];
            MF_UNROLL for (int i = 0; i < D; ++i) xc[i * D * 64] = -col[i];
        }
    }
    MF_DEV void store(const RedSys<T>& out, long idx, T scalar) const {
        store_sym<T, D>(out.Dv + idx * D * D, Phi);
        store_vec<T, D>(out.tv + idx * D, t);
        store_vec<T, D>(out.gU + idx * D, gU);
        T* gu = out.GU + idx * D * D;
        T* f = out.F + idx * D * D;
        MF_UNROLL for (int i = 0; i < D; ++i) {
            MF_UNROLL for (int j = 0; j <= i; ++j) { const T v = sp.G(i, j); gu[i * D + j] = v; gu[j * D + i] = v; }
            MF_UNROLL for (int j = 0; j < D; ++j) f[i * D + j] = sp.X(i, j);
        }
        out.sc[idx] = scalar;
    }
};

// Phi(lower) += H^T R^-1 H, t += H^T R^-1 y, returns y^T R^-1 y; one (output, output) pair at a time from global memory
template <typename T, int D>
MF_DEV T obs_apply_pairs(const T* __restrict__ Hk, const T* __restrict__ yk, const T* __restrict__ Ri, int m, T (&Phi)[D][D],
                         T (&t)[D]) {
    T yry = T(0);
    for (int o = 0; o < m; ++o) {
        T ry = T(0);
        for (int p = 0; p < m; ++p) ry += Ri[o * m + p] * yk[p];
        yry += yk[o] * ry;
        T ho[D];
        MF_UNROLL for (int i = 0; i < D; ++i) ho[i] = Hk[o * D + i];
        MF_UNROLL for (int i = 0; i < D; ++i) t[i] += ho[i] * ry;
        for (int p = 0; p < m; ++p) {
            const T r = Ri[o * m + p];
            T hp[D];
            MF_UNROLL for (int j = 0; j < D; ++j) hp[j] = r * Hk[p * D + j];
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Phi[i][j] += ho[i] * hp[j];
        }
    }
    return yry;
}

// The same with H_k (m x D) and y_k staged in LDS by the step's load group (element e of lane l at word e * 64 + l): with
// several outputs the pair-by-pair form above pays a dependent global round trip per (output, output) pair - 12 per step at
// m = 3 with one or two waves per CU to hide them.  OBS_LDS is chosen by the launcher when the image still fits the CU with the
// same number of waves (mf_inst.hip: x_obs_lds).
template <typename T, int D> struct LdsObs {
    static constexpr int bytes(int m) { return (m * D + m) * 64 * (int)sizeof(T); }      // sized for the call's m outputs
    T* h;
    T* y;
    MF_DEV void bind(T* base, int lane, int m) { h = base + lane; y = base + m * D * 64 + lane; }
    // the step's observation rows: global -> registers -> LDS, issued with the step's other loads
    MF_DEV void stage(const T* __restrict__ Hk, const T* __restrict__ yk, int m) const {
        T hv[MF_MAXM][D], yv[MF_MAXM];
        MF_UNROLL for (int o = 0; o < MF_MAXM; ++o) {
            const int oc = o < m ? o : m - 1;                       // clamped: always a valid address, unused rows ignored
            yv[o] = yk[oc];
            MF_UNROLL for (int i = 0; i < D; ++i) hv[o][i] = Hk[oc * D + i];
        }
        MF_UNROLL for (int o = 0; o < MF_MAXM; ++o) {
            if (o < m) {
                y[o * 64] = yv[o];
                MF_UNROLL for (int i = 0; i < D; ++i) h[(o * D + i) * 64] = hv[o][i];
            }
        }
    }
    MF_DEV T apply(const T* __restrict__ Ri, int m, T (&Phi)[D][D], T (&t)[D]) const {
        T yry = T(0);
        for (int o = 0; o < m; ++o) {
            T ry = T(0);
            for (int p = 0; p < m; ++p) ry += Ri[o * m + p] * y[p * 64];
            yry += y[o * 64] * ry;
            T ho[D];
            MF_UNROLL for (int i = 0; i < D; ++i) ho[i] = h[(o * D + i) * 64];
            MF_UNROLL for (int i = 0; i < D; ++i) t[i] += ho[i] * ry;
            for (int p = 0; p < m; ++p) {
                const T r = Ri[o * m + p];
                T hp[D];
                MF_UNROLL for (int j = 0; j < D; ++j) hp[j] = r * h[(p * D + j) * 64];
                MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) Phi[i][j] += ho[i] * hp[j];
            }
        }
        return yry;
    }
};

// Level 0, chunk convention of kf_chunk_kernel: chunk c owns blocks [c T / P, (c+1) T / P).
template <typename T, int D, bool OBS_LDS = false>
__global__ void __launch_bounds__(64) kf_chunk_x_kernel(KfArgs<T> a, RedSys<T> out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x;
    const long total = a.B * a.P;
    const long id_raw = (long)blockIdx.x * 64 + lane;
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;
    const long s = id / a.P, c = id % a.P;
    const long k0 = (c * a.Tn) / a.P, k1 = valid ? ((c + 1) * a.Tn) / a.P : k0;
    const int m = a.m;
    const T* As = a.A + s * (a.Tn - 1) * D * D;
    const T* Qs = a.cholQ + s * (a.Tn - 1) * D * D;
    const T* bs = a.b + s * (a.Tn - 1) * D;
    const T* Hs = a.H + s * a.Tn * m * D;
    const T* ys = a.y + s * a.Tn * m;

    ElimX<T, D> E;
    E.init(reinterpret_cast<T*>(smem_raw), lane);
    LdsObs<T, D> ob;
    ob.bind(reinterpret_cast<T*>(smem_raw + LdsSpike<T, D>::BYTES), lane, a.m);
    LogAcc<T> laC;
    laC.init();
    T acc_yry = T(0), acc_ww = T(0);
    auto observe = [&](long k, const T* Ri) {
        if constexpr (OBS_LDS) return ob.apply(Ri, m, E.Phi, E.t);
        else return obs_apply_pairs<T, D>(Hs + k * m * D, ys + k * m, Ri, m, E.Phi, E.t);
    };

    for (long k = k0; k < k1; ++k) {
        T Ci[D][D], w[D], Bm[D][D];
        {
            // this step's loads as one group (pointer selects instead of branches; block 0 has no transition: clamped, unused)
            T C[D][D], mvec[D];
            const long kt = k > 0 ? k - 1 : 0;
            load_lower<T, D>(k == 0 ? a.cholP0 + s * D * D : Qs + kt * D * D, C);
            load_vec<T, D>(k == 0 ? a.mu0 + s * D : bs + kt * D, mvec);
            load_mat<T, D, D>(As + kt * D * D, Bm);
            if constexpr (OBS_LDS) ob.stage(Hs + k * m * D, ys + k * m, m);
            __builtin_amdgcn_sched_barrier(0);
            tri_inv_lower<T, D>(C, Ci, laC, E.bad);
            laC.renorm();
            trimul_lower_vec<T, D>(Ci, mvec, w);
            acc_ww += dot_self<T, D>(w);
        }
        const T* Ri = a.rinv_per_step ? a.Rinv + (s * a.Tn + k) * m * m : a.Rinv;
        if (k == 0) {
            trimulT_self_lower<T, D>(Ci, E.Phi);
            trimulT_lower_vec<T, D>(Ci, w, E.t);
            acc_yry += observe(k, Ri);
            continue;
        }
        T btw[D];
        trimul_lower_inplace<T, D, D>(Ci, Bm);                 // B = C^-1 A
        gemv_t<T, D, D>(Bm, w, btw);                           // A^T Q^-1 m
        if (k == k0) {
            // block k-1 is the separator on the left: GU = B^T B, gU = -B^T w, X = -C^-T B
            MF_UNROLL for (int i = 0; i < D; ++i) {
                T row[D];
                MF_UNROLL for (int j = 0; j <= i; ++j) row[j] = Bm[0][i] * Bm[0][j];
                MF_UNROLL for (int q = 1; q < D; ++q) MF_UNROLL for (int j = 0; j <= i; ++j) row[j] += Bm[q][i] * Bm[q][j];
                MF_UNROLL for (int j = 0; j <= i; ++j) E.sp.setG(i, j, row[j]);
                E.gU[i] = -btw[i];
            }
            neg_trimulT_lower_inplace<T, D, D>(Ci, Bm);
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) E.sp.setX(i, j, Bm[i][j]);
            trimulT_self_lower<T, D>(Ci, E.Phi);
            trimulT_lower_vec<T, D>(Ci, w, E.t);
            acc_yry += observe(k, Ri);
            continue;
        }
        syrk_tn_lower<T, D, D>(Bm, E.Phi, T(1));               // D_{k-1} complete
        MF_UNROLL for (int i = 0; i < D; ++i) E.t[i] -= btw[i];
        E.eliminate_main();
        E.eliminate_spike();
        trsm_right_lower_t<T, D, D>(E.Phi, E.Li, Bm);          // Y = B L^-T
        neg_trimulT_lower_inplace<T, D, D>(Ci, Bm);            // W = -C^-T Y      (the factor in Phi is dead from here on)
        E.propagate_spike(Bm);
        T wz[D];
        gemv_n<T, D, D>(Bm, E.t, wz);
        trimulT_self_lower<T, D>(Ci, E.Phi);                   // Q_k^-1 straight into Phi
        {
            T rn[D];
            trimulT_lower_vec<T, D>(Ci, w, rn);
            MF_UNROLL for (int i = 0; i < D; ++i) E.t[i] = rn[i] - wz[i];
        }
        acc_yry += observe(k, Ri);
        syrk_nt_lower<T, D, D>(Bm, E.Phi, T(-1));
    }
    if (valid) {
        const T scalar = T(-0.5) * (acc_yry + acc_ww) + T(0.5) * E.quad - laC.value() - E.laL.value();
        E.store(out, id, scalar);
        if (E.bad && a.info) raise_info(a.info);
    }
}

// Reduction level with the spike in LDS: RedSys(n) -> RedSys(P).
template <typename T, int D>
__global__ void __launch_bounds__(64) red_chunk_x_kernel(RedSys<T> in, RedSys<T> out, long B, long P, int* info) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x;
    const long total = B * P;
    const long id_raw = (long)blockIdx.x * 64 + lane;
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;
    const long s = id / P, c = id % P;
    const long k0 = (c * in.n) / P, k1 = valid ? ((c + 1) * in.n) / P : k0;
    ElimX<T, D> E;
    E.init(reinterpret_cast<T*>(smem_raw), lane);
    T acc_sc = T(0);
    for (long k = k0; k < k1; ++k) {
        T Dn[D][D], rn[D], sc;
        load_red_block<T, D>(in, s, k, Dn, rn, sc);
        acc_sc += sc;
        if (k == 0) {
            MF_UNROLL for (int i = 0; i < D; ++i) {
                E.t[i] = rn[i];
                MF_UNROLL for (int j = 0; j <= i; ++j) E.Phi[i][j] = Dn[i][j];
            }
            continue;
        }
        T W[D][D];
        load_mat<T, D, D>(in.F + (s * in.f_stride + k + in.f_off) * D * D, W);
        if (k == k0) {
            MF_UNROLL for (int i = 0; i < D; ++i) {
                E.t[i] = rn[i];
                MF_UNROLL for (int j = 0; j < D; ++j) E.sp.setX(i, j, W[i][j]);
                MF_UNROLL for (int j = 0; j <= i; ++j) E.Phi[i][j] = Dn[i][j];
            }
            continue;
        }
        E.eliminate_main();
        E.eliminate_spike();
        trsm_right_lower_t<T, D, D>(E.Phi, E.Li, W);           // W = S L^-T
        E.propagate_spike(W);
        T wz[D];
        gemv_n<T, D, D>(W, E.t, wz);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            E.t[i] = rn[i] - wz[i];
            MF_UNROLL for (int j = 0; j <= i; ++j) E.Phi[i][j] = Dn[i][j];
        }
        syrk_nt_lower<T, D, D>(W, E.Phi, T(-1));
    }
    if (valid) {
        const T scalar = acc_sc + T(0.5) * E.quad - E.laL.value();
        E.store(out, id, scalar);
        if (E.bad && info) raise_info(info);
    }
}

}  // namespace mf
