// Row forms (mf_row.hpp: one 16-lane DPP row per sub-problem) of the forward covariance / mean scans in time behind
// StateSpaceModel.marginals (state_space_model.py:232-275) and of the local form of kl_divergence (state_space_model.py:528-593),
// i.e. of par_tak_*<SRC = 1> (mf_btd_par.hpp) and ssm_kl_local_kernel (mf_kl_grad.hpp): same levels, same workspace tensors and
// results.  At d = 9 those are the kernels a sparse-variational model's ELBO spends its time in (BASELINE config 4: the KL of
// the posterior chain against the prior was 1.72 ms at 512 series x 1000 steps, 1.4 ms of it in these four kernels, each of
// them a ~5 k-instruction step on one lane with the composed map in LDS).
//
// The recursion  Sigma_p = A Sigma_{p-1} A^T + C C^T,  mu_p = A mu_{p-1} + b  in row layout (lane i holds row i):
//     A S      = sum_k own A_k * bcast_k(S_j)           (P Q form)
//     (AS) A^T = sum_k own (AS)_k * bcast_j(A_k)        (P Q^T form)
//     C C^T    = sum_{k <= j} own C_k * bcast_j(C_k)
// and a vector is one more column handled by the same instructions (distributed over the lanes: lane i holds element i).
#pragma once
#include "mf_kl_grad.hpp"
#include "mf_row_par.hpp"

namespace mf {
namespace row {

// out[j] (+)= sum_k own a[k] * bcast_k(b[j])   for j < NJ:   rows of A B from rows of A and of B
template <typename T, int D, int NJ> MF_DEV void row_mul(const T (&a)[D], const T (&b)[NJ], T (&out)[NJ]) {
    using P = Dpp<T>;
    sfor<D>([&](auto k) {
        constexpr int kk = decltype(k)::value;
        sfor<NJ>([&](auto j) { P::template fmac<kk>(out[decltype(j)::value], b[decltype(j)::value], a[kk]); });
    });
}
// out[j] += sum_k own a[k] * bcast_j(b[k])     rows of A B^T from rows of A and of B
template <typename T, int D> MF_DEV void row_mul_t(const T (&a)[D], const T (&b)[D], T (&out)[D]) {
    using P = Dpp<T>;
    sfor<D>([&](auto k) {
        constexpr int kk = decltype(k)::value;
        sfor<D>([&](auto j) { P::template fmac<decltype(j)::value>(out[decltype(j)::value], b[kk], a[kk]); });
    });
}
// out[j] = sum_{k <= j} own c[k] * bcast_j(c[k]):  rows of C C^T for lower-triangular C (c: own row, zero above the diagonal)
template <typename T, int D> MF_DEV void row_cct(const T (&c)[D], T (&out)[D]) {
    using P = Dpp<T>;
    sfor<D>([&](auto j) { out[decltype(j)::value] = T(0); });
    sfor<D>([&](auto k) {
        constexpr int kk = decltype(k)::value;
        sfor2<kk, D>([&](auto j) { P::template fmac<decltype(j)::value>(out[decltype(j)::value], c[kk], c[kk]); });
    });
}
template <typename T, int D> MF_DEV void load_row_lower(const T* __restrict__ blk, int rc, T (&v)[D]) {
    sfor<D>([&](auto j) { v[decltype(j)::value] = decltype(j)::value <= rc ? blk[rc * D + decltype(j)::value] : T(0); });
}

// From the rows of a lower-triangular C (zero above the diagonal) and the own 1 / C[r][r]:
//   CiT: lane l holds column l of C^-1 (CiT[j] = Ci[j][l]),  Qi: lane i holds row i of Q^-1 = C^-T C^-1.
template <typename T, int D> MF_DEV void row_qinv(T (&Crow)[D], T dinv, int r, T (&CiT)[D], T (&Qi)[D]) {
    using P = Dpp<T>;
    T acc[D];
    sfor<D>([&](auto i) { acc[decltype(i)::value] = r == decltype(i)::value ? T(1) : T(0); });
    fence(Crow);
    fence1(dinv);
    sfor<D>([&](auto kq) {
        constexpr int kk = decltype(kq)::value;
        CiT[kk] = acc[kk] * P::template bcast<kk>(dinv);
        sfor2<kk + 1, D>([&](auto i) { P::template fnmac<decltype(i)::value>(acc[decltype(i)::value], Crow[kk], CiT[kk]); });
    });
    fence(CiT);
    sfor<D>([&](auto j) { Qi[decltype(j)::value] = T(0); });
    sfor<D>([&](auto l) {                                   // Ci[l][j] = 0 for l < j: exact zeros, skipped
        constexpr int ll = decltype(l)::value;
        sfor<ll + 1>([&](auto j) { P::template fmac<decltype(j)::value>(Qi[decltype(j)::value], CiT[ll], CiT[ll]); });
    });
}
// one position of the level-0 kernels:  Sigma(p) = Mp Sigma(p-1) Mp^T + N_p  (mu(p) = Mp mu(p-1) + o_p)
//   SRC 0 (block Takahashi on a Cholesky factor, block_tri_diag.py:318-337, backward): position p = block k = n-1-p;
//          Sigma_k = L_k^-T L_k^-1 + G_k^T Sigma_{k+1} G_k with G_k = W_k L_k^-1: Mp = G_k^T = L_k^-T W_k^T, N = L_k^-T L_k^-1;
//          src.a = ldiag, src.b = lsub.
//   SRC 1 (marginal covariances / means, forward): block p; Mp = A_{p-1}, N = C C^T with C = cholQ_{p-1} (p = 0: cholP0, Mp = 0),
//          o = b_{p-1} (mu0);  src.a = cholQ, src.b = A, src.c0 = cholP0.
//   SRC 2 (their adjoint, M_k = N_k + A_k^T M_{k+1} A_k, backward): position p = block n-1-p; Mp = A_k^T (rows = columns of
//          A_k; position 0: none), N_k read from a buffer of symmetric blocks;  src.a = N [B,n,D,D], src.b = A.
template <typename T, int D> struct RowCovStep {
    T Arow[D], Nn[D], o;         // own row of Mp (zero for position 0), own row of N_p, own offset element
};
template <typename T, int D, int SRC, bool MEAN>
MF_DEV void load_cov_step(const TakSrc<T>& src, const T* mu0, const T* b, long s, long n, long p, int rc, RowCovStep<T, D>& d) {
    const T keep = p > 0 ? T(1) : T(0);
    if constexpr (SRC == 0) {
        const long k = n - 1 - p;
        const T* lblk = src.a + (s * n + k) * D * D;
        T Crow[D], CiT[D];
        load_row_lower<T, D>(lblk, rc, Crow);
        row_qinv<T, D>(Crow, t_rcp<T>(lblk[rc * (D + 1)]), rc, CiT, d.Nn);          // N = L^-T L^-1; CiT: own row of L^-T
        sfor<D>([&](auto j) { d.Arow[decltype(j)::value] = T(0); });
        if (n > 1) {
            T Wt[D];
            load_col<T, D>(src.b + (s * (n - 1) + (k < n - 1 ? k : n - 2)) * D * D, rc, Wt);   // row rc of W^T
            fence(Wt);
            row_mul<T, D, D>(CiT, Wt, d.Arow);                                      // rows of L^-T W^T = G^T
        }
        d.o = T(0);
    } else if constexpr (SRC == 1) {
        const long kt = p > 0 ? p - 1 : 0;
        const T* cblk = p > 0 ? src.a + (s * (n - 1) + kt) * D * D : src.c0 + s * D * D;
        T Crow[D];
        load_row_lower<T, D>(cblk, rc, Crow);
        if (n > 1) load_row<T, D>(src.b + (s * (n - 1) + kt) * D * D, rc, d.Arow);
        else sfor<D>([&](auto j) { d.Arow[decltype(j)::value] = T(0); });
        if constexpr (MEAN) d.o = p > 0 ? b[(s * (n - 1) + kt) * D + rc] : mu0[s * D + rc];
        else d.o = T(0);
        fence(Crow);
        row_cct<T, D>(Crow, d.Nn);
    } else if constexpr (SRC == 3) {       // forward, explicit terms: Mp = M_{p-1} (its own row), N_p from a buffer of symmetric blocks
        load_row<T, D>(src.a + (s * n + p) * D * D, rc, d.Nn);
        if (n > 1) load_row<T, D>(src.b + (s * (n - 1) + (p > 0 ? p - 1 : 0)) * D * D, rc, d.Arow);
        else sfor<D>([&](auto j) { d.Arow[decltype(j)::value] = T(0); });
        d.o = T(0);
    } else {
        const long k = n - 1 - p;
        load_row<T, D>(src.a + (s * n + k) * D * D, rc, d.Nn);
        if (n > 1) load_col<T, D>(src.b + (s * (n - 1) + (k < n - 1 ? k : n - 2)) * D * D, rc, d.Arow);
        else sfor<D>([&](auto j) { d.Arow[decltype(j)::value] = T(0); });
        d.o = T(0);
    }
    sfor<D>([&](auto j) { d.Arow[decltype(j)::value] *= keep; });
}

// ---- level 0 up-sweep (par_tak_up0_kernel<SRC = 1>): the chunk's map  Sigma -> Mc Sigma Mc^T + Nc,  mu -> Mc mu + q ----
template <typename T, int D, int SRC, bool MEAN>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, false)) row_cov_up0_kernel(long B, long n, long len, long P, TakSrc<T> src,
                                                                                           T* __restrict__ oG, T* __restrict__ oN,
                                                                                           TakMeanUp<T> mup) {
    const RowChunkId q = row_chunk_id<D>(B, P);
    const long p0 = q.c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T Mr[D + 1], Nr[D];          // rows of Mc with the offset q as column D; rows of Nc
    for (long p = p0; p < p1; ++p) {
        asm volatile("s_nop 4");
        RowCovStep<T, D> d;
        load_cov_step<T, D, SRC, MEAN>(src, mup.mu0, mup.b, q.s, n, p, q.rc, d);
        T (&Nn)[D] = d.Nn;
        if (p == p0) {
            sfor<D>([&](auto j) { Mr[decltype(j)::value] = d.Arow[decltype(j)::value]; Nr[decltype(j)::value] = Nn[decltype(j)::value]; });
            Mr[D] = d.o;
        } else {
            T T1[D + 1], T2[D];
            sfor<D + 1>([&](auto j) { T1[decltype(j)::value] = T(0); });
            sfor<D>([&](auto j) { T2[decltype(j)::value] = T(0); });
            fence(Mr);
            fence(Nr);
            row_mul<T, D, D + 1>(d.Arow, Mr, T1);                     // A [Mc | q]
            row_mul<T, D, D>(d.Arow, Nr, T2);                         // A Nc
            fence(d.Arow);
            row_mul_t<T, D>(T2, d.Arow, Nn);                          // + (A Nc) A^T
            sfor<D>([&](auto j) { Mr[decltype(j)::value] = T1[decltype(j)::value]; Nr[decltype(j)::value] = Nn[decltype(j)::value]; });
            Mr[D] = T1[D] + d.o;
        }
    }
    if (q.valid && q.r < D) {
        sfor<D>([&](auto j) {
            constexpr int jj = decltype(j)::value;
            oG[q.id * D * D + jj * D + q.r] = Mr[jj];                 // G = Mc^T
            oN[q.id * D * D + q.r * D + jj] = Nr[jj];
            if constexpr (MEAN) mup.oM[q.id * D * D + q.r * D + jj] = Mr[jj];
        });
        if constexpr (MEAN) mup.oc[q.id * D + q.r] = Mr[D];
    }
}

// ---- reduced levels, up (par_tak_up_kernel): compose the maps of a run of chunks ----
template <typename T, int D>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, false)) row_cov_up_kernel(long B, long n, long len, long P,
                                                                                          const T* __restrict__ Gs, const T* __restrict__ Ns,
                                                                                          T* __restrict__ oG, T* __restrict__ oN) {
    const RowChunkId q = row_chunk_id<D>(B, P);
    const long p0 = q.c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T Mr[D], Nr[D];
    load_col<T, D>(Gs + (q.s * n + p0) * D * D, q.rc, Mr);            // rows of M = columns of G
    load_row<T, D>(Ns + (q.s * n + p0) * D * D, q.rc, Nr);
    for (long p = p0 + 1; p < p1; ++p) {
        asm volatile("s_nop 4");
        T Mp[D], Nn[D], T1[D], T2[D];
        load_col<T, D>(Gs + (q.s * n + p) * D * D, q.rc, Mp);
        load_row<T, D>(Ns + (q.s * n + p) * D * D, q.rc, Nn);
        sfor<D>([&](auto j) { T1[decltype(j)::value] = T(0); T2[decltype(j)::value] = T(0); });
        fence(Mr);
        fence(Nr);
        row_mul<T, D, D>(Mp, Mr, T1);
        row_mul<T, D, D>(Mp, Nr, T2);
        fence(Mp);
        row_mul_t<T, D>(T2, Mp, Nn);
        sfor<D>([&](auto j) { Mr[decltype(j)::value] = T1[decltype(j)::value]; Nr[decltype(j)::value] = Nn[decltype(j)::value]; });
    }
    if (q.valid && q.r < D) {
        sfor<D>([&](auto j) {
            constexpr int jj = decltype(j)::value;
            oG[q.id * D * D + jj * D + q.r] = Mr[jj];
            oN[q.id * D * D + q.r * D + jj] = Nr[jj];
        });
    }
}

// ---- reduced levels, down (par_tak_down_kernel): Sigma at every position of the level ----
template <typename T, int D>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, false)) row_cov_down_kernel(long B, long n, long len, long P,
                                                                                            const T* __restrict__ Gs, const T* __restrict__ Ns,
                                                                                            const T* __restrict__ up, T* __restrict__ Z) {
    const RowChunkId q = row_chunk_id<D>(B, P);
    const long p0 = q.c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T Sr[D];
    sfor<D>([&](auto j) { Sr[decltype(j)::value] = T(0); });
    if (q.c > 0) load_row<T, D>(up + (q.s * P + q.c - 1) * D * D, q.rc, Sr);
    for (long p = p0; p < p1; ++p) {
        asm volatile("s_nop 4");
        T Nn[D];
        load_row<T, D>(Ns + (q.s * n + p) * D * D, q.rc, Nn);
        if (p > 0) {
            T Mp[D], T2[D];
            load_col<T, D>(Gs + (q.s * n + p) * D * D, q.rc, Mp);
            sfor<D>([&](auto j) { T2[decltype(j)::value] = T(0); });
            fence(Sr);
            row_mul<T, D, D>(Mp, Sr, T2);
            fence(Mp);
            row_mul_t<T, D>(T2, Mp, Nn);
        }
        sfor<D>([&](auto j) { Sr[decltype(j)::value] = Nn[decltype(j)::value]; });
        if (q.valid && q.r < D) sfor<D>([&](auto j) { Z[(q.s * n + p) * D * D + q.r * D + decltype(j)::value] = Sr[decltype(j)::value]; });
    }
}

// ---- level 0 emit (par_tak_emit_kernel<SRC = 1>): every chunk restarts from its boundary values and writes the marginal
// covariances, Cov(x_p, x_{p-1}) = A Sigma_{p-1} (osub, optional) and - MEAN - the marginal means ----
template <typename T, int D, int SRC, bool MEAN>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, false)) row_cov_emit_kernel(long B, long n, long len, long P, TakSrc<T> src,
                                                                                            const T* __restrict__ up, T* __restrict__ odiag,
                                                                                            T* __restrict__ osub, TakMean<T> mean) {
    using Pp = Dpp<T>;
    const RowChunkId q = row_chunk_id<D>(B, P);
    const long p0 = q.c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T Sr[D], mu = T(0);
    sfor<D>([&](auto j) { Sr[decltype(j)::value] = T(0); });
    if (q.c > 0) {
        load_row<T, D>(up + (q.s * P + q.c - 1) * D * D, q.rc, Sr);
        if constexpr (MEAN) mu = mean.up[(q.s * P + q.c - 1) * D + q.rc];
    }
    const bool st = q.valid && q.r < D;
    for (long p = p0; p < p1; ++p) {
        asm volatile("s_nop 4");
        RowCovStep<T, D> d;
        load_cov_step<T, D, SRC, MEAN>(src, mean.mu0, mean.b, q.s, n, p, q.rc, d);
        T (&Nn)[D] = d.Nn;
        const long k = (SRC == 1 || SRC == 3) ? p : n - 1 - p;        // block this position writes
        if (p > 0) {
            T T2[D];
            sfor<D>([&](auto j) { T2[decltype(j)::value] = T(0); });
            fence(Sr);
            row_mul<T, D, D>(d.Arow, Sr, T2);                          // A Sigma_{p-1}
            if constexpr (SRC == 1) {
                if (osub && st) sfor<D>([&](auto j) { osub[(q.s * (n - 1) + p - 1) * D * D + q.r * D + decltype(j)::value] = T2[decltype(j)::value]; });
            } else if constexpr (SRC == 0 || SRC == 2) {       // sub-diagonal block of the inverse: -Sigma_{k+1} G_k = -(G^T Sigma)^T
                if (osub && st) sfor<D>([&](auto j) { osub[(q.s * (n - 1) + k) * D * D + decltype(j)::value * D + q.r] = -T2[decltype(j)::value]; });
            }
            fence(d.Arow);
            row_mul_t<T, D>(T2, d.Arow, Nn);
            if constexpr (MEAN) {
                T acc = d.o;
                fence1(mu);
                sfor<D>([&](auto k) { Pp::template fmac<decltype(k)::value>(acc, mu, d.Arow[decltype(k)::value]); });
                mu = acc;
            }
        } else if constexpr (MEAN) {
            mu = d.o;
        }
        sfor<D>([&](auto j) { Sr[decltype(j)::value] = Nn[decltype(j)::value]; });
        if (st) {
            sfor<D>([&](auto j) { odiag[(q.s * n + k) * D * D + q.r * D + decltype(j)::value] = Sr[decltype(j)::value]; });
            if constexpr (MEAN) mean.out[(q.s * n + k) * D + q.r] = mu;
        }
    }
}

// ---- the affine scan of the means alone (par_means_up0_kernel / par_means_emit_kernel): x_p = Mp x_{p-1} + o_p ----
// REV: the transposed recursion run backwards, lam_k = o_k + A_k^T lam_{k+1} (the adjoint of the means): position p is block
// n-1-p and its matrix is A_{n-1-p}^T.  offs [Br, n, D] may be NULL (zero offsets).  `sign` multiplies the matrix (the posterior
// offsets x_k = eta_k - U_k x_{k+1} are this scan with A = U^T, sign = -1, or with the chain's -U^T and sign = +1).
template <typename T, int D, bool REV>
MF_DEV void load_mean_step(const T* __restrict__ A, const T* __restrict__ offs, long s, long rr, long n, long p, int rc, T sign, T (&Mrow)[D], T& o) {
    const T keep = p > 0 ? sign : T(0);
    if (n > 1) {
        if (!REV) load_row<T, D>(A + (s * (n - 1) + (p > 0 ? p - 1 : 0)) * D * D, rc, Mrow);
        else load_col<T, D>(A + (s * (n - 1) + (p > 0 ? n - 1 - p : n - 2)) * D * D, rc, Mrow);
    } else {
        sfor<D>([&](auto j) { Mrow[decltype(j)::value] = T(0); });
    }
    sfor<D>([&](auto j) { Mrow[decltype(j)::value] *= keep; });
    o = offs ? offs[(rr * n + (REV ? n - 1 - p : p)) * D + rc] : T(0);
}
template <typename T, int D, bool REV>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, false)) row_means_up0_kernel(long Bl, long Br, long n, long len, long P,
                                                                                             const T* __restrict__ A, const T* __restrict__ offs,
                                                                                             T* __restrict__ oM, T* __restrict__ oc, T sign) {
    const RowChunkId q = row_chunk_id<D>(Br, P);
    const long rr = q.s, s = rr % Bl;
    const long p0 = q.c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T Mr[D + 1];                                                      // rows of the composed map with its offset as column D
    for (long p = p0; p < p1; ++p) {
        asm volatile("s_nop 4");
        T Mrow[D], o;
        load_mean_step<T, D, REV>(A, offs, s, rr, n, p, q.rc, sign, Mrow, o);
        if (p == p0) {
            sfor<D>([&](auto j) { Mr[decltype(j)::value] = Mrow[decltype(j)::value]; });
            Mr[D] = o;
        } else {
            T T1[D + 1];
            sfor<D + 1>([&](auto j) { T1[decltype(j)::value] = T(0); });
            fence(Mr);
            row_mul<T, D, D + 1>(Mrow, Mr, T1);
            sfor<D>([&](auto j) { Mr[decltype(j)::value] = T1[decltype(j)::value]; });
            Mr[D] = T1[D] + o;
        }
    }
    if (q.valid && q.r < D) {
        sfor<D>([&](auto j) { oM[q.id * D * D + q.r * D + decltype(j)::value] = Mr[decltype(j)::value]; });
        oc[q.id * D + q.r] = Mr[D];
    }
}
template <typename T, int D, bool REV>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, false)) row_means_emit_kernel(long Bl, long Br, long n, long len, long P,
                                                                                              const T* __restrict__ A, const T* __restrict__ offs,
                                                                                              const T* __restrict__ up, T* __restrict__ out, T sign) {
    using Pp = Dpp<T>;
    const RowChunkId q = row_chunk_id<D>(Br, P);
    const long rr = q.s, s = rr % Bl;
    const long p0 = q.c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    T x = T(0);
    if (q.c > 0) x = up[(rr * P + q.c - 1) * D + q.rc];
    for (long p = p0; p < p1; ++p) {
        asm volatile("s_nop 4");
        T Mrow[D], o;
        load_mean_step<T, D, REV>(A, offs, s, rr, n, p, q.rc, sign, Mrow, o);
        T acc = o;
        fence1(x);
        sfor<D>([&](auto l) { Pp::template fmac<decltype(l)::value>(acc, x, Mrow[decltype(l)::value]); });
        x = acc;
        if (q.valid && q.r < D) out[(rr * n + (REV ? n - 1 - p : p)) * D + q.r] = x;
    }
}

// the inputs of the `marginals` adjoint in the workspace layout of the scans (ssm_adjoint_sym_inputs_kernel): N = gS + gS^T, n = gm;
// a row per block - lane r reads row r and column r of gS
template <typename T, int D>
__global__ void __launch_bounds__(64) row_adjoint_sym_inputs_kernel(long blocks, const T* __restrict__ gm, const T* __restrict__ gS,
                                                                    T* __restrict__ oN, T* __restrict__ on) {
    const RowChunkId q = row_chunk_id<D>(blocks, 1);
    if (!(q.valid && q.r < D)) return;
    sfor<D>([&](auto j) {
        constexpr int jj = decltype(j)::value;
        oN[q.id * D * D + q.r * D + jj] = gS ? gS[q.id * D * D + q.r * D + jj] + gS[q.id * D * D + jj * D + q.r] : T(0);
    });
    on[q.id * D + q.r] = gm ? gm[q.id * D + q.r] : T(0);
}

// ---- kl_divergence, local form (ssm_kl_local_kernel): a row per (series, step) ----
// term(C2, C1, X): 1/2 [ |C2^-1 C1|_F^2 + |C2^-1 x|^2 (+ tr(W S W^T), W = C2^-1 X) ] + log|C2| - log|C1| with the D columns of X in
// the lanes < D and the vector x in lane D: both go through ONE substitution with C2's rows broadcast.
template <typename T, int D> struct RowKlTerm {
    using P = Dpp<T>;
    // in-lane forward substitution: col <- C2^-1 col   (C2r: own row of C2, dinv: own 1 / diagonal element)
    static MF_DEV void solve(T (&C2r)[D], T dinv, T (&col)[D]) {
        sfor<D>([&](auto kq) {
            constexpr int kk = decltype(kq)::value;
            col[kk] *= P::template bcast<kk>(dinv);
            sfor2<kk + 1, D>([&](auto i) { P::template fnmac<decltype(i)::value>(col[decltype(i)::value], C2r[kk], col[kk]); });
        });
    }
};

template <typename T, int D>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, false)) row_kl_local_kernel(
    long B, long Tn, const T* __restrict__ mu0_1, const T* __restrict__ C0_1, const T* __restrict__ A_1, const T* __restrict__ b_1,
    const T* __restrict__ C_1, const T* __restrict__ mu0_2, const T* __restrict__ C0_2, const T* __restrict__ A_2,
    const T* __restrict__ b_2, const T* __restrict__ C_2, const T* __restrict__ pm, const T* __restrict__ pS, T* __restrict__ part,
    T* __restrict__ oN, T* __restrict__ on, int* info) {
    using P = Dpp<T>;
    using K = RowKlTerm<T, D>;
    // one row per (series, step): id = s Tn + k.  The row does little more than two substitutions and one product, so its index
    // arithmetic counts: 32-bit when the problem allows (a 64-bit division is ~100 instructions on this machine)
    RowChunkId q;
    {
        const int lane = threadIdx.x;
        q.r = lane & 15;
        q.rc = q.r < D ? q.r : D - 1;
        const long total = B * Tn;
        const long id_raw = (long)blockIdx.x * 4 + (lane >> 4);
        q.valid = id_raw < total;
        q.id = q.valid ? id_raw : total - 1;
    }
    const long id = q.id;
    long s, k;
    if (B * Tn < (1L << 31)) { const unsigned su = (unsigned)id / (unsigned)Tn; s = su; k = (long)((unsigned)id - su * (unsigned)Tn); }
    else { s = id / Tn; k = id % Tn; }
    const int r = q.r, rc = q.rc;
    const T in_mat = r < D ? T(1) : T(0), in_vec = r == D ? T(1) : T(0);
    bool bad = false;
    T val = T(0);                                                     // this lane's share of the row's value
    T mdist = pm[id * D + rc];
    // ---- the transition k -> k + 1 (absent at the last step of a series: flagged, loads clamped) ----
    const bool has_t = k + 1 < Tn;
    const long tid = s * (Tn - 1) + (has_t ? k : (Tn > 1 ? Tn - 2 : 0));
    if (Tn > 1) {
        const T tf = has_t ? T(1) : T(0);
        T C2r[D], Xc[D], Yc[D], dAr[D];
        load_row_lower<T, D>(C_2 + tid * D * D, rc, C2r);
        const T c2d = C_2[tid * D * D + rc * (D + 1)], c1d = C_1[tid * D * D + rc * (D + 1)];
        bad |= has_t && r < D && (!(c2d != T(0)) || !(c1d != T(0)));
        T dinv = t_rcp<T>(c2d);
        {
            T a1[D], a2[D];
            load_col<T, D>(A_1 + tid * D * D, rc, a1);
            load_col<T, D>(A_2 + tid * D * D, rc, a2);
            sfor<D>([&](auto i) { Xc[decltype(i)::value] = (a1[decltype(i)::value] - a2[decltype(i)::value]) * in_mat; });
            load_row<T, D>(A_1 + tid * D * D, rc, a1);
            load_row<T, D>(A_2 + tid * D * D, rc, a2);
            sfor<D>([&](auto i) { dAr[decltype(i)::value] = a1[decltype(i)::value] - a2[decltype(i)::value]; });
        }
        sfor<D>([&](auto i) { Yc[decltype(i)::value] = decltype(i)::value >= rc ? C_1[tid * D * D + decltype(i)::value * D + rc] * in_mat : T(0); });
        // eps = db + dA m, distributed, then into lane D's column
        T eps = b_1[tid * D + rc] - b_2[tid * D + rc];
        fence1(mdist);
        sfor<D>([&](auto j) { P::template fmac<decltype(j)::value>(eps, mdist, dAr[decltype(j)::value]); });
        fence1(eps);
        sfor<D>([&](auto i) { P::template fmac<decltype(i)::value>(Xc[decltype(i)::value], eps, in_vec); });
        fence(C2r);
        fence1(dinv);
        K::solve(C2r, dinv, Xc);                                       // lanes < D: columns of W = C2^-1 dA; lane D: u = C2^-1 eps
        K::solve(C2r, dinv, Yc);                                       // lanes < D: columns of C2^-1 C1
        T acc = T(0);
        sfor<D>([&](auto i) { acc = __builtin_fma(Yc[decltype(i)::value], Yc[decltype(i)::value], acc); });
        // lanes < D: (W S W^T)_jj from own row of S; lane D: |u|^2 - the same multiply-add with its own column as "W S"
        T Srow[D], WS[D];
        load_row<T, D>(pS + id * D * D, rc, Srow);
        sfor<D>([&](auto i) { WS[decltype(i)::value] = Xc[decltype(i)::value] * in_vec; });
        fence(Xc);
        sfor<D>([&](auto l) {
            constexpr int ll = decltype(l)::value;
            const T sl = Srow[ll] * in_mat;
            sfor<D>([&](auto i) { P::template fmac<ll>(WS[decltype(i)::value], Xc[decltype(i)::value], sl); });
        });
        sfor<D>([&](auto i) { acc = __builtin_fma(WS[decltype(i)::value], Xc[decltype(i)::value], acc); });
        const T ratio = c2d * t_rcp<T>(c1d);                           // log|c2| - log|c1| in one logarithm
        const T lg = log(ratio < T(0) ? -ratio : ratio);
        val = tf * (T(0.5) * acc + in_mat * lg);
        if (oN) {
            // adjoint inputs of the backward: N = W^T W, n = W^T u (zero at the last step)
            T Nrow[D], nv = T(0);
            sfor<D>([&](auto j) { Nrow[decltype(j)::value] = T(0); });
            sfor<D>([&](auto i) {
                constexpr int ii = decltype(i)::value;
                sfor<D>([&](auto j) { P::template fmac<decltype(j)::value>(Nrow[decltype(j)::value], Xc[ii], Xc[ii]); });
                P::template fmac<D>(nv, Xc[ii], Xc[ii]);
            });
            if (q.valid && r < D) {
                sfor<D>([&](auto j) { oN[id * D * D + r * D + decltype(j)::value] = tf * Nrow[decltype(j)::value]; });
                on[id * D + r] = tf * nv;
            }
        }
    } else if (oN && q.valid && r < D) {
        sfor<D>([&](auto j) { oN[id * D * D + r * D + decltype(j)::value] = T(0); });
        on[id * D + r] = T(0);
    }
    // ---- the initial state (k = 0 rows only; whole rows take or skip the branch) ----
    if (k == 0) {
        asm volatile("s_nop 4");
        T C2r[D], Yc[D];
        load_row_lower<T, D>(C0_2 + s * D * D, rc, C2r);
        const T c2d = C0_2[s * D * D + rc * (D + 1)], c1d = C0_1[s * D * D + rc * (D + 1)];
        bad |= r < D && (!(c2d != T(0)) || !(c1d != T(0)));
        T dinv = t_rcp<T>(c2d);
        sfor<D>([&](auto i) { Yc[decltype(i)::value] = decltype(i)::value >= rc ? C0_1[s * D * D + decltype(i)::value * D + rc] * in_mat : T(0); });
        T d0 = mdist - mu0_2[s * D + rc];
        fence1(d0);
        sfor<D>([&](auto i) { P::template fmac<decltype(i)::value>(Yc[decltype(i)::value], d0, in_vec); });    // lane D: m_0 - mu0_2
        fence(C2r);
        fence1(dinv);
        K::solve(C2r, dinv, Yc);
        T acc = T(0);
        sfor<D>([&](auto i) { acc = __builtin_fma(Yc[decltype(i)::value], Yc[decltype(i)::value], acc); });
        const T ratio = c2d * t_rcp<T>(c1d);
        const T lg = log(ratio < T(0) ? -ratio : ratio);
        val += (in_mat + in_vec) * T(0.5) * acc + in_mat * lg;
    }
    // ---- sum over the lanes 0..D of the row ----
    asm volatile("s_nop 4");
    fence1(val);
    T tot = T(0);
    sfor<D + 1>([&](auto i) { tot += P::template bcast<decltype(i)::value>(val); });
    if (q.valid && r == 0) part[id] = tot - T(0.5) * T(D);
    if (q.valid && bad && info) raise_info(info);
}

// ---- kl_divergence with few series, fused: the level-0 emit of q1's covariance / mean scan evaluates the divergence's local
// terms on the way (row_cov_emit_kernel + row_kl_local_kernel in one kernel).  A chunk restarts (m, S) from its boundary values,
// and at position p it has exactly what the term of transition p-1 -> p needs, (m_{p-1}, S_{p-1}), in registers: the moments are
// neither written nor read back unless the caller wants them (the backward does: omean / ocov / ocross / oN / on).  One partial
// value per chunk; a wave per series adds them in a fixed order. ----
template <typename T, int D> struct RowKlChains {
    const T *mu0_1, *C0_1, *A_1, *b_1, *C_1, *mu0_2, *C0_2, *A_2, *b_2, *C_2;
};
template <typename T, int D>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, true)) row_kl_emit_kernel(      // (two waves per SIMD at d >= 8 fp64: no spills)
    long B, long n, long len, long Pn, RowKlChains<T, D> ch, const T* __restrict__ up_cov, const T* __restrict__ up_mean,
    T* __restrict__ omean, T* __restrict__ ocov, T* __restrict__ ocross, T* __restrict__ oN, T* __restrict__ on,
    T* __restrict__ part, int* info) {
    using P = Dpp<T>;
    using K = RowKlTerm<T, D>;
    const RowChunkId q = row_chunk_id<D>(B, Pn);
    const long s = q.s;
    const int r = q.r, rc = q.rc;
    const long p0 = q.c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    const bool st = q.valid && r < D;
    const T in_mat = r < D ? T(1) : T(0), in_vec = r == D ? T(1) : T(0);
    const TakSrc<T> src{ch.C_1, ch.A_1, ch.C0_1};
    bool bad = false;
    T val = T(0);
    T Sr[D], mu = T(0);
    sfor<D>([&](auto j) { Sr[decltype(j)::value] = T(0); });
    if (q.c > 0) {
        load_row<T, D>(up_cov + (s * Pn + q.c - 1) * D * D, rc, Sr);
        mu = up_mean[(s * Pn + q.c - 1) * D + rc];
    }
    // 1/2 [ |C2^-1 C1|_F^2 + |C2^-1 x|^2 ] + log|C2| - log|C1| from own rows / columns; X: columns of a matrix in lanes < D and the
    // vector x in lane D (both solved in place), returns this lane's share
    auto chol_term = [&](const T* c2blk, const T* c1blk, T (&Xc)[D]) {
        T C2r[D], Yc[D];
        load_row_lower<T, D>(c2blk, rc, C2r);
        const T c2d = c2blk[rc * (D + 1)], c1d = c1blk[rc * (D + 1)];
        bad |= r < D && (!(c2d != T(0)) || !(c1d != T(0)));
        T dinv = t_rcp<T>(c2d);
        sfor<D>([&](auto i) { Yc[decltype(i)::value] = decltype(i)::value >= rc ? c1blk[decltype(i)::value * D + rc] * in_mat : T(0); });
        fence(C2r);
        fence1(dinv);
        K::solve(C2r, dinv, Xc);
        K::solve(C2r, dinv, Yc);
        T acc = T(0);
        sfor<D>([&](auto i) { acc = __builtin_fma(Yc[decltype(i)::value], Yc[decltype(i)::value], acc); });
        const T ratio = c2d * t_rcp<T>(c1d);
        return T(0.5) * acc + in_mat * log(ratio < T(0) ? -ratio : ratio);
    };
    for (long p = p0; p < p1; ++p) {
        asm volatile("s_nop 4");
        RowCovStep<T, D> d;
        load_cov_step<T, D, 1, true>(src, ch.mu0_1, ch.b_1, s, n, p, rc, d);
        if (p == 0) {
            // ---- the initial state: m_0 = mu0_1 ----
            T Xc[D];
            T d0 = d.o - ch.mu0_2[s * D + rc];
            sfor<D>([&](auto i) { Xc[decltype(i)::value] = T(0); });
            fence1(d0);
            sfor<D>([&](auto i) { P::template fmac<decltype(i)::value>(Xc[decltype(i)::value], d0, in_vec); });
            T v = chol_term(ch.C0_2 + s * D * D, ch.C0_1 + s * D * D, Xc);
            T acc = T(0);
            sfor<D>([&](auto i) { acc = __builtin_fma(Xc[decltype(i)::value], Xc[decltype(i)::value], acc); });
            val += v + in_vec * T(0.5) * acc;
        } else {
            // ---- the term of transition p-1 -> p, from (m_{p-1}, S_{p-1}) = (mu, Sr) ----
            const long tid = s * (n - 1) + p - 1;
            T Xc[D], dAr[D];
            {
                T a1[D], a2[D];
                load_col<T, D>(ch.A_1 + tid * D * D, rc, a1);
                load_col<T, D>(ch.A_2 + tid * D * D, rc, a2);
                sfor<D>([&](auto i) { Xc[decltype(i)::value] = (a1[decltype(i)::value] - a2[decltype(i)::value]) * in_mat; });
                load_row<T, D>(ch.A_2 + tid * D * D, rc, a2);
                sfor<D>([&](auto i) { dAr[decltype(i)::value] = d.Arow[decltype(i)::value] - a2[decltype(i)::value]; });
            }
            T eps = d.o - ch.b_2[tid * D + rc];
            fence1(mu);
            sfor<D>([&](auto j) { P::template fmac<decltype(j)::value>(eps, mu, dAr[decltype(j)::value]); });      // db + dA m
            fence1(eps);
            sfor<D>([&](auto i) { P::template fmac<decltype(i)::value>(Xc[decltype(i)::value], eps, in_vec); });
            T v = chol_term(ch.C_2 + tid * D * D, ch.C_1 + tid * D * D, Xc);        // Xc: columns of W = C2^-1 dA; lane D: u = C2^-1 eps
            T WS[D], acc = T(0);
            sfor<D>([&](auto i) { WS[decltype(i)::value] = Xc[decltype(i)::value] * in_vec; });
            fence(Xc);
            sfor<D>([&](auto l) {
                constexpr int ll = decltype(l)::value;
                const T sl = Sr[ll] * in_mat;
                sfor<D>([&](auto i) { P::template fmac<ll>(WS[decltype(i)::value], Xc[decltype(i)::value], sl); });
            });
            sfor<D>([&](auto i) { acc = __builtin_fma(WS[decltype(i)::value], Xc[decltype(i)::value], acc); });
            val += v + T(0.5) * acc;
            if (oN) {                                                 // adjoint inputs of the backward at step p-1: N = W^T W, n = W^T u
                T Nrow[D], nv = T(0);
                sfor<D>([&](auto j) { Nrow[decltype(j)::value] = T(0); });
                sfor<D>([&](auto i) {
                    constexpr int ii = decltype(i)::value;
                    sfor<D>([&](auto j) { P::template fmac<decltype(j)::value>(Nrow[decltype(j)::value], Xc[ii], Xc[ii]); });
                    P::template fmac<D>(nv, Xc[ii], Xc[ii]);
                });
                if (st) {
                    sfor<D>([&](auto j) { oN[(s * n + p - 1) * D * D + r * D + decltype(j)::value] = Nrow[decltype(j)::value]; });
                    on[(s * n + p - 1) * D + r] = nv;
                }
            }
        }
        // ---- (m_p, S_p) ----
        if (p > 0) {
            T T2[D];
            sfor<D>([&](auto j) { T2[decltype(j)::value] = T(0); });
            fence(Sr);
            row_mul<T, D, D>(d.Arow, Sr, T2);
            if (ocross && st) sfor<D>([&](auto j) { ocross[(s * (n - 1) + p - 1) * D * D + r * D + decltype(j)::value] = T2[decltype(j)::value]; });
            fence(d.Arow);
            row_mul_t<T, D>(T2, d.Arow, d.Nn);
            T acc = d.o;
            fence1(mu);
            sfor<D>([&](auto l) { P::template fmac<decltype(l)::value>(acc, mu, d.Arow[decltype(l)::value]); });
            mu = acc;
        } else {
            mu = d.o;
        }
        sfor<D>([&](auto j) { Sr[decltype(j)::value] = d.Nn[decltype(j)::value]; });
        if (st) {
            if (ocov) sfor<D>([&](auto j) { ocov[(s * n + p) * D * D + r * D + decltype(j)::value] = Sr[decltype(j)::value]; });
            if (omean) omean[(s * n + p) * D + r] = mu;
            if (oN && p + 1 == n) {                                   // no transition leaves the last step
                sfor<D>([&](auto j) { oN[(s * n + p) * D * D + r * D + decltype(j)::value] = T(0); });
                on[(s * n + p) * D + r] = T(0);
            }
        }
    }
    asm volatile("s_nop 4");
    fence1(val);
    T tot = T(0);
    sfor<D + 1>([&](auto i) { tot += P::template bcast<decltype(i)::value>(val); });
    if (q.valid && r == 0) part[q.id] = tot - T(0.5) * T(D) * T(p1 - p0);
    if (q.valid && bad && info) raise_info(info);
}

}   // namespace row
}   // namespace mf
