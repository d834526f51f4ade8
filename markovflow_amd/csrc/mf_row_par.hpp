// Row forms (mf_row.hpp: one 16-lane DPP row per chunk, matrix rows across the lanes) of the parallel-in-time Cholesky and
// solve of mf_btd_par.hpp - same levels, same workspace tensors, same results; what changes is who works a block step.
// With ONE lane per chunk a block step of the hierarchy is ~1.8 k instructions issued by a single lane at >= 4 cycles each
// (2.3-2.5 us at d = 6 fp32), and with one long chain (BASELINE config 3: B = 1, T = 100000) the ~70
// DEPENDENT steps of the up / down sweeps are the run time.  Spread over the lanes of a row the same step is a few hundred
// instructions: d x d products are d^2 broadcast-FMAs, the Cholesky's trailing update and both substitutions d^2 / 2.
//
// Reference semantics: SymmetricBlockTriDiagonal.cholesky (block_tri_diag.py:423-436, natural-order factor) and
// LowerTriangularBlockTriDiagonal.solve (block_tri_diag.py:339-351).
#pragma once
#include "mf_btd_par.hpp"
#include "mf_row.hpp"

namespace mf {
namespace row {

// wavefronts per SIMD the kernels below are compiled for: 128 registers (4 waves) hold the state up to 7 doubles per row
// (PF: two sets of step data are live, one step of prefetch - chosen by the launcher when the rows are too few to hide a load)
// From d = 10 on (mf_inst.hip compiled with the row kernels only) the unrolled step itself needs more: fp32 three (two with
// prefetch), fp64 two up to d = 12 without prefetch, else one (512 registers with the AGPRs).
constexpr int row_par_waves(int elem, int d, bool pf) {
    if (d >= 10) return elem == 4 ? (pf ? 2 : 3) : (d <= 12 && !pf ? 2 : 1);
    return elem * d > 56 ? (pf ? 2 : 3) : 4;
}

// chunk (series s, chunk c) of the row this lane belongs to; rows past the end repeat the last chunk and store nothing
struct RowChunkId {
    long s, c, id;
    int r, rc;
    bool valid;
};
template <int D> MF_DEV RowChunkId row_chunk_id(long B, long P) {
    RowChunkId q;
    const int lane = threadIdx.x;
    q.r = lane & 15;
    q.rc = q.r < D ? q.r : D - 1;                      // idle lanes shadow lane D - 1 (valid addresses, results unused)
    const long total = B * P;
    const long id_raw = (long)blockIdx.x * 4 + (lane >> 4);
    q.valid = id_raw < total;
    q.id = q.valid ? id_raw : total - 1;
    q.s = q.id / P;
    q.c = q.id % P;
    return q;
}

// Elimination state of the factorisation hierarchy: pivot rows, the coupling to the chunk's left neighbour (columns) and what
// the chunk's interior adds to that neighbour's pivot (rows).  No right-hand side here (lane D idles).
template <typename T, int D> struct RowFact {
    using P = Dpp<T>;
    T Phi[D], Xa[D], GU[D];
    bool bad;
    MF_DEV void init() {
        sfor<D>([&](auto i) { Phi[decltype(i)::value] = T(0); Xa[decltype(i)::value] = T(0); GU[decltype(i)::value] = T(0); });
        bad = false;
    }
    // Factorises the current pivot (rows in Phi) and moves on: W = S L^-T (S: rows of the coupling with the next block),
    // Phi <- Dn - W W^T; with SPIKE also V = L^-1 X, GU -= V^T V, X <- -W V.  EMIT: the factor's rows (explicit zeros above the
    // diagonal) go to lout and W's rows to wout (either may be null).
    template <bool SPIKE, bool EMIT> MF_DEV void advance(T (&S)[D], const T (&Dn)[D], int r, T* lout, T* wout) {
        T W[D];
        sfor<D>([&](auto j) {
            constexpr int jj = decltype(j)::value;
            fence1(Phi[jj]);
            const T s = P::template bcast<jj>(Phi[jj]);
            bad |= !(s > T(0));
            const T inv = row_rsqrt(s);
            Phi[jj] *= inv;
            if constexpr (SPIKE) Xa[jj] *= inv;
            W[jj] = S[jj] * inv;
            fence1(Phi[jj]);
            sfor2<jj + 1, D>([&](auto k) {
                constexpr int kk = decltype(k)::value;
                P::template fnmac<kk>(Phi[kk], Phi[jj], Phi[jj]);
                if constexpr (SPIKE) P::template fnmac<kk>(Xa[kk], Phi[jj], Xa[jj]);
                P::template fnmac<kk>(S[kk], Phi[jj], W[jj]);
            });
        });
        if constexpr (EMIT) {
            if (r < D) {
                if (lout) sfor<D>([&](auto j) { lout[decltype(j)::value] = decltype(j)::value <= r ? Phi[decltype(j)::value] : T(0); });
                if (wout) sfor<D>([&](auto j) { wout[decltype(j)::value] = W[decltype(j)::value]; });
            }
        }
        fence(W);
        if constexpr (SPIKE) {
            fence(Xa);
            T Xn[D];
            sfor<D>([&](auto i) { Xn[decltype(i)::value] = T(0); });
            sfor<D>([&](auto k) {
                constexpr int kk = decltype(k)::value;
                sfor<D>([&](auto j) {
                    constexpr int jj = decltype(j)::value;
                    P::template fnmac<jj>(GU[jj], Xa[kk], Xa[kk]);
                    P::template fnmac<jj>(Xn[jj], W[kk], Xa[kk]);
                });
            });
            sfor<D>([&](auto i) { Xa[decltype(i)::value] = Xn[decltype(i)::value]; });
        }
        sfor<D>([&](auto j) { Phi[decltype(j)::value] = Dn[decltype(j)::value]; });
        sfor<D>([&](auto k) {
            constexpr int kk = decltype(k)::value;
            sfor<D>([&](auto j) { P::template fnmac<decltype(j)::value>(Phi[decltype(j)::value], W[kk], W[kk]); });
        });
    }
    // factor of the current pivot only (the last block of an emitting chunk)
    MF_DEV void factor(int r, T* lout) {
        sfor<D>([&](auto j) {
            constexpr int jj = decltype(j)::value;
            fence1(Phi[jj]);
            const T s = P::template bcast<jj>(Phi[jj]);
            bad |= !(s > T(0));
            Phi[jj] *= row_rsqrt(s);
            fence1(Phi[jj]);
            sfor2<jj + 1, D>([&](auto k) { P::template fnmac<decltype(k)::value>(Phi[decltype(k)::value], Phi[jj], Phi[jj]); });
        });
        if (r < D && lout) sfor<D>([&](auto j) { lout[decltype(j)::value] = decltype(j)::value <= r ? Phi[decltype(j)::value] : T(0); });
    }
};

template <typename T, int D> MF_DEV void load_row(const T* __restrict__ blk, int rc, T (&v)[D]) {
    sfor<D>([&](auto j) { v[decltype(j)::value] = blk[rc * D + decltype(j)::value]; });
}
template <typename T, int D> MF_DEV void load_col(const T* __restrict__ blk, int rc, T (&v)[D]) {
    sfor<D>([&](auto j) { v[decltype(j)::value] = blk[decltype(j)::value * D + rc]; });
}

// Every kernel below loads the data of step k + 1 BEFORE it works step k (branch-free: clamped indices and flags): with one
// chain per row nothing else hides the ~1-2 us of a dependent global load, and a row's step is now shorter than that.

// ---- Cholesky: up-sweep (par_chol_up_kernel).  REDUCED: the level has future parts (Gf, GU); level 0 has none. ----
template <typename T, int D> struct RowUpStep { T Dn[D], g1[D], g2[D], S[D]; T f2; };
template <typename T, int D, bool REDUCED, bool PF>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, PF)) row_chol_up_kernel(ParLevel<T> in, long B, long len, long P, T* __restrict__ oDv,
                                                           T* __restrict__ oGf, T* __restrict__ oGU, T* __restrict__ oF,
                                                           int* info) {
    const RowChunkId q = row_chunk_id<D>(B, P);
    const long k0 = q.c * len;
    long k1 = k0 + len;
    if (k1 > in.n) k1 = in.n;
    RowFact<T, D> E;
    E.init();
    auto load = [&](long k, RowUpStep<T, D>& d) {
        if (!REDUCED && in.rev) {
            // level 0 of the block-REVERSED matrix (upper_diagonal_lower): position k is block n-1-k, the coupling transposed
            const long kc = k > 0 ? k : 1;
            load_row<T, D>(in.Dv + (q.s * in.n + (in.n - 1 - k)) * D * D, q.rc, d.Dn);
            load_col<T, D>(in.F + (q.s * (in.n - 1) + in.n - 1 - kc) * D * D, q.rc, d.S);
            return;
        }
        load_row<T, D>(in.Dv + (q.s * in.n + k) * D * D, q.rc, d.Dn);
        if constexpr (REDUCED) {
            const bool has2 = k + 1 < in.n;
            load_row<T, D>(in.Gf + (q.s * in.n + k) * D * D, q.rc, d.g1);
            load_row<T, D>(in.GU + (q.s * in.n + (has2 ? k + 1 : k)) * D * D, q.rc, d.g2);
            d.f2 = has2 ? T(1) : T(0);
        }
        const long kc = k > 0 ? k : 1;                               // block 0 has no coupling: clamped, unused
        load_row<T, D>(in.F + (q.s * in.f_stride + kc + in.f_off) * D * D, q.rc, d.S);
    };
    RowUpStep<T, D> cur, nxt;
    if constexpr (PF) load(k0, cur);
    {   // coupling of the chunk's first block to its left neighbour, as columns
        T xc[D];
        const long kc = k0 > 0 ? k0 : 1;
        if (!REDUCED && in.rev) load_row<T, D>(in.F + (q.s * (in.n - 1) + in.n - 1 - kc) * D * D, q.rc, xc);
        else load_col<T, D>(in.F + (q.s * in.f_stride + kc + in.f_off) * D * D, q.rc, xc);
        const T keep = k0 > 0 ? T(1) : T(0);
        sfor<D>([&](auto j) { E.Xa[decltype(j)::value] = xc[decltype(j)::value] * keep; });
    }
    for (long k = k0; k < k1; ++k) {
        asm volatile("s_nop 4");
        if constexpr (PF) load(k + 1 < k1 ? k + 1 : k, nxt); else load(k, cur);
        const bool last = k + 1 == k1;
        T fut[D];
        if constexpr (REDUCED) sfor<D>([&](auto j) { fut[decltype(j)::value] = __builtin_fma(cur.g2[decltype(j)::value], cur.f2, cur.g1[decltype(j)::value]); });
        else sfor<D>([&](auto j) { fut[decltype(j)::value] = T(0); });
        if (last) {
            if (q.valid && q.r < D) sfor<D>([&](auto j) { oGf[q.id * D * D + q.r * D + decltype(j)::value] = fut[decltype(j)::value]; });
        } else if (REDUCED) {
            sfor<D>([&](auto j) { cur.Dn[decltype(j)::value] += fut[decltype(j)::value]; });
        }
        if (k == k0) sfor<D>([&](auto j) { E.Phi[decltype(j)::value] = cur.Dn[decltype(j)::value]; });
        else E.template advance<true, false>(cur.S, cur.Dn, q.r, nullptr, nullptr);
        if constexpr (PF) cur = nxt;
    }
    if (q.valid && q.r < D) {
        sfor<D>([&](auto j) {
            constexpr int jj = decltype(j)::value;
            oDv[q.id * D * D + q.r * D + jj] = E.Phi[jj];
            oGU[q.id * D * D + q.r * D + jj] = E.GU[jj];
            oF[q.id * D * D + jj * D + q.r] = E.Xa[jj];
        });
    }
    if (q.valid && E.bad && info) raise_info(info);
}

// ---- Cholesky: down-sweep on a reduced level (par_chol_down_kernel; also the serial walk of the coarsest level) ----
template <typename T, int D, bool PF>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, PF)) row_chol_down_kernel(ParLevel<T> lv, long B, long len, long P, const T* __restrict__ up,
                                                             T* __restrict__ Pn, int* info) {
    const RowChunkId q = row_chunk_id<D>(B, P);
    const long k0 = q.c * len;
    long k1 = k0 + len;
    if (k1 > lv.n) k1 = lv.n;
    RowFact<T, D> E;
    E.init();
    // step data of block k: its own pivot part, the future parts of block k - 1 (Gf[k-1] + GU[k]), the coupling; block 0: clamped
    auto load = [&](long k, RowUpStep<T, D>& d) {
        const long kc = k > 0 ? k : 1;
        load_row<T, D>(lv.Dv + (q.s * lv.n + k) * D * D, q.rc, d.Dn);
        load_row<T, D>(lv.Gf + (q.s * lv.n + kc - 1) * D * D, q.rc, d.g1);
        load_row<T, D>(lv.GU + (q.s * lv.n + (kc < lv.n ? kc : 0)) * D * D, q.rc, d.g2);
        load_row<T, D>(lv.F + (q.s * lv.f_stride + (kc < lv.n ? kc : 0) + lv.f_off) * D * D, q.rc, d.S);
    };
    RowUpStep<T, D> cur, nxt;
    if constexpr (PF) load(k0, cur);
    if (q.c > 0) load_row<T, D>(up + (q.s * P + q.c - 1) * D * D, q.rc, E.Phi);
    for (long k = k0; k < k1; ++k) {
        asm volatile("s_nop 4");
        if constexpr (PF) load(k + 1 < k1 ? k + 1 : k, nxt); else load(k, cur);
        if (k > 0) {
            // the pivot of block k - 1 at the moment block k is reached: natural-order pivot + its future part
            sfor<D>([&](auto j) { E.Phi[decltype(j)::value] += cur.g1[decltype(j)::value] + cur.g2[decltype(j)::value]; });
            E.template advance<false, false>(cur.S, cur.Dn, q.r, nullptr, nullptr);
        } else {
            sfor<D>([&](auto j) { E.Phi[decltype(j)::value] = cur.Dn[decltype(j)::value]; });
        }
        if (q.valid && q.r < D) sfor<D>([&](auto j) { Pn[(q.s * lv.n + k) * D * D + q.r * D + decltype(j)::value] = E.Phi[decltype(j)::value]; });
        if constexpr (PF) cur = nxt;
    }
    if (q.valid && E.bad && info) raise_info(info);
}

// ---- Cholesky: level 0 emits the factor (par_chol_emit_kernel): chunk c restarts from the pivot of block c len - 1 ----
template <typename T, int D, bool PF>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, PF)) row_chol_emit_kernel(long B, long n, long len, long P, const T* __restrict__ diag,
                                                             const T* __restrict__ sub, const T* __restrict__ up,
                                                             T* __restrict__ ldiag, T* __restrict__ lsub, int* info) {
    const RowChunkId q = row_chunk_id<D>(B, P);
    const long k0 = q.c * len;
    long k1 = k0 + len;
    if (k1 > n) k1 = n;
    RowFact<T, D> E;
    E.init();
    const bool st = q.valid && q.r < D;
    auto lrow = [&](long k) { return st ? ldiag + (q.s * n + k) * D * D + q.r * D : nullptr; };
    auto wrow = [&](long kw) { return st ? lsub + (q.s * (n - 1) + kw) * D * D + q.r * D : nullptr; };
    struct Step { T Dn[D], S[D]; };
    // step k: the pivot part of block k + 1 and the coupling sub[k] of block k + 1 with block k (clamped at the end)
    auto load = [&](long k, Step& d) {
        const long kn = k + 1 < n ? k + 1 : n - 1, ks = k < n - 1 ? k : n - 2;
        load_row<T, D>(diag + (q.s * n + kn) * D * D, q.rc, d.Dn);
        load_row<T, D>(sub + (q.s * (n - 1) + (ks > 0 ? ks : 0)) * D * D, q.rc, d.S);
    };
    Step cur, nxt;
    if constexpr (PF) load(k0, cur);
    {
        T Dn[D];
        load_row<T, D>(diag + (q.s * n + k0) * D * D, q.rc, Dn);
        if (q.c > 0) {
            T S[D];
            load_row<T, D>(up + (q.s * P + q.c - 1) * D * D, q.rc, E.Phi);
            load_row<T, D>(sub + (q.s * (n - 1) + k0 - 1) * D * D, q.rc, S);
            E.template advance<false, true>(S, Dn, q.r, nullptr, wrow(k0 - 1));
        } else {
            sfor<D>([&](auto j) { E.Phi[decltype(j)::value] = Dn[decltype(j)::value]; });
        }
    }
    for (long k = k0; k + 1 < k1; ++k) {
        asm volatile("s_nop 4");
        if constexpr (PF) load(k + 2 < k1 ? k + 1 : k, nxt); else load(k, cur);
        E.template advance<false, true>(cur.S, cur.Dn, q.r, lrow(k), wrow(k));
        if constexpr (PF) cur = nxt;
    }
    E.factor(q.r, lrow(k1 - 1));
    if (q.valid && E.bad && info) raise_info(info);
}

// ---- Solve: affine recursion z_p = M_p z_{p-1} + c_p over positions p (p = k, or n - 1 - k for the transposed solve) ----
// level 0 -> level 1 (par_solve_up0_kernel): the composite map (Pm, q) of every chunk.  Lanes < D hold the columns of Pm, lane D
// holds q: the coupling product and the substitution with the factor act on all D + 1 columns in the same instructions.
template <typename T, int D, bool PF>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, PF)) row_solve_up0_kernel(long Bl, long Br, long n, long len, long P,
                                                             const T* __restrict__ ldiag, const T* __restrict__ lsub,
                                                             const T* __restrict__ rhs, int transpose, T* __restrict__ oM,
                                                             T* __restrict__ oc) {
    using Pp = Dpp<T>;
    const RowChunkId q = row_chunk_id<D>(Br, P);
    const long rr = q.s, s = rr % Bl;
    const long p0 = q.c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    const bool vec = q.r >= D;                                       // lane D (and its idle shadows): the vector column
    struct Step { T Lrow[D], Wrow[D], rv[D], diag; };
    auto load = [&](long p, Step& d) {
        const long k = transpose ? n - 1 - p : p;
        load_row<T, D>(ldiag + (s * n + k) * D * D, q.rc, d.Lrow);
        d.diag = ldiag[(s * n + k) * D * D + q.rc * (D + 1)];
        const T* rv = rhs + (rr * n + k) * D;
        sfor<D>([&](auto i) { d.rv[decltype(i)::value] = rv[decltype(i)::value]; });
        long kw = transpose ? k : k - 1;                             // coupling of position p with p - 1; position 0: clamped, unused
        kw = kw < 0 ? 0 : (kw > n - 2 ? n - 2 : kw);
        load_row<T, D>(lsub + (s * (n - 1) + kw) * D * D, q.rc, d.Wrow);
    };
    Step cur, nxt;
    if constexpr (PF) load(p0, cur);
    T col[D], wc[D];
    sfor<D>([&](auto i) { col[decltype(i)::value] = T(0); wc[decltype(i)::value] = T(0); });
    if (p0 > 0) {
        // the chunk starts from the identity map: Pm = -Wop, i.e. own column of W (forward) or own row (transposed)
        const long k = transpose ? n - 1 - p0 : p0, kw = transpose ? k : k - 1;
        const T* wblk = lsub + (s * (n - 1) + kw) * D * D;
        if (transpose) load_row<T, D>(wblk, q.rc, wc); else load_col<T, D>(wblk, q.rc, wc);
    }
    for (long p = p0; p < p1; ++p) {
        asm volatile("s_nop 4");
        if constexpr (PF) load(p + 1 < p1 ? p + 1 : p, nxt); else load(p, cur);
        T out[D];
        T dinv = t_rcp<T>(cur.diag);
        sfor<D>([&](auto i) { out[decltype(i)::value] = vec ? cur.rv[decltype(i)::value] : T(0); });
        if (p == p0) {
            sfor<D>([&](auto i) { out[decltype(i)::value] = vec ? out[decltype(i)::value] : -wc[decltype(i)::value]; });
        } else {
            fence(cur.Wrow);
            if (!transpose) {
                sfor<D>([&](auto j) {
                    constexpr int jj = decltype(j)::value;
                    sfor<D>([&](auto i) { Pp::template fnmac<decltype(i)::value>(out[decltype(i)::value], cur.Wrow[jj], col[jj]); });
                });
            } else {
                sfor<D>([&](auto j) {
                    constexpr int jj = decltype(j)::value;
                    sfor<D>([&](auto i) { Pp::template fnmac<jj>(out[decltype(i)::value], cur.Wrow[decltype(i)::value], col[jj]); });
                });
            }
        }
        fence(cur.Lrow);
        fence1(dinv);
        if (!transpose) {
            sfor<D>([&](auto kq) {
                constexpr int kk = decltype(kq)::value;
                col[kk] = out[kk] * Pp::template bcast<kk>(dinv);
                sfor2<kk + 1, D>([&](auto i) { Pp::template fnmac<decltype(i)::value>(out[decltype(i)::value], cur.Lrow[kk], col[kk]); });
            });
        } else {
            sfor<D>([&](auto kq) {
                constexpr int kk = D - 1 - decltype(kq)::value;
                col[kk] = out[kk] * Pp::template bcast<kk>(dinv);
                sfor<kk>([&](auto i) { Pp::template fnmac<kk>(out[decltype(i)::value], cur.Lrow[decltype(i)::value], col[kk]); });
            });
        }
        if constexpr (PF) cur = nxt;
    }
    if (q.valid) {
        if (q.r < D) sfor<D>([&](auto i) { oM[q.id * D * D + decltype(i)::value * D + q.r] = col[decltype(i)::value]; });
        else if (q.r == D) sfor<D>([&](auto i) { oc[q.id * D + decltype(i)::value] = col[decltype(i)::value]; });
    }
}

// level l -> level l + 1 (l >= 1): compose explicit maps (par_affine_up_kernel)
template <typename T, int D, bool PF>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, PF)) row_affine_up_kernel(long Br, long n, long len, long P, const T* __restrict__ M,
                                                             const T* __restrict__ cv, T* __restrict__ oM, T* __restrict__ oc) {
    using Pp = Dpp<T>;
    const RowChunkId q = row_chunk_id<D>(Br, P);
    const long rr = q.s;
    const long p0 = q.c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    const bool vec = q.r >= D;
    struct Step { T Mrow[D], cp[D]; };
    auto load = [&](long p, Step& d) {
        load_row<T, D>(M + (rr * n + p) * D * D, q.rc, d.Mrow);
        const T* cp = cv + (rr * n + p) * D;
        sfor<D>([&](auto i) { d.cp[decltype(i)::value] = cp[decltype(i)::value]; });
    };
    Step cur, nxt;
    if constexpr (PF) load(p0 + 1 < p1 ? p0 + 1 : p0, cur);
    T col[D];
    {
        T mc[D];
        load_col<T, D>(M + (rr * n + p0) * D * D, q.rc, mc);
        const T* c0 = cv + (rr * n + p0) * D;
        sfor<D>([&](auto i) { col[decltype(i)::value] = vec ? c0[decltype(i)::value] : mc[decltype(i)::value]; });
    }
    for (long p = p0 + 1; p < p1; ++p) {
        asm volatile("s_nop 4");
        if constexpr (PF) load(p + 1 < p1 ? p + 1 : p, nxt); else load(p, cur);
        T out[D];
        sfor<D>([&](auto i) { out[decltype(i)::value] = vec ? cur.cp[decltype(i)::value] : T(0); });
        fence(cur.Mrow);
        sfor<D>([&](auto j) {
            constexpr int jj = decltype(j)::value;
            sfor<D>([&](auto i) { Pp::template fmac<decltype(i)::value>(out[decltype(i)::value], cur.Mrow[jj], col[jj]); });
        });
        sfor<D>([&](auto i) { col[decltype(i)::value] = out[decltype(i)::value]; });
        if constexpr (PF) cur = nxt;
    }
    if (q.valid) {
        if (q.r < D) sfor<D>([&](auto i) { oM[q.id * D * D + decltype(i)::value * D + q.r] = col[decltype(i)::value]; });
        else if (q.r == D) sfor<D>([&](auto i) { oc[q.id * D + decltype(i)::value] = col[decltype(i)::value]; });
    }
}

// down-sweep on a level >= 1 (and, with len >= n and up = null, the serial walk of the coarsest level): z distributed over
// the lanes (lane i holds z_i), one broadcast-FMA per column of M
template <typename T, int D, bool PF>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, PF)) row_affine_down_kernel(long Br, long n, long len, long P, const T* __restrict__ M,
                                                               const T* __restrict__ cv, const T* __restrict__ up,
                                                               T* __restrict__ Z) {
    using Pp = Dpp<T>;
    const RowChunkId q = row_chunk_id<D>(Br, P);
    const long rr = q.s;
    const long p0 = q.c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    struct Step { T Mrow[D], c; };
    auto load = [&](long p, Step& d) {
        load_row<T, D>(M + (rr * n + p) * D * D, q.rc, d.Mrow);
        d.c = cv[(rr * n + p) * D + q.rc];
    };
    Step cur, nxt;
    if constexpr (PF) load(p0, cur);
    T z = T(0);
    if (q.c > 0) z = up[(rr * P + q.c - 1) * D + q.rc];
    for (long p = p0; p < p1; ++p) {
        asm volatile("s_nop 4");
        if constexpr (PF) load(p + 1 < p1 ? p + 1 : p, nxt); else load(p, cur);
        T acc = cur.c;
        if (p > 0) {
            fence1(z);
            sfor<D>([&](auto j) { Pp::template fmac<decltype(j)::value>(acc, z, cur.Mrow[decltype(j)::value]); });
        }
        z = acc;
        if (q.valid && q.r < D) Z[(rr * n + p) * D + q.r] = z;
        if constexpr (PF) cur = nxt;
    }
}

// level 0 (par_solve_emit_kernel): every chunk redoes its substitution from the known incoming vector and writes the solution;
// z, the right-hand side and the substitution are distributed over the lanes (lane i holds component i)
template <typename T, int D, bool PF>
__global__ void __launch_bounds__(64, row_par_waves(sizeof(T), D, PF)) row_solve_emit_kernel(long Bl, long Br, long n, long len, long P,
                                                              const T* __restrict__ ldiag, const T* __restrict__ lsub,
                                                              const T* __restrict__ rhs, const T* __restrict__ up, int transpose,
                                                              T* __restrict__ outp) {
    using Pp = Dpp<T>;
    const RowChunkId q = row_chunk_id<D>(Br, P);
    const long rr = q.s, s = rr % Bl;
    const long p0 = q.c * len;
    long p1 = p0 + len;
    if (p1 > n) p1 = n;
    struct Step { T Lv[D], Wv[D], diag, x; };
    // own row of L and of W (forward) or own columns (transposed)
    auto load = [&](long p, Step& d) {
        const long k = transpose ? n - 1 - p : p;
        const T* lblk = ldiag + (s * n + k) * D * D;
        long kw = transpose ? k : k - 1;
        kw = kw < 0 ? 0 : (kw > n - 2 ? n - 2 : kw);
        const T* wblk = lsub + (s * (n - 1) + kw) * D * D;
        if (!transpose) { load_row<T, D>(lblk, q.rc, d.Lv); load_row<T, D>(wblk, q.rc, d.Wv); }
        else { load_col<T, D>(lblk, q.rc, d.Lv); load_col<T, D>(wblk, q.rc, d.Wv); }
        d.diag = lblk[q.rc * (D + 1)];
        d.x = rhs[(rr * n + k) * D + q.rc];
    };
    Step cur, nxt;
    if constexpr (PF) load(p0, cur);
    T z = T(0);
    if (q.c > 0) z = up[(rr * P + q.c - 1) * D + q.rc];
    for (long p = p0; p < p1; ++p) {
        asm volatile("s_nop 4");
        if constexpr (PF) load(p + 1 < p1 ? p + 1 : p, nxt); else load(p, cur);
        const long k = transpose ? n - 1 - p : p;
        // entries outside the lower triangle forced to zero
        sfor<D>([&](auto j) {
            constexpr int jj = decltype(j)::value;
            const bool in_tri = transpose ? jj >= q.rc : jj <= q.rc;
            cur.Lv[jj] = in_tri ? cur.Lv[jj] : T(0);
        });
        const T dinv = t_rcp<T>(cur.diag);
        T x = cur.x;
        if (p > 0) {
            fence1(z);
            sfor<D>([&](auto j) { Pp::template fnmac<decltype(j)::value>(x, z, cur.Wv[decltype(j)::value]); });
        }
        T res = T(0);
        if (!transpose) {
            sfor<D>([&](auto kq) {
                constexpr int kk = decltype(kq)::value;
                T xs = x * dinv;
                res = q.rc == kk ? xs : res;
                fence1(xs);
                Pp::template fnmac<kk>(x, xs, cur.Lv[kk]);           // x_i -= L[i][kk] z_kk   (zero above the diagonal)
            });
        } else {
            sfor<D>([&](auto kq) {
                constexpr int kk = D - 1 - decltype(kq)::value;
                T xs = x * dinv;
                res = q.rc == kk ? xs : res;
                fence1(xs);
                Pp::template fnmac<kk>(x, xs, cur.Lv[kk]);           // x_i -= L[kk][i] z_kk   (own column of L)
            });
        }
        z = res;
        if (q.valid && q.r < D) outp[(rr * n + k) * D + q.r] = z;
        if constexpr (PF) cur = nxt;
    }
}

}   // namespace row
}   // namespace mf
