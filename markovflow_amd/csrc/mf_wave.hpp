// Wave kernels: ONE WAVEFRONT per (series, time-chunk) for state dimensions 16 <= d <= 32 - between the row kernels (mf_row.hpp:
// one 16-lane DPP row per chunk, d <= 15) and the LDS-tile engine (mf_big.hpp: one 256-thread workgroup per chunk, built for
// d = 64).  At d = 16 the engine walks every phase of a single 16 x 16 tile with four wavefronts of which one has work, a
// workgroup barrier between phases and every operand in LDS: 18 ms where d = 15 takes 0.96 (profiles/r04_d16_cliff.txt).
//
// Here every DP x DP matrix (DP = 16 NT, NT = 1 or 2; identity-padded beyond d) lives in REGISTERS as NT x NT tiles in the
// accumulator layout of the 16x16x4 matrix-core instruction:
//     lane (r, q) = (lane & 15, lane >> 4), element e of tile (ti, tj)  <->  M[16 ti + row(q, e)][16 tj + r],
//     row(q, e) = q + 4 e (v_mfma_f64_16x16x4_f64), 4 q + e (v_mfma_f32_16x16x4_f32).
// In that layout a tile IS the B operand of the instruction (the K index runs over the rows the lane holds, MFMA e takes
// K = row(q, e)) and the tile of the TRANSPOSE is the A operand, so every product of the form  P^T Q  needs no data movement at
// all:  out(ti, tj) += sum_tk sum_e mfma(P(tk, ti)[e], Q(tk, tj)[e]).  The step of the partitioned elimination (same mathematics
// and RedSys output as kf_chunk_kernel, mf_kernels.hpp; reference kalman_filter.py:184-255, state_space_model.py:431-483,
// block_tri_diag.py:423-436) is arranged so that every product is of that form (Q^-1 symmetric: it is its own transpose):
//     Q^-1 = Ci^T Ci            S = -(Q^-1)^T A           Phi -= A^T S            (coupling S = -Q^-1 A, pivot += A^T Q^-1 A)
//     V  = (LiT)^T X            GU -= V^T V               WT = (LiT)^T S^T        (spike V = Li X;  W = S Li^T)
//     Phi' = Q'^-1 - (WT)^T WT  X' = -(WT)^T V
// with Ci = chol(Q_k)^-1, L = chol(Phi), Li = L^-1; the transposes LiT, S^T are one pass through a wave-private LDS image each.
// The 16 x 16 diagonal tiles are factored / inverted inside the wave: a row per lane, the broadcasts as the DPP row_newbcast
// operand of the consuming v_fmac (mf_row.hpp's primitive; all four 16-lane rows run it redundantly), in and out of the
// accumulator layout through the same LDS image.  No workgroup barrier anywhere; LDS is a staging buffer of NT^2 tile images.
// Vectors are held per lane either by column ("cv": lane (r, q) has v[16 tj + r]) or by row ("rv": v[16 ti + row(q, e)]), the
// two forms a matrix-vector product in this layout consumes and produces.
#pragma once
#include "mf_row.hpp"

namespace mf {
namespace wv {

using row::Dpp;
using row::fence;
using row::fence1;
using row::sfor;
using row::sfor2;

template <typename T> struct Tr;
template <> struct Tr<double> {
    typedef double v4 __attribute__((ext_vector_type(4)));
    typedef double v2 __attribute__((ext_vector_type(2)));
    static constexpr int LD = 18;        // image row stride: rows 16-byte aligned, transposed reads conflict-free
    static MF_DEV constexpr int row(int q, int e) { return q + 4 * e; }
    static MF_DEV v4 mfma(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
};
template <> struct Tr<float> {
    typedef float v4 __attribute__((ext_vector_type(4)));
    typedef float v2 __attribute__((ext_vector_type(2)));
    static constexpr int LD = 20;
    static MF_DEV constexpr int row(int q, int e) { return 4 * q + e; }
    static MF_DEV v4 mfma(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};

// the wave's LDS traffic is ordered by the hardware (one wave, in-order LDS queue); this keeps the COMPILER from moving
// LDS accesses across and waits for the reads / writes issued so far - without touching the vector-memory counter, so the
// next step's global loads stay in flight
// keeps the instruction scheduler from interleaving two phases of a step (their live ranges would add up)
MF_DEV void phase() { __builtin_amdgcn_sched_barrier(0); }
MF_DEV void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct Lane {
    int r, q;
};

enum { S_FULL = 0, S_LOWER = 1, S_UPPER = 2 };
// is tile (i, j) of a matrix with tile structure S non-zero?
template <int S> MF_DEV constexpr bool nz(int i, int j) { return S == S_FULL || (S == S_LOWER ? j <= i : i <= j); }

template <typename T, int NT> struct Mat {
    typename Tr<T>::v4 t[NT][NT];
    MF_DEV void zero() {
        MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j) t[i][j] = typename Tr<T>::v4{0, 0, 0, 0};
    }
};
template <typename T, int NT> struct CV { T v[NT]; };        // lane (r, q): v[tj] = vec[16 tj + r]
template <typename T, int NT> struct RV { T v[NT][4]; };     // lane (r, q): v[ti][e] = vec[16 ti + row(q, e)]

template <typename T> MF_DEV typename Tr<T>::v4 identity_tile(const Lane& ln) {
    typename Tr<T>::v4 t;
    MF_UNROLL for (int e = 0; e < 4; ++e) t[e] = (Tr<T>::row(ln.q, e) == ln.r) ? T(1) : T(0);
    return t;
}

enum { OP_SET = 0, OP_ADD = 1, OP_SUB = 2, OP_NEG = 3 };
// out (OP) P^T Q.  PS / QS: tile structure of P / Q (all-zero tiles are skipped); OS = S_UPPER: the result is symmetric and only
// its tiles ti <= tj are formed.
template <typename T, int NT, int PS, int QS, int OS, int OP>
MF_DEV void tn(Mat<T, NT>& out, const Mat<T, NT>& P, const Mat<T, NT>& Q) {
    using v4 = typename Tr<T>::v4;
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        MF_UNROLL for (int tj = 0; tj < NT; ++tj) {
            if (OS == S_UPPER && ti > tj) continue;
            v4 acc = (OP == OP_ADD) ? out.t[ti][tj] : v4{0, 0, 0, 0};
            MF_UNROLL for (int tk = 0; tk < NT; ++tk) {
                if (!nz<PS>(tk, ti) || !nz<QS>(tk, tj)) continue;
                MF_UNROLL for (int e = 0; e < 4; ++e) acc = Tr<T>::mfma(P.t[tk][ti][e], Q.t[tk][tj][e], acc);
            }
            if (OP == OP_SUB) out.t[ti][tj] -= acc;
            else if (OP == OP_NEG) out.t[ti][tj] = -acc;
            else out.t[ti][tj] = acc;
        }
}

// ---- wave-private LDS image: NT x NT tiles of 16 x LD ------------------------------------------------------------------------
template <typename T> MF_DEV void tile_to_image(const typename Tr<T>::v4& t, T* img, const Lane& ln) {
    MF_UNROLL for (int e = 0; e < 4; ++e) img[Tr<T>::row(ln.q, e) * Tr<T>::LD + ln.r] = t[e];
}
template <typename T> MF_DEV void image_to_tile_t(typename Tr<T>::v4& t, const T* img, const Lane& ln) {
    MF_UNROLL for (int e = 0; e < 4; ++e) t[e] = img[ln.r * Tr<T>::LD + Tr<T>::row(ln.q, e)];
}
// out = in^T (tiles of structure S of `in`; the others of `out` are left alone)
template <typename T, int NT, int S> MF_DEV void transpose(Mat<T, NT>& out, const Mat<T, NT>& in, T* lds, const Lane& ln) {
    constexpr int TS = 16 * Tr<T>::LD;
    lds_fence();
    MF_UNROLL for (int i = 0; i < NT; ++i)
        MF_UNROLL for (int j = 0; j < NT; ++j)
            if (nz<S>(i, j)) tile_to_image<T>(in.t[i][j], lds + (i * NT + j) * TS, ln);
    lds_fence();
    MF_UNROLL for (int i = 0; i < NT; ++i)
        MF_UNROLL for (int j = 0; j < NT; ++j)
            if (nz<S>(i, j)) image_to_tile_t<T>(out.t[j][i], lds + (i * NT + j) * TS, ln);
    lds_fence();
}
template <typename T> MF_DEV void transpose_tile(typename Tr<T>::v4& out, const typename Tr<T>::v4& in, T* img, const Lane& ln) {
    lds_fence();
    tile_to_image<T>(in, img, ln);
    lds_fence();
    image_to_tile_t<T>(out, img, ln);
    lds_fence();
}

// row r of a tile (its lane's 16 values) from the accumulator layout, through the image
template <typename T> MF_DEV void tile_rows(const typename Tr<T>::v4& t, T* img, T (&a)[16], const Lane& ln) {
    using v2 = typename Tr<T>::v2;
    lds_fence();
    tile_to_image<T>(t, img, ln);
    lds_fence();
    MF_UNROLL for (int k = 0; k < 8; ++k) {
        const v2 p = *reinterpret_cast<const v2*>(img + ln.r * Tr<T>::LD + 2 * k);
        a[2 * k] = p[0];
        a[2 * k + 1] = p[1];
    }
    lds_fence();
}
template <typename T> MF_DEV T sel4(int q, T v0, T v1, T v2, T v3) {
    const T lo = (q & 1) ? v1 : v0, hi = (q & 1) ? v3 : v2;
    return (q & 2) ? hi : lo;
}
// x[row] = X[row][r] held by every lane of column r  ->  the tile of X
template <typename T> MF_DEV void cols_to_tile(const T (&x)[16], typename Tr<T>::v4& t, const Lane& ln) {
    MF_UNROLL for (int e = 0; e < 4; ++e)
        t[e] = sel4<T>(ln.q, x[Tr<T>::row(0, e)], x[Tr<T>::row(1, e)], x[Tr<T>::row(2, e)], x[Tr<T>::row(3, e)]);
}

// ---- the 16 x 16 diagonal tiles: a row per lane, DPP row_newbcast operands (all four 16-lane rows redundantly) -------------------
// P (symmetric positive definite tile, lower triangle used) -> Li = chol(P)^-1 as a tile; la picks up the pivots (= diag(L)^2).
template <typename T>
MF_DEV void chol_inv_tile(const typename Tr<T>::v4& P, typename Tr<T>::v4& Li, T* img, const Lane& ln, LogAcc<T>& la, bool& bad) {
    using D = Dpp<T>;
    T a[16], x[16];
    tile_rows<T>(P, img, a, ln);
    sfor<16>([&](auto i) { x[decltype(i)::value] = (ln.r == decltype(i)::value) ? T(1) : T(0); });
    sfor<16>([&](auto j) {
        constexpr int jj = decltype(j)::value;
        fence1(a[jj]);
        const T s = D::template bcast<jj>(a[jj]);
        bad |= !(s > T(0));
        const T inv = row::row_rsqrt(s);
        la.mul(s);
        if constexpr (jj == 7) la.renorm();
        a[jj] *= inv;                  // L[r][j]
        x[jj] *= inv;                  // Li[j][r]
        fence1(a[jj]);
        sfor2<jj + 1, 16>([&](auto k) {
            constexpr int kk = decltype(k)::value;
            D::template fnmac<kk>(a[kk], a[jj], a[jj]);          // P[r][k] -= L[k][j] L[r][j]
            D::template fnmac<kk>(x[kk], a[jj], x[jj]);          // x[k]   -= L[k][j] x[j]
        });
    });
    la.renorm();
    cols_to_tile<T>(x, Li, ln);
}
// C (lower-triangular tile) -> Ci = C^-1 as a tile; la picks up diag(C).
template <typename T>
MF_DEV void tri_inv_tile(const typename Tr<T>::v4& C, typename Tr<T>::v4& Ci, T* img, const Lane& ln, LogAcc<T>& la, bool& bad) {
    using D = Dpp<T>;
    T a[16], x[16], dinv[16];
    tile_rows<T>(C, img, a, ln);
    fence(a);
    sfor<16>([&](auto k) {
        constexpr int kk = decltype(k)::value;
        const T c = D::template bcast<kk>(a[kk]);
        bad |= !(c != T(0));
        dinv[kk] = t_rcp<T>(c);
        la.mul(c);
        if constexpr (kk == 7) la.renorm();
    });
    la.renorm();
    sfor<16>([&](auto i) { x[decltype(i)::value] = (ln.r == decltype(i)::value) ? T(1) : T(0); });
    sfor<16>([&](auto k) {
        constexpr int kk = decltype(k)::value;
        x[kk] *= dinv[kk];
        sfor2<kk + 1, 16>([&](auto i) { D::template fnmac<decltype(i)::value>(x[decltype(i)::value], a[kk], x[kk]); });   // -= C[i][k] x[k]
    });
    cols_to_tile<T>(x, Ci, ln);
}

// ---- several diagonal tiles at once: the four 16-lane rows of the wavefront work on NTL = 1, 2 or 4 DIFFERENT tiles --------------
// (tile i in rows [i * 4 / NTL, (i + 1) * 4 / NTL)): the same instruction stream, a quarter / half of the per-tile cost.  In: the
// tiles in the accumulator layout -> NTL images -> each lane reads its row of ITS tile.  Out: the first row of lanes of every
// tile writes the columns of the inverse into the tile's image, from where every lane reads the inverse - or its transpose,
// whichever the caller multiplies with - back in the accumulator layout.  la / bad: per LANE (the lane's own tile); the caller
// extracts a tile's values from a lane of its rows.
template <int NTL> MF_DEV int tile_of_row(int q) { return NTL == 4 ? q : (NTL == 2 ? (q >> 1) : 0); }
template <typename T, int NTL> MF_DEV void rows_in(const typename Tr<T>::v4 (&t)[NTL], T* img, T (&a)[16], const Lane& ln) {
    using v2 = typename Tr<T>::v2;
    constexpr int TS = 16 * Tr<T>::LD;
    lds_fence();
    MF_UNROLL for (int i = 0; i < NTL; ++i) tile_to_image<T>(t[i], img + i * TS, ln);
    lds_fence();
    const T* mine = img + tile_of_row<NTL>(ln.q) * TS + ln.r * Tr<T>::LD;
    MF_UNROLL for (int k = 0; k < 8; ++k) {
        const v2 p = *reinterpret_cast<const v2*>(mine + 2 * k);
        a[2 * k] = p[0];
        a[2 * k + 1] = p[1];
    }
    lds_fence();
}
// TR: read the transposes
template <typename T, int NTL, bool TR> MF_DEV void cols_out(const T (&x)[16], typename Tr<T>::v4 (&t)[NTL], T* img, const Lane& ln) {
    constexpr int TS = 16 * Tr<T>::LD;
    if ((ln.q & (4 / NTL - 1)) == 0) {                      // the first row of lanes of every tile
        T* mine = img + tile_of_row<NTL>(ln.q) * TS + ln.r;
        MF_UNROLL for (int row = 0; row < 16; ++row) mine[row * Tr<T>::LD] = x[row];
    }
    lds_fence();
    MF_UNROLL for (int i = 0; i < NTL; ++i) {
        if (TR) image_to_tile_t<T>(t[i], img + i * TS, ln);
        else {
            MF_UNROLL for (int e = 0; e < 4; ++e) t[i][e] = img[i * TS + Tr<T>::row(ln.q, e) * Tr<T>::LD + ln.r];
        }
    }
    lds_fence();
}
// P[i] (symmetric positive definite, lower triangle used) -> out[i] = chol(P[i])^-1 (TR: its transpose)
template <typename T, int NTL, bool TR>
MF_DEV void chol_inv_tiles(const typename Tr<T>::v4 (&P)[NTL], typename Tr<T>::v4 (&out)[NTL], T* img, const Lane& ln, LogAcc<T>& la,
                           bool& bad) {
    using D = Dpp<T>;
    T a[16], x[16];
    rows_in<T, NTL>(P, img, a, ln);
    sfor<16>([&](auto i) { x[decltype(i)::value] = (ln.r == decltype(i)::value) ? T(1) : T(0); });
    sfor<16>([&](auto j) {
        constexpr int jj = decltype(j)::value;
        fence1(a[jj]);
        const T s = D::template bcast<jj>(a[jj]);
        bad |= !(s > T(0));
        const T inv = row::row_rsqrt(s);
        la.mul(s);
        if constexpr (jj == 7) la.renorm();
        a[jj] *= inv;
        x[jj] *= inv;
        fence1(a[jj]);
        sfor2<jj + 1, 16>([&](auto k) {
            constexpr int kk = decltype(k)::value;
            D::template fnmac<kk>(a[kk], a[jj], a[jj]);
            D::template fnmac<kk>(x[kk], a[jj], x[jj]);
        });
    });
    la.renorm();
    cols_out<T, NTL, TR>(x, out, img, ln);
}
// C[i] (lower triangular) -> out[i] = C[i]^-1 (TR: its transpose)
template <typename T, int NTL, bool TR>
MF_DEV void tri_inv_tiles(const typename Tr<T>::v4 (&C)[NTL], typename Tr<T>::v4 (&out)[NTL], T* img, const Lane& ln, LogAcc<T>& la,
                          bool& bad) {
    using D = Dpp<T>;
    T a[16], x[16], dinv[16];
    rows_in<T, NTL>(C, img, a, ln);
    fence(a);
    sfor<16>([&](auto k) {
        constexpr int kk = decltype(k)::value;
        const T c = D::template bcast<kk>(a[kk]);
        bad |= !(c != T(0));
        dinv[kk] = t_rcp<T>(c);
        la.mul(c);
        if constexpr (kk == 7) la.renorm();
    });
    la.renorm();
    sfor<16>([&](auto i) { x[decltype(i)::value] = (ln.r == decltype(i)::value) ? T(1) : T(0); });
    sfor<16>([&](auto k) {
        constexpr int kk = decltype(k)::value;
        x[kk] *= dinv[kk];
        sfor2<kk + 1, 16>([&](auto i) { D::template fnmac<decltype(i)::value>(x[decltype(i)::value], a[kk], x[kk]); });
    });
    cols_out<T, NTL, TR>(x, out, img, ln);
}
// the value a per-lane accumulator holds for tile i of NTL (a lane of the tile's first row)
template <typename T, int NTL> MF_DEV T of_tile(T v, int i) { return __shfl(v, 16 * (i * (4 / NTL)), 64); }

// C (lower, tiles 00 [10 11]) -> Ci = C^-1 (S_LOWER).  NT = 2: c10t = (C10)^T, which the caller reads transposed from memory; the
// two diagonal tiles are inverted side by side (two 16-lane rows each).  la: per lane, the diagonal of the lane's own tile.
template <typename T, int NT>
MF_DEV void tri_inv_mat(const Mat<T, NT>& C, const typename Tr<T>::v4& c10t, Mat<T, NT>& Ci, T* lds, const Lane& ln, LogAcc<T>& la,
                        bool& bad) {
    using v4 = typename Tr<T>::v4;
    if constexpr (NT == 1) {
        const v4 in[1] = {C.t[0][0]};
        v4 out[1];
        tri_inv_tiles<T, 1, false>(in, out, lds, ln, la, bad);
        Ci.t[0][0] = out[0];
    } else {
        const v4 in[2] = {C.t[0][0], C.t[1][1]};
        v4 out[2], cit11;
        tri_inv_tiles<T, 2, false>(in, out, lds, ln, la, bad);
        image_to_tile_t<T>(cit11, lds + 16 * Tr<T>::LD, ln);                                    // Ci11^T: the image is still there
        lds_fence();
        Ci.t[0][0] = out[0];
        Ci.t[1][1] = out[1];
        // Ci10 = -Ci11 (C10 Ci00): both products in the P^T Q form
        v4 g = {0, 0, 0, 0}, h = {0, 0, 0, 0};
        MF_UNROLL for (int e = 0; e < 4; ++e) g = Tr<T>::mfma(c10t[e], Ci.t[0][0][e], g);           // C10 Ci00
        MF_UNROLL for (int e = 0; e < 4; ++e) h = Tr<T>::mfma(cit11[e], g[e], h);                   // Ci11 (C10 Ci00)
        Ci.t[1][0] = -h;
        Ci.t[0][1] = v4{0, 0, 0, 0};
    }
}
// log of the product a per-lane accumulator of tri_inv_mat holds, over all NT diagonal tiles
template <typename T, int NT> MF_DEV T tri_logdet(const LogAcc<T>& la) {
    const T v = la.value();
    if constexpr (NT == 1) return v;
    else return of_tile<T, 2>(v, 0) + of_tile<T, 2>(v, 1);
}
// Phi (symmetric, tiles ti <= tj valid) -> LiT = (chol(Phi)^-1)^T (S_UPPER); Phi's tile (1,1) is consumed
template <typename T, int NT>
MF_DEV void chol_inv_mat(Mat<T, NT>& Phi, Mat<T, NT>& LiT, T* lds, const Lane& ln, LogAcc<T>& la, bool& bad) {
    using v4 = typename Tr<T>::v4;
    {
        const v4 in[1] = {Phi.t[0][0]};
        v4 out[1];
        chol_inv_tiles<T, 1, true>(in, out, lds, ln, la, bad);
        LiT.t[0][0] = out[0];
    }
    if constexpr (NT == 2) {
        v4 li00, lt01 = {0, 0, 0, 0}, acc = {0, 0, 0, 0}, z = {0, 0, 0, 0}, h = {0, 0, 0, 0};
        MF_UNROLL for (int e = 0; e < 4; ++e) li00[e] = lds[Tr<T>::row(ln.q, e) * Tr<T>::LD + ln.r];            // L00^-1: the image is still there
        lds_fence();
        MF_UNROLL for (int e = 0; e < 4; ++e) lt01 = Tr<T>::mfma(LiT.t[0][0][e], Phi.t[0][1][e], lt01);      // (L10)^T = Li00 Phi01
        MF_UNROLL for (int e = 0; e < 4; ++e) acc = Tr<T>::mfma(lt01[e], lt01[e], acc);                      // L10 L10^T
        Phi.t[1][1] -= acc;
        {
            const v4 in[1] = {Phi.t[1][1]};
            v4 out[1];
            chol_inv_tiles<T, 1, true>(in, out, lds, ln, la, bad);
            LiT.t[1][1] = out[0];
        }
        MF_UNROLL for (int e = 0; e < 4; ++e) z = Tr<T>::mfma(lt01[e], li00[e], z);                          // Z = L10 Li00
        MF_UNROLL for (int e = 0; e < 4; ++e) h = Tr<T>::mfma(z[e], LiT.t[1][1][e], h);                      // Z^T Li11^T
        LiT.t[0][1] = -h;
        LiT.t[1][0] = v4{0, 0, 0, 0};
    }
}

// ---- vectors -------------------------------------------------------------------------------------------------------------------
template <typename T> MF_DEV T xor_rows(T x) {        // sum over the four 16-lane rows (same r)
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}
template <typename T> MF_DEV T sum16(T x) {           // sum over the sixteen lanes of a row
    x += __shfl_xor(x, 1, 64);
    x += __shfl_xor(x, 2, 64);
    x += __shfl_xor(x, 4, 64);
    x += __shfl_xor(x, 8, 64);
    return x;
}
// y = M^T v
template <typename T, int NT, int S> MF_DEV void tn_mv(CV<T, NT>& y, const Mat<T, NT>& M, const RV<T, NT>& v) {
    MF_UNROLL for (int tj = 0; tj < NT; ++tj) {
        T acc = T(0);
        MF_UNROLL for (int ti = 0; ti < NT; ++ti) {
            if (!nz<S>(ti, tj)) continue;
            MF_UNROLL for (int e = 0; e < 4; ++e) acc = __builtin_fma(M.t[ti][tj][e], v.v[ti][e], acc);
        }
        y.v[tj] = xor_rows<T>(acc);
    }
}
template <typename T, int NT> MF_DEV void cv_to_rv(RV<T, NT>& out, const CV<T, NT>& in, const Lane& ln) {
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        MF_UNROLL for (int e = 0; e < 4; ++e) out.v[ti][e] = __shfl(in.v[ti], 16 * ln.q + Tr<T>::row(ln.q, e), 64);
}
template <typename T, int NT> MF_DEV T dot_cv(const CV<T, NT>& a, const CV<T, NT>& b) {
    T s = T(0);
    MF_UNROLL for (int tj = 0; tj < NT; ++tj) s = __builtin_fma(a.v[tj], b.v[tj], s);
    return s;
}

// ---- global memory <-> the accumulator layout ----------------------------------------------------------------------------------
// g: d x d row-major.  lower: the strict upper triangle reads as zero; idpad: ones on the padded diagonal.
// EX: d == 16 NT exactly - no padding, every bound check folds away.
template <typename T, int NT, int S, bool EX = false>
MF_DEV void load_mat(Mat<T, NT>& m, const T* __restrict__ g, int d, bool lower, bool idpad, const Lane& ln) {
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        MF_UNROLL for (int tj = 0; tj < NT; ++tj) {
            if (!nz<S>(ti, tj)) { m.t[ti][tj] = typename Tr<T>::v4{0, 0, 0, 0}; continue; }
            MF_UNROLL for (int e = 0; e < 4; ++e) {
                const int i = 16 * ti + Tr<T>::row(ln.q, e), j = 16 * tj + ln.r;
                const bool in = (EX || (i < d && j < d)) && (!lower || j <= i);
                const T v = g[in ? i * d + j : 0];
                m.t[ti][tj][e] = in ? v : ((!EX && idpad && i == j && i >= d) ? T(1) : T(0));
            }
        }
}
// the same from a matrix stored TRANSPOSED in memory is never needed: F of a reduced system is read through load_mat_t
template <typename T, int NT>
MF_DEV void load_mat_t(Mat<T, NT>& m, const T* __restrict__ g, int d, const Lane& ln) {      // m = g^T
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        MF_UNROLL for (int tj = 0; tj < NT; ++tj)
            MF_UNROLL for (int e = 0; e < 4; ++e) {
                const int i = 16 * ti + Tr<T>::row(ln.q, e), j = 16 * tj + ln.r;
                const bool in = i < d && j < d;
                const T v = g[in ? j * d + i : 0];
                m.t[ti][tj][e] = in ? v : T(0);
            }
}
// tile (ti, tj) of a d x d row-major matrix, TRANSPOSED (t = g[16 ti ..][16 tj ..]^T)
template <typename T, bool EX = false>
MF_DEV void load_tile_t(typename Tr<T>::v4& t, const T* __restrict__ g, int d, int ti, int tj, const Lane& ln) {
    MF_UNROLL for (int e = 0; e < 4; ++e) {
        const int i = 16 * ti + ln.r, j = 16 * tj + Tr<T>::row(ln.q, e);          // element (j', i') of the transposed tile
        const bool in = EX || (i < d && j < d);
        const T v = g[in ? i * d + j : 0];
        t[e] = in ? v : T(0);
    }
}
// the d x d corner of a matrix; SYM: tiles ti <= tj hold a symmetric matrix, the lower off-diagonal tile comes through LDS
template <typename T, int NT, bool SYM, bool EX = false>
MF_DEV void store_mat(T* __restrict__ g, Mat<T, NT>& m, int d, T* lds, const Lane& ln) {
    if constexpr (SYM && NT == 2) transpose_tile<T>(m.t[1][0], m.t[0][1], lds, ln);
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        MF_UNROLL for (int tj = 0; tj < NT; ++tj)
            MF_UNROLL for (int e = 0; e < 4; ++e) {
                const int i = 16 * ti + Tr<T>::row(ln.q, e), j = 16 * tj + ln.r;
                if (EX || (i < d && j < d)) g[i * d + j] = m.t[ti][tj][e];
            }
}
template <typename T, int NT> MF_DEV void load_cv(CV<T, NT>& v, const T* __restrict__ g, int d, const Lane& ln) {
    MF_UNROLL for (int tj = 0; tj < NT; ++tj) {
        const int j = 16 * tj + ln.r;
        const T x = g[j < d ? j : 0];
        v.v[tj] = j < d ? x : T(0);
    }
}
template <typename T, int NT> MF_DEV void load_rv(RV<T, NT>& v, const T* __restrict__ g, int d, const Lane& ln) {
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        MF_UNROLL for (int e = 0; e < 4; ++e) {
            const int i = 16 * ti + Tr<T>::row(ln.q, e);
            const T x = g[i < d ? i : 0];
            v.v[ti][e] = i < d ? x : T(0);
        }
}
template <typename T, int NT> MF_DEV void store_cv(T* __restrict__ g, const CV<T, NT>& v, int d, const Lane& ln) {
    MF_UNROLL for (int tj = 0; tj < NT; ++tj) {
        const int j = 16 * tj + ln.r;
        if (ln.q == 0 && j < d) g[j] = v.v[tj];
    }
}

// ---- the elimination state of one chunk (Elim of mf_kernels.hpp on register tiles) ----------------------------------------------
// what an elimination leaves for the step that follows it (declared per step by the caller: as members of the chunk's state these
// values - written and read under "this step eliminates" - would be carried around the step loop on the other path, 24 registers)
template <typename T, int NT> struct WaveFact {
    Mat<T, NT> LiT, V;
    RV<T, NT> z_rv;
};
template <typename T, int NT> struct WaveElim {
    Mat<T, NT> Phi;      // symmetric (tiles ti <= tj): pivot of the current block
    Mat<T, NT> X;        // coupling current block <-> the chunk's left separator
    Mat<T, NT> GU;       // symmetric: accumulated contribution to the separator's pivot
    CV<T, NT> t, gU;     // right-hand side of the current block; contribution to the separator's
    T quad;              // per lane: sum over its columns of z^2 (summed over the lanes of a row at the end)
    LogAcc<T> laL;       // pivots of the eliminated blocks
    bool bad;

    MF_DEV void init() {
        Phi.zero(); X.zero(); GU.zero();
        MF_UNROLL for (int j = 0; j < NT; ++j) { t.v[j] = T(0); gU.v[j] = T(0); }
        quad = T(0);
        laL.init();
        bad = false;
    }
    // Factor the complete pivot in Phi, z = L^-1 t, spike V = L^-1 X folded into the separator.
    template <bool SPIKE> MF_DEV void eliminate(WaveFact<T, NT>& f, T* lds, const Lane& ln) {
        chol_inv_mat<T, NT>(Phi, f.LiT, lds, ln, laL, bad);
        phase();
        after_factor<SPIKE>(f, ln);
    }
    // ... with f.LiT already in place (the paired pass of wave_kf_pair_kernel factors the pivot beside the next chol(Q))
    template <bool SPIKE> MF_DEV void after_factor(WaveFact<T, NT>& f, const Lane& ln) {
        RV<T, NT> t_rv;
        cv_to_rv<T, NT>(t_rv, t, ln);
        CV<T, NT> z;
        tn_mv<T, NT, S_UPPER>(z, f.LiT, t_rv);                    // z = Li t
        quad += dot_cv<T, NT>(z, z);
        cv_to_rv<T, NT>(f.z_rv, z, ln);
        if constexpr (SPIKE) {
            tn<T, NT, S_UPPER, S_FULL, S_FULL, OP_SET>(f.V, f.LiT, X);        // V = Li X
            tn<T, NT, S_FULL, S_FULL, S_UPPER, OP_SUB>(GU, f.V, f.V);         // GU -= V^T V
            CV<T, NT> vz;
            tn_mv<T, NT, S_FULL>(vz, f.V, f.z_rv);
            MF_UNROLL for (int j = 0; j < NT; ++j) gU.v[j] -= vz.v[j];
        }
    }
    // After eliminate(): the next block couples to the eliminated one through W with WT = W^T given; Dn / rn are its own parts.
    template <bool SPIKE> MF_DEV void advance(const WaveFact<T, NT>& f, const Mat<T, NT>& WT, const Mat<T, NT>& Dn, const CV<T, NT>& rn) {
        CV<T, NT> wz;
        tn_mv<T, NT, S_FULL>(wz, WT, f.z_rv);                                 // W z
        MF_UNROLL for (int j = 0; j < NT; ++j) t.v[j] = rn.v[j] - wz.v[j];
        MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = i; j < NT; ++j) Phi.t[i][j] = Dn.t[i][j];
        tn<T, NT, S_FULL, S_FULL, S_UPPER, OP_SUB>(Phi, WT, WT);              // Phi = Dn - W W^T
        if constexpr (SPIKE) tn<T, NT, S_FULL, S_FULL, S_FULL, OP_NEG>(X, WT, f.V);   // X = -W V
    }
};

template <typename T, int NT>
MF_DEV void store_chunk_wave(const RedSys<T>& out, long idx, int d, WaveElim<T, NT>& E, T scalar, T* lds, const Lane& ln) {
    const long dd = long(d) * d;
    store_mat<T, NT, true>(out.Dv + idx * dd, E.Phi, d, lds, ln);
    store_mat<T, NT, true>(out.GU + idx * dd, E.GU, d, lds, ln);
    store_mat<T, NT, false>(out.F + idx * dd, E.X, d, lds, ln);
    store_cv<T, NT>(out.tv + idx * d, E.t, d, ln);
    store_cv<T, NT>(out.gU + idx * d, E.gU, d, ln);
    if (ln.r == 0 && ln.q == 0) out.sc[idx] = scalar;
}

template <typename T> struct WvArgs {
    long B, Tn;
    int d, m;
    const T *mu0, *cholP0, *A, *b, *cholQ, *H, *y, *Rinv;
    int rinv_per_step;
    long P, L;          // chunks per series, transitions per chunk
    int* info;
};

constexpr int WV_MAXM = 4;

// observation rows of one block in both vector forms (M rows; rows >= m read as zero)
template <typename T, int NT, int M> struct ObsRows {
    CV<T, NT> hc[M];
    RV<T, NT> hr[M];
    T y[M];
    MF_DEV void load(const T* __restrict__ Hk, const T* __restrict__ yk, int d, int m, const Lane& ln) {
        MF_UNROLL for (int o = 0; o < M; ++o) {
            const bool on = o < m;
            load_cv<T, NT>(hc[o], Hk + (on ? o * d : 0), on ? d : 0, ln);
            load_rv<T, NT>(hr[o], Hk + (on ? o * d : 0), on ? d : 0, ln);
            const T v = yk[on ? o : 0];
            y[o] = on ? v : T(0);
        }
    }
};
// Dn += H^T R^-1 H (tiles ti <= tj), rn += H^T R^-1 y; returns y^T R^-1 y.  Ri: M x M, zero beyond m.
template <typename T, int NT, int M>
MF_DEV T obs_apply(const ObsRows<T, NT, M>& ob, const T (&Ri)[M][M], Mat<T, NT>& Dn, CV<T, NT>& rn) {
    T ry[M], yry = T(0);
    MF_UNROLL for (int o = 0; o < M; ++o) {
        ry[o] = T(0);
        MF_UNROLL for (int p = 0; p < M; ++p) ry[o] = __builtin_fma(Ri[o][p], ob.y[p], ry[o]);
        yry = __builtin_fma(ob.y[o], ry[o], yry);
    }
    MF_UNROLL for (int o = 0; o < M; ++o) {
        RV<T, NT> g;                                         // row o of R^-1 H, by row index
        MF_UNROLL for (int ti = 0; ti < NT; ++ti)
            MF_UNROLL for (int e = 0; e < 4; ++e) {
                T acc = T(0);
                MF_UNROLL for (int p = 0; p < M; ++p) acc = __builtin_fma(Ri[o][p], ob.hr[p].v[ti][e], acc);
                g.v[ti][e] = acc;
            }
        MF_UNROLL for (int ti = 0; ti < NT; ++ti)
            MF_UNROLL for (int tj = ti; tj < NT; ++tj)
                MF_UNROLL for (int e = 0; e < 4; ++e) Dn.t[ti][tj][e] = __builtin_fma(g.v[ti][e], ob.hc[o].v[tj], Dn.t[ti][tj][e]);
        MF_UNROLL for (int tj = 0; tj < NT; ++tj) rn.v[tj] = __builtin_fma(ob.hc[o].v[tj], ry[o], rn.v[tj]);
    }
    return yry;
}
template <typename T, int M> MF_DEV void load_rinv(T (&Ri)[M][M], const T* __restrict__ R, int m) {
    MF_UNROLL for (int o = 0; o < M; ++o)
        MF_UNROLL for (int p = 0; p < M; ++p) {
            const bool on = o < m && p < m;
            const T v = R[on ? o * m + p : 0];
            Ri[o][p] = on ? v : T(0);
        }
}

// Level 0: wavefront (s, c) eliminates the transitions [c L, min((c+1) L, T-1)) of series s (the partition and the reduced system
// of big_kf_chunk_kernel, mf_big_impl.hpp: the levels behind it do not know which kernel produced their input).
// WPE: wavefronts per SIMD the register allocation is held to (the diagonal-tile chains are fp64 / fp32 VALU work that one
// wavefront alone issues at a third of the rate two or more reach together: scripts/micro/dpp_f64_rate.hip)
template <typename T, int NT, int M, int WPE, bool EX>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) wave_kf_chunk_kernel(WvArgs<T> a, RedSys<T> out) {
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * 16 * Tr<T>::LD];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long id = blockIdx.x, s = id / a.P, c = id % a.P;
    int d = EX ? 16 * NT : a.d;          // EX: the state dimension fills its tiles exactly (d = 16, 32): no padding, no bound checks
    const int m = a.m;
    const long nt = a.Tn - 1, tau0 = c * a.L;
    long len = nt - tau0;
    if (len > a.L) len = a.L;
    if (len < 0) len = 0;
    const bool spike = c > 0;
    const long dd = long(d) * d;

    WaveElim<T, NT> E;
    E.init();
    LogAcc<T> laC;
    laC.init();
    T acc_ww = T(0), acc_yry = T(0);          // per lane (ww: own columns; yry: uniform)
    T Ri[M][M];
    if (!a.rinv_per_step) load_rinv<T, M>(Ri, a.Rinv, m);

    Mat<T, NT> C, Am, Dn;
    typename Tr<T>::v4 c10t = {0, 0, 0, 0};
    CV<T, NT> rn, mv_cv;
    RV<T, NT> mv_rv, rn0_rv;
    ObsRows<T, NT, M> ob;

    // the block's own terms from its Cholesky factor C: Dn = Q^-1 (symmetric, BOTH off-diagonal tiles: it is an operand of the
    // coupling), rn = Q^-1 mvec (also by row: rn0_rv); then the observation terms on top (kalman_filter.py:86-101)
    auto own_terms = [&](long blk) {
        {
            Mat<T, NT> Ci;
            tri_inv_mat<T, NT>(C, c10t, Ci, lds, ln, laC, E.bad);
            tn<T, NT, S_LOWER, S_LOWER, S_UPPER, OP_SET>(Dn, Ci, Ci);           // Q^-1 = Ci^T Ci
        }
        if constexpr (NT == 2) transpose_tile<T>(Dn.t[1][0], Dn.t[0][1], lds, ln);
        tn_mv<T, NT, S_FULL>(rn, Dn, mv_rv);                                    // Q^-1 mvec
        acc_ww += dot_cv<T, NT>(rn, mv_cv);                                     // |C^-1 mvec|^2 = mvec^T Q^-1 mvec
        cv_to_rv<T, NT>(rn0_rv, rn, ln);
    };
    auto obs_terms = [&](long blk) {
        if (a.rinv_per_step) load_rinv<T, M>(Ri, a.Rinv + (s * a.Tn + blk) * m * m, m);
        acc_yry += obs_apply<T, NT, M>(ob, Ri, Dn, rn);
    };

    if (c == 0) {   // block 0: the prior
        load_mat<T, NT, S_LOWER, EX>(C, a.cholP0 + s * dd, d, true, true, ln);
        if constexpr (NT == 2) load_tile_t<T, EX>(c10t, a.cholP0 + s * dd, d, 1, 0, ln);
        load_rv<T, NT>(mv_rv, a.mu0 + s * d, d, ln);
        load_cv<T, NT>(mv_cv, a.mu0 + s * d, d, ln);
        ob.load(a.H + (s * a.Tn) * m * d, a.y + (s * a.Tn) * m, d, m, ln);
        own_terms(0);
        obs_terms(0);
        MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = i; j < NT; ++j) E.Phi.t[i][j] = Dn.t[i][j];
        E.t = rn;
    }
    Mat<T, NT> Cn;
    typename Tr<T>::v4 c10tn = {0, 0, 0, 0};
    if (len > 0) {
        load_mat<T, NT, S_LOWER, EX>(Cn, a.cholQ + (s * nt + tau0) * dd, d, true, true, ln);
        if constexpr (NT == 2) load_tile_t<T, EX>(c10tn, a.cholQ + (s * nt + tau0) * dd, d, 1, 0, ln);
    }
    for (long j = 0; j < len; ++j) {
        const long tau = tau0 + j, blk = tau + 1;
        // (the lane coordinates and d are made opaque once per step: everything derived from them - load offsets, padding masks,
        // the unit vectors of the substitutions - would otherwise be hoisted out of the loop and live, or spill, across it)
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        if constexpr (!EX) asm volatile("" : "+s"(d));
        // the step opens with the inversion of chol(Q): its factor was fetched one step ahead (everything else is first used a few
        // microseconds into the step, behind that inversion, and is simply loaded here)
        C = Cn;
        c10t = c10tn;
        {
            const long tn = (j + 1 < len) ? tau + 1 : tau;
            load_mat<T, NT, S_LOWER, EX>(Cn, a.cholQ + (s * nt + tn) * dd, d, true, true, ln);
            if constexpr (NT == 2) load_tile_t<T, EX>(c10tn, a.cholQ + (s * nt + tn) * dd, d, 1, 0, ln);
        }
        load_mat<T, NT, S_FULL, EX>(Am, a.A + (s * nt + tau) * dd, d, false, false, ln);
        load_rv<T, NT>(mv_rv, a.b + (s * nt + tau) * d, d, ln);
        load_cv<T, NT>(mv_cv, a.b + (s * nt + tau) * d, d, ln);
        ob.load(a.H + (s * a.Tn + blk) * m * d, a.y + (s * a.Tn + blk) * m, d, m, ln);
        own_terms(blk);
        phase();
        Mat<T, NT> S;
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_NEG>(S, Dn, Am);                   // S = -Q^-1 A: the coupling to block k - 1
        CV<T, NT> btw;
        tn_mv<T, NT, S_FULL>(btw, Am, rn0_rv);                                  // A^T Q^-1 mvec
        obs_terms(blk);
        phase();
        if (j == 0 && spike) {
            // the block on the left is the chunk's separator: its coupling seeds the spike
            tn<T, NT, S_FULL, S_FULL, S_UPPER, OP_NEG>(E.GU, Am, S);            // GU = A^T Q^-1 A
            E.X = S;
            MF_UNROLL for (int k = 0; k < NT; ++k) E.gU.v[k] = -btw.v[k];
            MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int k = i; k < NT; ++k) E.Phi.t[i][k] = Dn.t[i][k];
            E.t = rn;
        } else {
            tn<T, NT, S_FULL, S_FULL, S_UPPER, OP_SUB>(E.Phi, Am, S);           // D_{k-1} += A^T Q^-1 A: complete
            MF_UNROLL for (int k = 0; k < NT; ++k) E.t.v[k] -= btw.v[k];
            phase();
            WaveFact<T, NT> f;
            if (spike) E.template eliminate<true>(f, lds, ln); else E.template eliminate<false>(f, lds, ln);
            phase();
            Mat<T, NT> ST, WT;
            transpose<T, NT, S_FULL>(ST, S, lds, ln);
            tn<T, NT, S_UPPER, S_FULL, S_FULL, OP_SET>(WT, f.LiT, ST);          // W^T = Li S^T
            phase();
            if (spike) E.template advance<true>(f, WT, Dn, rn); else E.template advance<false>(f, WT, Dn, rn);
        }
    }
    const T ww = sum16<T>(acc_ww), quad = sum16<T>(E.quad);
    const T scalar = T(-0.5) * (acc_yry + ww) + T(0.5) * quad - tri_logdet<T, NT>(laC) - T(0.5) * E.laL.value();
    store_chunk_wave<T, NT>(out, id, d, E, scalar, lds, ln);
    if (__any(E.bad) && threadIdx.x == 0 && a.info) raise_info(a.info);
}

// ---- d <= 16: the step's two factorisations in ONE pass ------------------------------------------------------------------------
// A step of wave_kf_chunk_kernel factors two diagonal tiles one after the other - chol(Q_k)^-1 when it opens and the pivot's Cholesky
// factor half way through - and each pass runs redundantly on all four 16-lane rows: two thirds of the step's ~1 050 vector
// instructions.  The inverse of the NEXT transition's chol(Q) depends on nothing of this step, so here the rows [0, 2) of the
// wavefront factor the pivot while the rows [2, 4) invert chol(Q_{k+1}): one instruction stream, the roles told apart by three
// per-lane constants (the tri rows see a pivot of 1 - rsq gives 1 - and their reciprocal diagonal as the scale of the inverse's row;
// the Cholesky update of the remaining columns is multiplied by 0 there).
// P (SPD tile, lower triangle used) -> LiT = chol(P)^-T;  C (lower-triangular tile) -> Ci = C^-1.  la (per lane): the pivots of P in
// the rows [0, 2), the diagonal of C in the rows [2, 4).  img: two tile images.
template <typename T>
MF_DEV void chol_tri_pair(const typename Tr<T>::v4& P, const typename Tr<T>::v4& C, typename Tr<T>::v4& LiT, typename Tr<T>::v4& Ci,
                          T* img, const Lane& ln, LogAcc<T>& la, bool& bad) {
    using D = Dpp<T>;
    using v4 = typename Tr<T>::v4;
    constexpr int TS = 16 * Tr<T>::LD;
    T a[16], x[16];
    const v4 in[2] = {P, C};
    rows_in<T, 2>(in, img, a, ln);
    const bool chol = ln.q < 2;
    const T dg = img[(chol ? 0 : TS) + ln.r * Tr<T>::LD + ln.r];          // the lane's diagonal element (the images are still there)
    lds_fence();
    bad |= !chol && !(dg != T(0));
    const T dinv = chol ? T(1) : t_rcp<T>(dg);
    const T cm = chol ? T(1) : T(0);
    sfor<16>([&](auto i) { x[decltype(i)::value] = (ln.r == decltype(i)::value) ? T(1) : T(0); });
    sfor<16>([&](auto j) {
        constexpr int jj = decltype(j)::value;
        fence1(a[jj]);
        const T s = D::template bcast<jj>(a[jj]);
        bad |= chol && !(s > T(0));
        la.mul(s);
        if constexpr (jj == 7) la.renorm();
        const T inv = row::row_rsqrt(chol ? s : T(1));
        T dj = dinv;
        fence1(dj);
        const T xs = inv * D::template bcast<jj>(dj);
        a[jj] *= inv;                  // L[r][j]  |  C[r][j]
        x[jj] *= xs;                   // Li[j][r] |  Ci[j][r]
        T am = a[jj] * cm;
        fence1(a[jj]);
        fence1(am);
        sfor2<jj + 1, 16>([&](auto k) {
            constexpr int kk = decltype(k)::value;
            D::template fnmac<kk>(a[kk], a[jj], am);             // P[r][k] -= L[k][j] L[r][j]   (nothing in the tri rows)
            D::template fnmac<kk>(x[kk], a[jj], x[jj]);          // x[k]   -= L[k][j] x[j]       (C[k][j] there)
        });
    });
    la.renorm();
    if ((ln.q & 1) == 0) {                                       // the first row of lanes of either tile: the columns of the inverse
        T* mine = img + (chol ? 0 : TS) + ln.r;
        MF_UNROLL for (int row = 0; row < 16; ++row) mine[row * Tr<T>::LD] = x[row];
    }
    lds_fence();
    image_to_tile_t<T>(LiT, img, ln);
    MF_UNROLL for (int e = 0; e < 4; ++e) Ci[e] = img[TS + Tr<T>::row(ln.q, e) * Tr<T>::LD + ln.r];
    lds_fence();
}

// The same for 2 x 2 tiles (16 < d <= 32), where a step has THREE passes (both diagonal tiles of chol(Q) side by side, then the
// pivot's two diagonal tiles one after the other): the first pass takes the pivot's tile (0, 0) in the first row of lanes and the two
// tiles of the next chol(Q) in the second and third.  NTL tiles, 4 / NTL rows of lanes each; `chol`: the lane's tile is factored
// (SPD, lower triangle used), else inverted (lower triangular).  The inverses are left in the NTL images, row-major.
template <typename T, int NTL>
MF_DEV void chol_tri_pass(const typename Tr<T>::v4 (&in)[NTL], bool chol, T* img, const Lane& ln, LogAcc<T>& la, bool& bad) {
    using D = Dpp<T>;
    constexpr int TS = 16 * Tr<T>::LD;
    T a[16], x[16];
    rows_in<T, NTL>(in, img, a, ln);
    const int tile = tile_of_row<NTL>(ln.q);
    const T dg = img[tile * TS + ln.r * Tr<T>::LD + ln.r];
    lds_fence();
    bad |= !chol && !(dg != T(0));
    const T dinv = chol ? T(1) : t_rcp<T>(dg);
    const T cm = chol ? T(1) : T(0);
    sfor<16>([&](auto i) { x[decltype(i)::value] = (ln.r == decltype(i)::value) ? T(1) : T(0); });
    sfor<16>([&](auto j) {
        constexpr int jj = decltype(j)::value;
        fence1(a[jj]);
        const T s = D::template bcast<jj>(a[jj]);
        bad |= chol && !(s > T(0));
        la.mul(s);
        if constexpr (jj == 7) la.renorm();
        const T inv = row::row_rsqrt(chol ? s : T(1));
        T dj = dinv;
        fence1(dj);
        const T xs = inv * D::template bcast<jj>(dj);
        a[jj] *= inv;
        x[jj] *= xs;
        T am = a[jj] * cm;
        fence1(a[jj]);
        fence1(am);
        sfor2<jj + 1, 16>([&](auto k) {
            constexpr int kk = decltype(k)::value;
            D::template fnmac<kk>(a[kk], a[jj], am);
            D::template fnmac<kk>(x[kk], a[jj], x[jj]);
        });
    });
    la.renorm();
    if ((ln.q & (4 / NTL - 1)) == 0) {
        T* mine = img + tile * TS + ln.r;
        MF_UNROLL for (int row = 0; row < 16; ++row) mine[row * Tr<T>::LD] = x[row];
    }
    lds_fence();
}
template <typename T> MF_DEV void image_to_tile(typename Tr<T>::v4& t, const T* img, const Lane& ln) {
    MF_UNROLL for (int e = 0; e < 4; ++e) t[e] = img[Tr<T>::row(ln.q, e) * Tr<T>::LD + ln.r];
}
// the log-determinants the paired passes collect: `la` by role of the lane's row (first pass), `lb` the pivot's second diagonal tile
// (all rows alike), `lp` the passes that only invert a chol(Q) (chunk prologue, seed step of a spike; tri_inv_mat's layout)
template <typename T, int NT> struct PairLogs {
    LogAcc<T> la, lb, lp;
    MF_DEV void init() { la.init(); lb.init(); lp.init(); }
    MF_DEV T pivots() const {                    // log prod of the pivots = 2 log|L|
        if constexpr (NT == 1) return of_tile<T, 2>(la.value(), 0);
        else return of_tile<T, 4>(la.value(), 0) + lb.value();
    }
    MF_DEV T chol_q() const {                    // log prod diag chol(Q) over the blocks
        if constexpr (NT == 1) return of_tile<T, 2>(la.value(), 1);
        else {
            const T v = la.value();
            return of_tile<T, 4>(v, 1) + of_tile<T, 4>(v, 2) + tri_logdet<T, 2>(lp);
        }
    }
};
// C (S_LOWER; c10t = C10^T) -> Ci = C^-1, nothing to factor beside it
template <typename T, int NT>
MF_DEV void pair_tri_only(const Mat<T, NT>& C, const typename Tr<T>::v4& c10t, Mat<T, NT>& Ci, T* lds, const Lane& ln, PairLogs<T, NT>& lg,
                          bool& bad) {
    if constexpr (NT == 1) {
        typename Tr<T>::v4 unused;
        chol_tri_pair<T>(identity_tile<T>(ln), C.t[0][0], unused, Ci.t[0][0], lds, ln, lg.la, bad);
    } else {
        tri_inv_mat<T, NT>(C, c10t, Ci, lds, ln, lg.lp, bad);
    }
}
// Phi (symmetric, tiles ti <= tj; consumed) -> LiT = chol(Phi)^-T (S_UPPER);  C (S_LOWER; c10t = C10^T) -> Ci = C^-1 (S_LOWER)
template <typename T, int NT>
MF_DEV void pair_both(Mat<T, NT>& Phi, const Mat<T, NT>& C, const typename Tr<T>::v4& c10t, Mat<T, NT>& LiT, Mat<T, NT>& Ci, T* lds,
                      const Lane& ln, PairLogs<T, NT>& lg, bool& bad) {
    using v4 = typename Tr<T>::v4;
    if constexpr (NT == 1) {
        chol_tri_pair<T>(Phi.t[0][0], C.t[0][0], LiT.t[0][0], Ci.t[0][0], lds, ln, lg.la, bad);
    } else {
        constexpr int TS = 16 * Tr<T>::LD;
        const v4 in[4] = {Phi.t[0][0], C.t[0][0], C.t[1][1], identity_tile<T>(ln)};
        chol_tri_pass<T, 4>(in, ln.q == 0, lds, ln, lg.la, bad);
        v4 li00, cit11;
        image_to_tile_t<T>(LiT.t[0][0], lds, ln);
        image_to_tile<T>(li00, lds, ln);
        image_to_tile<T>(Ci.t[0][0], lds + TS, ln);
        image_to_tile<T>(Ci.t[1][1], lds + 2 * TS, ln);
        image_to_tile_t<T>(cit11, lds + 2 * TS, ln);
        lds_fence();
        // the pivot's second block column (chol_inv_mat)
        v4 lt01 = {0, 0, 0, 0}, acc = {0, 0, 0, 0}, z = {0, 0, 0, 0}, h = {0, 0, 0, 0};
        MF_UNROLL for (int e = 0; e < 4; ++e) lt01 = Tr<T>::mfma(LiT.t[0][0][e], Phi.t[0][1][e], lt01);      // (L10)^T = Li00 Phi01
        MF_UNROLL for (int e = 0; e < 4; ++e) acc = Tr<T>::mfma(lt01[e], lt01[e], acc);                      // L10 L10^T
        Phi.t[1][1] -= acc;
        {
            const v4 p11[1] = {Phi.t[1][1]};
            v4 out[1];
            chol_inv_tiles<T, 1, true>(p11, out, lds, ln, lg.lb, bad);
            LiT.t[1][1] = out[0];
        }
        MF_UNROLL for (int e = 0; e < 4; ++e) z = Tr<T>::mfma(lt01[e], li00[e], z);                          // Z = L10 Li00
        MF_UNROLL for (int e = 0; e < 4; ++e) h = Tr<T>::mfma(z[e], LiT.t[1][1][e], h);                      // Z^T Li11^T
        LiT.t[0][1] = -h;
        LiT.t[1][0] = v4{0, 0, 0, 0};
        // the off-diagonal tile of chol(Q)^-1 (tri_inv_mat)
        v4 g = {0, 0, 0, 0}, h2 = {0, 0, 0, 0};
        MF_UNROLL for (int e = 0; e < 4; ++e) g = Tr<T>::mfma(c10t[e], Ci.t[0][0][e], g);                    // C10 Ci00
        MF_UNROLL for (int e = 0; e < 4; ++e) h2 = Tr<T>::mfma(cit11[e], g[e], h2);                          // Ci11 (C10 Ci00)
        Ci.t[1][0] = -h2;
        Ci.t[0][1] = v4{0, 0, 0, 0};
    }
}
template <typename T, int NT> MF_DEV void identity_mat(Mat<T, NT>& m, const Lane& ln) {
    MF_UNROLL for (int i = 0; i < NT; ++i)
        MF_UNROLL for (int j = 0; j < NT; ++j) m.t[i][j] = (i == j) ? identity_tile<T>(ln) : typename Tr<T>::v4{0, 0, 0, 0};
}

// wave_kf_chunk_kernel with the paired passes: same arithmetic per block, same reduced system.
template <typename T, int NT, int M, int WPE, bool EX>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) wave_kf_pair_kernel(WvArgs<T> a, RedSys<T> out) {
    using v4 = typename Tr<T>::v4;
    __shared__ __attribute__((aligned(16))) T lds[(NT == 1 ? 2 : 4) * 16 * Tr<T>::LD];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long id = blockIdx.x, s = id / a.P, c = id % a.P;
    int d = EX ? 16 * NT : a.d;
    const int m = a.m;
    const long nt = a.Tn - 1, tau0 = c * a.L;
    long len = nt - tau0;
    if (len > a.L) len = a.L;
    if (len < 0) len = 0;
    const bool spike = c > 0;
    const long dd = long(d) * d;

    WaveElim<T, NT> E;
    E.init();
    PairLogs<T, NT> lg;
    lg.init();
    T acc_ww = T(0), acc_yry = T(0);
    T Ri[M][M];
    if (!a.rinv_per_step) load_rinv<T, M>(Ri, a.Rinv, m);

    Mat<T, NT> Am, Dn, Ci;
    CV<T, NT> rn, mv_cv;
    RV<T, NT> mv_rv, rn0_rv;
    ObsRows<T, NT, M> ob;

    auto own_terms = [&]() {                    // from Ci = chol(Q)^-1 of the block: Dn = Q^-1 (both off-diagonal tiles), rn = Q^-1 mvec
        tn<T, NT, S_LOWER, S_LOWER, S_UPPER, OP_SET>(Dn, Ci, Ci);
        if constexpr (NT == 2) transpose_tile<T>(Dn.t[1][0], Dn.t[0][1], lds, ln);
        tn_mv<T, NT, S_FULL>(rn, Dn, mv_rv);
        acc_ww += dot_cv<T, NT>(rn, mv_cv);
        cv_to_rv<T, NT>(rn0_rv, rn, ln);
    };
    auto obs_terms = [&](long blk) {
        if (a.rinv_per_step) load_rinv<T, M>(Ri, a.Rinv + (s * a.Tn + blk) * m * m, m);
        acc_yry += obs_apply<T, NT, M>(ob, Ri, Dn, rn);
    };

    Mat<T, NT> Cn;
    v4 c10tn = {0, 0, 0, 0};
    if (c == 0) {   // block 0: the prior
        load_mat<T, NT, S_LOWER, EX>(Cn, a.cholP0 + s * dd, d, true, true, ln);
        if constexpr (NT == 2) load_tile_t<T, EX>(c10tn, a.cholP0 + s * dd, d, 1, 0, ln);
        load_rv<T, NT>(mv_rv, a.mu0 + s * d, d, ln);
        load_cv<T, NT>(mv_cv, a.mu0 + s * d, d, ln);
        ob.load(a.H + (s * a.Tn) * m * d, a.y + (s * a.Tn) * m, d, m, ln);
        pair_tri_only<T, NT>(Cn, c10tn, Ci, lds, ln, lg, E.bad);
        own_terms();
        obs_terms(0);
        MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = i; j < NT; ++j) E.Phi.t[i][j] = Dn.t[i][j];
        E.t = rn;
    }
    if (len > 0) {
        load_mat<T, NT, S_LOWER, EX>(Cn, a.cholQ + (s * nt + tau0) * dd, d, true, true, ln);
        if constexpr (NT == 2) load_tile_t<T, EX>(c10tn, a.cholQ + (s * nt + tau0) * dd, d, 1, 0, ln);
        pair_tri_only<T, NT>(Cn, c10tn, Ci, lds, ln, lg, E.bad);
    }
    for (long j = 0; j < len; ++j) {
        const long tau = tau0 + j, blk = tau + 1;
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        if constexpr (!EX) asm volatile("" : "+s"(d));
        const bool more = j + 1 < len;
        {
            const T* cq = a.cholQ + (s * nt + (more ? tau + 1 : tau)) * dd;
            load_mat<T, NT, S_LOWER, EX>(Cn, cq, d, true, true, ln);
            if constexpr (NT == 2) load_tile_t<T, EX>(c10tn, cq, d, 1, 0, ln);
        }
        load_mat<T, NT, S_FULL, EX>(Am, a.A + (s * nt + tau) * dd, d, false, false, ln);
        load_rv<T, NT>(mv_rv, a.b + (s * nt + tau) * d, d, ln);
        load_cv<T, NT>(mv_cv, a.b + (s * nt + tau) * d, d, ln);
        ob.load(a.H + (s * a.Tn + blk) * m * d, a.y + (s * a.Tn + blk) * m, d, m, ln);
        own_terms();
        phase();
        Mat<T, NT> S;
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_NEG>(S, Dn, Am);                   // S = -Q^-1 A
        CV<T, NT> btw;
        tn_mv<T, NT, S_FULL>(btw, Am, rn0_rv);                                  // A^T Q^-1 mvec
        obs_terms(blk);
        phase();
        if (!more) {                                                            // the last step has no successor: an identity
            identity_mat<T, NT>(Cn, ln);
            c10tn = v4{0, 0, 0, 0};
        }
        if (j == 0 && spike) {
            tn<T, NT, S_FULL, S_FULL, S_UPPER, OP_NEG>(E.GU, Am, S);            // GU = A^T Q^-1 A
            E.X = S;
            MF_UNROLL for (int k = 0; k < NT; ++k) E.gU.v[k] = -btw.v[k];
            MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int k = i; k < NT; ++k) E.Phi.t[i][k] = Dn.t[i][k];
            E.t = rn;
            pair_tri_only<T, NT>(Cn, c10tn, Ci, lds, ln, lg, E.bad);
        } else {
            tn<T, NT, S_FULL, S_FULL, S_UPPER, OP_SUB>(E.Phi, Am, S);           // D_{k-1} += A^T Q^-1 A: complete
            MF_UNROLL for (int k = 0; k < NT; ++k) E.t.v[k] -= btw.v[k];
            phase();
            WaveFact<T, NT> f;
            pair_both<T, NT>(E.Phi, Cn, c10tn, f.LiT, Ci, lds, ln, lg, E.bad);
            phase();
            if (spike) E.template after_factor<true>(f, ln); else E.template after_factor<false>(f, ln);
            phase();
            Mat<T, NT> ST, WT;
            transpose<T, NT, S_FULL>(ST, S, lds, ln);
            tn<T, NT, S_UPPER, S_FULL, S_FULL, OP_SET>(WT, f.LiT, ST);          // W^T = Li S^T
            phase();
            if (spike) E.template advance<true>(f, WT, Dn, rn); else E.template advance<false>(f, WT, Dn, rn);
        }
    }
    const T ww = sum16<T>(acc_ww), quad = sum16<T>(E.quad);
    const T scalar = T(-0.5) * (acc_yry + ww) + T(0.5) * quad - lg.chol_q() - T(0.5) * lg.pivots();
    store_chunk_wave<T, NT>(out, id, d, E, scalar, lds, ln);
    if (__any(E.bad) && threadIdx.x == 0 && a.info) raise_info(a.info);
}

// ---- StateSpaceModel._build_precision (+ H^T R^-1 H, + information vector) for 16 <= d <= 32 -------------------------------------
// (state_space_model.py:431-483, kalman_filter.py:86-101,149-156).  One wavefront per (series, block k): block k's diagonal block
// needs Q_k^-1 and - when a transition leaves it - A_{k+1}^T Q_{k+1}^-1 A_{k+1}; its sub-diagonal block is -Q_{k+1}^-1 A_{k+1}.
// The two Cholesky factors (four diagonal tiles at d > 16) are inverted SIDE BY SIDE in the rows of the wavefront (one pass).
// Every Q^-1 is formed twice over the launch (by the blocks on both sides of its transition): fully parallel, no second pass.
// The tile engine's version is a 256-thread workgroup per block: 10.9 ms at B = 512, T = 1000, d = 16 (profiles/r05_bigops_d16.txt).
template <typename T, int NT, int M, bool EX>
__global__ void __launch_bounds__(64) wave_ssm_precision_kernel(WvArgs<T> a, T* __restrict__ diag, T* __restrict__ sub,
                                                               T* __restrict__ eta) {
    using v4 = typename Tr<T>::v4;
    constexpr int NTL = 2 * NT, TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NTL * TS];
    const Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long id = blockIdx.x, s = id / a.Tn, k = id % a.Tn;
    const int d = EX ? 16 * NT : a.d, m = a.m;
    const long nt = a.Tn - 1, dd = long(d) * d;
    const bool has_next = k + 1 < a.Tn;
    LogAcc<T> la;
    la.init();
    bool bad = false;
    Mat<T, NT> C0, C1, Am;
    v4 c10t[2] = {v4{0, 0, 0, 0}, v4{0, 0, 0, 0}};
    const T* c0p = k == 0 ? a.cholP0 + s * dd : a.cholQ + (s * nt + k - 1) * dd;
    load_mat<T, NT, S_LOWER, EX>(C0, c0p, d, true, true, ln);
    if constexpr (NT == 2) load_tile_t<T, EX>(c10t[0], c0p, d, 1, 0, ln);
    MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j) C1.t[i][j] = (i == j) ? identity_tile<T>(ln) : v4{0, 0, 0, 0};
    Am.zero();
    if (has_next) {
        load_mat<T, NT, S_LOWER, EX>(C1, a.cholQ + (s * nt + k) * dd, d, true, true, ln);
        if constexpr (NT == 2) load_tile_t<T, EX>(c10t[1], a.cholQ + (s * nt + k) * dd, d, 1, 0, ln);
        load_mat<T, NT, S_FULL, EX>(Am, a.A + (s * nt + k) * dd, d, false, false, ln);
    }
    // both factors' inverses: Ci[0], Ci[1] (S_LOWER)
    Mat<T, NT> Ci[2];
    {
        v4 in[NTL], out[NTL];
        MF_UNROLL for (int f = 0; f < 2; ++f) MF_UNROLL for (int i = 0; i < NT; ++i) in[f * NT + i] = (f == 0 ? C0 : C1).t[i][i];
        tri_inv_tiles<T, NTL, false>(in, out, lds, ln, la, bad);
        MF_UNROLL for (int f = 0; f < 2; ++f) {
            Ci[f].zero();
            MF_UNROLL for (int i = 0; i < NT; ++i) Ci[f].t[i][i] = out[f * NT + i];
        }
        if constexpr (NT == 2) {
            v4 cit11[2];
            MF_UNROLL for (int f = 0; f < 2; ++f) image_to_tile_t<T>(cit11[f], lds + (f * NT + 1) * TS, ln);
            lds_fence();
            MF_UNROLL for (int f = 0; f < 2; ++f) {
                v4 g = {0, 0, 0, 0}, h = {0, 0, 0, 0};
                MF_UNROLL for (int e = 0; e < 4; ++e) g = Tr<T>::mfma(c10t[f][e], Ci[f].t[0][0][e], g);
                MF_UNROLL for (int e = 0; e < 4; ++e) h = Tr<T>::mfma(cit11[f][e], g[e], h);
                Ci[f].t[1][0] = -h;
            }
        }
    }
    Mat<T, NT> Dn, Q1;
    CV<T, NT> rn;
    MF_UNROLL for (int j = 0; j < NT; ++j) rn.v[j] = T(0);
    tn<T, NT, S_LOWER, S_LOWER, S_UPPER, OP_SET>(Dn, Ci[0], Ci[0]);                 // Q_k^-1
    if constexpr (NT == 2) transpose_tile<T>(Dn.t[1][0], Dn.t[0][1], lds, ln);
    if (eta) {
        RV<T, NT> mv;
        load_rv<T, NT>(mv, k == 0 ? a.mu0 + s * d : a.b + (s * nt + k - 1) * d, d, ln);
        tn_mv<T, NT, S_FULL>(rn, Dn, mv);                                           // Q_k^-1 m_k
    }
    if (a.H) {
        ObsRows<T, NT, M> ob;
        T Ri[M][M];
        load_rinv<T, M>(Ri, a.rinv_per_step ? a.Rinv + (s * a.Tn + k) * m * m : a.Rinv, m);
        ob.load(a.H + (s * a.Tn + k) * m * d, a.y ? a.y + (s * a.Tn + k) * m : a.H, d, m, ln);
        if (!a.y) { MF_UNROLL for (int o = 0; o < M; ++o) ob.y[o] = T(0); }         // no observation term in eta
        (void)obs_apply<T, NT, M>(ob, Ri, Dn, rn);
    }
    if (has_next) {
        tn<T, NT, S_LOWER, S_LOWER, S_UPPER, OP_SET>(Q1, Ci[1], Ci[1]);             // Q_{k+1}^-1
        if constexpr (NT == 2) transpose_tile<T>(Q1.t[1][0], Q1.t[0][1], lds, ln);
        Mat<T, NT> S;
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_NEG>(S, Q1, Am);                       // S_k = -Q_{k+1}^-1 A_{k+1}
        tn<T, NT, S_FULL, S_FULL, S_UPPER, OP_SUB>(Dn, Am, S);                      // + A^T Q^-1 A
        if (eta) {
            RV<T, NT> mv1, r1_rv;
            CV<T, NT> r1, btw;
            load_rv<T, NT>(mv1, a.b + (s * nt + k) * d, d, ln);
            tn_mv<T, NT, S_FULL>(r1, Q1, mv1);
            cv_to_rv<T, NT>(r1_rv, r1, ln);
            tn_mv<T, NT, S_FULL>(btw, Am, r1_rv);
            MF_UNROLL for (int j = 0; j < NT; ++j) rn.v[j] -= btw.v[j];            // - A^T Q^-1 b
        }
        store_mat<T, NT, false, EX>(sub + (s * nt + k) * dd, S, d, lds, ln);
    }
    store_mat<T, NT, true, EX>(diag + id * dd, Dn, d, lds, ln);
    if (eta) store_cv<T, NT>(eta + id * d, rn, d, ln);
    (void)bad;
}

// ---- StateSpaceModel.kl_divergence for 16 <= d <= 32: the per-block terms of KL(chain 1 || chain 2) on register tiles ---------------
// (state_space_model.py:528-593).  The reference assembles P2 = chain 2's precision, multiplies it block by block with chain 1's
// marginal / subsequent covariances and sums (:569-573), forms |L2^T (mu2 - mu1)|^2 and two log-determinants.  Here one wavefront per
// (series, block k) forms block row k of P2 in registers exactly as wave_ssm_precision_kernel does - D_k = Q_k^-1 + A^T Q_{k+1}^-1 A,
// S_k = -Q_{k+1}^-1 A - and reduces it on the spot against what chain 1 contributes to that block:
//     term_k = tr(D_k Sigma_kk) + 2 tr(S_k^T C_k) + delta_k^T D_k delta_k + 2 delta_{k+1}^T S_k delta_k
//              + 2 log|chol2_k| - 2 log|chol1_k| - d                               (C_k = Cov(x_{k+1}, x_k), delta = mu2 - mu1)
// so that KL = 1/2 sum_k term_k.  P2 never exists in memory (the composition it replaces wrote and re-read 2 x [B, T, d, d] and ran
// six element-wise / reduction launches of torch over them: 6.8 ms at B = 512, T = 1000, d = 16).
// a: chain 2 (cholP0, A, cholQ); chol1_0 / chol1_q: chain 1's factors (only their diagonals are read).
template <typename T, int NT, bool EX>
__global__ void __launch_bounds__(64) wave_ssm_kl_terms_kernel(WvArgs<T> a, const T* __restrict__ chol1_0, const T* __restrict__ chol1_q,
                                                              const T* __restrict__ cov, const T* __restrict__ cross,
                                                              const T* __restrict__ mdiff, T* __restrict__ terms) {
    using v4 = typename Tr<T>::v4;
    constexpr int NTL = 2 * NT, TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NTL * TS];
    const Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long id = blockIdx.x, s = id / a.Tn, k = id % a.Tn;
    const int d = EX ? 16 * NT : a.d;
    const long nt = a.Tn - 1, dd = long(d) * d;
    const bool has_next = k + 1 < a.Tn;
    LogAcc<T> la;
    la.init();
    bool bad = false;
    Mat<T, NT> C0, C1, Am;
    v4 c10t[2] = {v4{0, 0, 0, 0}, v4{0, 0, 0, 0}};
    const T* c0p = k == 0 ? a.cholP0 + s * dd : a.cholQ + (s * nt + k - 1) * dd;
    load_mat<T, NT, S_LOWER, EX>(C0, c0p, d, true, true, ln);
    if constexpr (NT == 2) load_tile_t<T, EX>(c10t[0], c0p, d, 1, 0, ln);
    MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j) C1.t[i][j] = (i == j) ? identity_tile<T>(ln) : v4{0, 0, 0, 0};
    Am.zero();
    if (has_next) {
        load_mat<T, NT, S_LOWER, EX>(C1, a.cholQ + (s * nt + k) * dd, d, true, true, ln);
        if constexpr (NT == 2) load_tile_t<T, EX>(c10t[1], a.cholQ + (s * nt + k) * dd, d, 1, 0, ln);
        load_mat<T, NT, S_FULL, EX>(Am, a.A + (s * nt + k) * dd, d, false, false, ln);
    }
    Mat<T, NT> Ci[2];
    T logdet2;
    {
        v4 in[NTL], out[NTL];
        MF_UNROLL for (int f = 0; f < 2; ++f) MF_UNROLL for (int i = 0; i < NT; ++i) in[f * NT + i] = (f == 0 ? C0 : C1).t[i][i];
        tri_inv_tiles<T, NTL, false>(in, out, lds, ln, la, bad);
        // la (per lane): the diagonal of the lane's own tile; block k's factor is the tiles [0, NT)
        const T lv = la.value();
        logdet2 = of_tile<T, NTL>(lv, 0);
        if constexpr (NT == 2) logdet2 += of_tile<T, NTL>(lv, 1);
        MF_UNROLL for (int f = 0; f < 2; ++f) {
            Ci[f].zero();
            MF_UNROLL for (int i = 0; i < NT; ++i) Ci[f].t[i][i] = out[f * NT + i];
        }
        if constexpr (NT == 2) {
            v4 cit11[2];
            MF_UNROLL for (int f = 0; f < 2; ++f) image_to_tile_t<T>(cit11[f], lds + (f * NT + 1) * TS, ln);
            lds_fence();
            MF_UNROLL for (int f = 0; f < 2; ++f) {
                v4 g = {0, 0, 0, 0}, h = {0, 0, 0, 0};
                MF_UNROLL for (int e = 0; e < 4; ++e) g = Tr<T>::mfma(c10t[f][e], Ci[f].t[0][0][e], g);
                MF_UNROLL for (int e = 0; e < 4; ++e) h = Tr<T>::mfma(cit11[f][e], g[e], h);
                Ci[f].t[1][0] = -h;
            }
        }
    }
    Mat<T, NT> Dn, S;
    tn<T, NT, S_LOWER, S_LOWER, S_UPPER, OP_SET>(Dn, Ci[0], Ci[0]);                 // Q_k^-1 (tiles ti <= tj)
    S.zero();
    if (has_next) {
        Mat<T, NT> Q1;
        tn<T, NT, S_LOWER, S_LOWER, S_UPPER, OP_SET>(Q1, Ci[1], Ci[1]);             // Q_{k+1}^-1
        if constexpr (NT == 2) transpose_tile<T>(Q1.t[1][0], Q1.t[0][1], lds, ln);
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_NEG>(S, Q1, Am);                       // S_k = -Q_{k+1}^-1 A_{k+1}
        tn<T, NT, S_FULL, S_FULL, S_UPPER, OP_SUB>(Dn, Am, S);                      // + A^T Q^-1 A
    }
    if constexpr (NT == 2) transpose_tile<T>(Dn.t[1][0], Dn.t[0][1], lds, ln);     // the full symmetric block
    // trace terms: element-wise against chain 1's covariance blocks, in the layout the tiles are in
    T acc = T(0);
    {
        Mat<T, NT> Sg;
        load_mat<T, NT, S_FULL, EX>(Sg, cov + (s * a.Tn + k) * dd, d, false, false, ln);
        MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j)
            MF_UNROLL for (int e = 0; e < 4; ++e) {
                // (the padded diagonal of D is one, the padding of Sigma zero: nothing to mask)
                acc = __builtin_fma(Dn.t[i][j][e], Sg.t[i][j][e], acc);
            }
        if (has_next) {
            load_mat<T, NT, S_FULL, EX>(Sg, cross + (s * nt + k) * dd, d, false, false, ln);
            MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j)
                MF_UNROLL for (int e = 0; e < 4; ++e) acc = __builtin_fma(T(2) * S.t[i][j][e], Sg.t[i][j][e], acc);
        }
    }
    T total = xor_rows<T>(sum16<T>(acc));
    // Mahalanobis terms
    {
        RV<T, NT> dk_rv;
        CV<T, NT> dk_cv, y;
        load_rv<T, NT>(dk_rv, mdiff + (s * a.Tn + k) * d, d, ln);
        load_cv<T, NT>(dk_cv, mdiff + (s * a.Tn + k) * d, d, ln);
        tn_mv<T, NT, S_FULL>(y, Dn, dk_rv);                                         // D_k delta_k (D symmetric)
        T mh = dot_cv<T, NT>(y, dk_cv);
        if (has_next) {
            RV<T, NT> dn_rv;
            load_rv<T, NT>(dn_rv, mdiff + (s * a.Tn + k + 1) * d, d, ln);
            tn_mv<T, NT, S_FULL>(y, S, dn_rv);                                      // S_k^T delta_{k+1}
            mh = __builtin_fma(T(2), dot_cv<T, NT>(y, dk_cv), mh);
        }
        total += sum16<T>(mh);
    }
    // log-determinants: chain 2's from the inversion above, chain 1's from the diagonal of its factor
    {
        const T* c1p = k == 0 ? chol1_0 + s * dd : chol1_q + (s * nt + k - 1) * dd;
        T l1 = T(0);
        MF_UNROLL for (int i = 0; i < NT; ++i) {
            const int j = 16 * i + ln.r;
            const T v = c1p[j < d ? j * d + j : 0];
            l1 += (j < d) ? log(v < T(0) ? -v : v) : T(0);
        }
        total += T(2) * logdet2 - T(2) * sum16<T>(l1) - T(d);
    }
    if (threadIdx.x == 0) terms[id] = total;
    (void)bad;
}
// out[s] = scale * sum_k terms[s][k]: one wavefront per series, a fixed summation order (deterministic)
template <typename T> __global__ void __launch_bounds__(64) row_sums_kernel(long B, long n, const T* __restrict__ terms, T scale, T* __restrict__ out) {
    const long s = blockIdx.x;
    T acc = T(0);
    for (long k = threadIdx.x; k < n; k += 64) acc += terms[s * n + k];
    acc = xor_rows<T>(sum16<T>(acc));
    if (threadIdx.x == 0) out[s] = scale * acc;
}

}  // namespace wv
}  // namespace mf
