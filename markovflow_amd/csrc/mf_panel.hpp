// Panel kernels: ONE WORKGROUP of NT wavefronts per (series, time-chunk) for state dimensions 32 < d <= 64 (NT = 3, 4 tiles of 16 per
// side), the register-tile design of mf_wave.hpp carried to the sizes where one wavefront cannot hold a matrix.
//
// Wavefront w OWNS COLUMN PANEL w of every DP x DP matrix of the step (DP = 16 NT): NT tiles in the accumulator layout of the 16x16x4
// matrix-core instruction, lane (r, q), element e of tile ti  <->  M[16 ti + row(q, e)][16 w + r]  (4 NT registers per matrix in
// fp32).  As in mf_wave.hpp every product is arranged as  P^T Q :  the panel of the result needs the wavefront's own panel of Q as
// the B operand - registers, no movement - and ALL of P as the A operand, which is the one thing that goes through LDS: the owners
// write their panels of P into an "accumulator image" (row c = column of P, 4 consecutive words = the 4 K-values a lane feeds to
// the 4 instructions of a K-tile), one workgroup barrier, and every wavefront reads the tiles of P it needs with one 16-byte LDS read
// per tile (fp64: two).  One write + NT reads per 4 NT matrix instructions: a quarter of an LDS instruction per MFMA, against the 1.8
// of the LDS-tile engine (mf_big_impl.hpp), where both operands of every product were re-read from LDS.
//
// Three images (52 KB in fp32 at d = 64) + the observation images + the diagonal-tile slots = 71 KB and < 256 registers per lane:
// TWO WORKGROUPS PER CU in fp32, so that one chunk's GEMM phases fill the matrix pipe while the other walks its diagonal-tile chain
// (the LDS-tile engine: 119 KB, 512 registers, one workgroup per CU, matrix pipe 20 % busy).  fp64 runs the same code with one
// workgroup per CU (v_mfma_f64_16x16x4_f64, images of 34 KB).
//
// Same mathematics and RedSys output as wave_kf_chunk_kernel / big_kf_chunk_kernel (reference kalman_filter.py:184-255,
// state_space_model.py:431-483, block_tri_diag.py:423-436): the reduction levels behind level 0 do not know which kernel produced
// their input.  The 16 x 16 diagonal tiles are factored / inverted inside one wavefront by mf_wave.hpp's DPP routines; the
// factorisation of the pivot is the blocked right-looking Cholesky on the UPPER factor U = L^T (row j of U is spread over the panels
// of the wavefronts w >= j), followed by a back-substitution for U^-1 = L^-T that every wavefront runs on its own panel.
#pragma once
#include "mf_wave.hpp"

namespace mf {
namespace pn {

using wv::Lane;
using wv::Tr;
using wv::lds_fence;

template <typename T> struct V4A { typedef typename Tr<T>::v4 type __attribute__((aligned(16))); };

template <typename T, int NT> struct PG {
    static constexpr int DP = 16 * NT;
    static constexpr int LD = DP + (sizeof(T) == 4 ? 4 : 2);      // image row stride (rows 16-byte aligned)
    static constexpr int IMG = DP * LD;
    static constexpr int TLD = Tr<T>::LD, SLOT = 16 * TLD;        // a diagonal-tile slot: mf_wave.hpp's tile image
};
// where K-value k (0..15) of a 16-block sits in an image row, so that the four values of a lane's q are consecutive
template <typename T> MF_DEV int pos16(int r) { return sizeof(T) == 4 ? r : 4 * (r & 3) + (r >> 2); }

// "this wavefront's panel index is j", opaque to the optimiser: a chain of  if (w == j) x = t[j]  over all j would otherwise be
// recognised as  x = t[w]  - a dynamically indexed register array, which lives in scratch memory
MF_DEV bool is_wave(int w, int j) {
    int x = w;
    asm volatile("" : "+s"(x));
    return x == j;
}

template <typename T, int NT> struct Panel {
    typename Tr<T>::v4 t[NT];
    MF_DEV void zero() { MF_UNROLL for (int i = 0; i < NT; ++i) t[i] = typename Tr<T>::v4{0, 0, 0, 0}; }
};
template <typename T, int NT> struct RV { T v[NT][4]; };          // lane (r, q): v[ti][e] = vec[16 ti + row(q, e)]

// ---- accumulator images ---------------------------------------------------------------------------------------------------------
// The image "holds" a matrix P: tile(tk, ti) returns tile (tk, ti) of P in the accumulator layout, i.e. the A operand of P^T Q.
template <typename T, int LD> MF_DEV typename Tr<T>::v4 img_tile(const T* img, int tk, int ti, const Lane& ln) {
    return *reinterpret_cast<const typename V4A<T>::type*>(img + (16 * ti + ln.r) * LD + 16 * tk + 4 * ln.q);
}
// store tile (tk, ti) of the held matrix from its accumulator layout
template <typename T, int LD> MF_DEV void img_put(T* img, int tk, int ti, const typename Tr<T>::v4& t, const Lane& ln) {
    *reinterpret_cast<typename V4A<T>::type*>(img + (16 * ti + ln.r) * LD + 16 * tk + 4 * ln.q) = t;
}
// store tile (ti, tj) of M from its accumulator layout into an image that holds M^T
template <typename T, int LD> MF_DEV void img_put_t(T* img, int ti, int tj, const typename Tr<T>::v4& t, const Lane& ln) {
    const int p = 16 * tj + pos16<T>(ln.r);
    MF_UNROLL for (int e = 0; e < 4; ++e) img[(16 * ti + Tr<T>::row(ln.q, e)) * LD + p] = t[e];
}

template <typename T> MF_DEV typename Tr<T>::v4 mm(const typename Tr<T>::v4& a, const typename Tr<T>::v4& b, typename Tr<T>::v4 acc) {
    MF_UNROLL for (int e = 0; e < 4; ++e) acc = Tr<T>::mfma(a[e], b[e], acc);
    return acc;
}

enum { OP_SET = 0, OP_ADD = 1, OP_SUB = 2, OP_NEG = 3 };
// out[ti] (OP) sum_tk P(tk, ti)^T Q[tk]  for the ti with ti_on(ti) and the tk with tk_on(tk, ti); P from the image.  The four K-values of
// a tile are consecutive instructions on ONE accumulator; the output tiles are independent chains that the compiler interleaves.
template <typename T, int NT, int KT, int LD, int OP, typename TiOn, typename TkOn>
MF_DEV void tn_img(Panel<T, NT>& out, const T* img, const typename Tr<T>::v4 (&Q)[KT], const Lane& ln, TiOn ti_on, TkOn tk_on) {
    using v4 = typename Tr<T>::v4;
    v4 acc[NT];
    MF_UNROLL for (int ti = 0; ti < NT; ++ti) acc[ti] = (OP == OP_ADD) ? out.t[ti] : v4{0, 0, 0, 0};
    MF_UNROLL for (int tk = 0; tk < KT; ++tk) {
        MF_UNROLL for (int ti = 0; ti < NT; ++ti) {
            if (!ti_on(ti) || !tk_on(tk, ti)) continue;
            const v4 a = img_tile<T, LD>(img, tk, ti, ln);
            acc[ti] = mm<T>(a, Q[tk], acc[ti]);
        }
    }
    MF_UNROLL for (int ti = 0; ti < NT; ++ti) {
        if (!ti_on(ti)) continue;
        if (OP == OP_SUB) out.t[ti] -= acc[ti];
        else if (OP == OP_NEG) out.t[ti] = -acc[ti];
        else out.t[ti] = acc[ti];
    }
}

// ---- vectors: the wavefront's own 16 entries (one per lane r, the same in the four rows q) <-> the whole vector by row -------------
template <typename T> MF_DEV void vec_put(T* vec, int w, T x, const Lane& ln) {
    if (ln.q == 0) vec[16 * w + ln.r] = x;
}
template <typename T, int NT> MF_DEV void vec_rv(RV<T, NT>& v, const T* vec, const Lane& ln) {
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        MF_UNROLL for (int e = 0; e < 4; ++e) v.v[ti][e] = vec[16 * ti + Tr<T>::row(ln.q, e)];
}
// own entries of M^T v (M: the wavefront's panel, KT tiles high)
template <typename T, int KT, typename On> MF_DEV T mv_panel(const typename Tr<T>::v4 (&M)[KT], const T (&v)[KT][4], On on) {
    T acc = T(0);
    MF_UNROLL for (int ti = 0; ti < KT; ++ti) {
        if (!on(ti)) continue;
        MF_UNROLL for (int e = 0; e < 4; ++e) acc = __builtin_fma(M[ti][e], v[ti][e], acc);
    }
    return wv::xor_rows<T>(acc);
}

// ---- global memory <-> panels ------------------------------------------------------------------------------------------------------
// panel w of a d x d row-major matrix.  lower: the strict upper triangle reads as zero; idpad: ones on the padded diagonal
template <typename T, int NT, bool EX>
MF_DEV void load_panel(Panel<T, NT>& p, const T* __restrict__ g, int d, int w, bool lower, bool idpad, const Lane& ln) {
    MF_UNROLL for (int ti = 0; ti < NT; ++ti) {
        if (lower && ti < w) { p.t[ti] = typename Tr<T>::v4{0, 0, 0, 0}; continue; }
        MF_UNROLL for (int e = 0; e < 4; ++e) {
            const int i = 16 * ti + Tr<T>::row(ln.q, e), j = 16 * w + ln.r;
            const bool in = (EX || (i < d && j < d)) && (!lower || j <= i);
            const T v = g[in ? i * d + j : 0];
            p.t[ti][e] = in ? v : ((!EX && idpad && i == j && i >= d) ? T(1) : T(0));
        }
    }
}
// panel w of g^T (g: d x d row-major): 16 contiguous bytes per lane and tile in fp32
template <typename T, int NT, bool EX>
MF_DEV void load_panel_t(Panel<T, NT>& p, const T* __restrict__ g, int d, int w, const Lane& ln) {
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        MF_UNROLL for (int e = 0; e < 4; ++e) {
            const int i = 16 * ti + Tr<T>::row(ln.q, e), j = 16 * w + ln.r;          // element (i, j) of g^T = g[j][i]
            const bool in = EX || (i < d && j < d);
            const T v = g[in ? j * d + i : 0];
            p.t[ti][e] = in ? v : T(0);
        }
}
// rows of an mo x d matrix (observation matrix), panel w: MT tiles high
template <typename T, int MT>
MF_DEV void load_rows_panel(typename Tr<T>::v4 (&p)[MT], const T* __restrict__ g, int mo, int d, int w, const Lane& ln) {
    MF_UNROLL for (int to = 0; to < MT; ++to)
        MF_UNROLL for (int e = 0; e < 4; ++e) {
            const int i = 16 * to + Tr<T>::row(ln.q, e), j = 16 * w + ln.r;
            const bool in = i < mo && j < d;
            const T v = g[in ? i * d + j : 0];
            p[to][e] = in ? v : T(0);
        }
}
template <typename T, int KT> MF_DEV void load_rv_g(T (&v)[KT][4], const T* __restrict__ g, int n, const Lane& ln) {
    MF_UNROLL for (int ti = 0; ti < KT; ++ti)
        MF_UNROLL for (int e = 0; e < 4; ++e) {
            const int i = 16 * ti + Tr<T>::row(ln.q, e);
            const T x = g[i < n ? i : 0];
            v[ti][e] = i < n ? x : T(0);
        }
}
template <typename T> MF_DEV T load_cv_g(const T* __restrict__ g, int d, int w, const Lane& ln) {
    const int j = 16 * w + ln.r;
    const T x = g[j < d ? j : 0];
    return j < d ? x : T(0);
}
// the d x d corner of the matrix whose panel w is p.  SYM: only the tiles ti <= w are valid and the matrix is symmetric - each tile
// is also written at its mirrored place
template <typename T, int NT, bool SYM>
MF_DEV void store_panel(T* __restrict__ g, const Panel<T, NT>& p, int d, int w, const Lane& ln) {
    MF_UNROLL for (int ti = 0; ti < NT; ++ti) {
        if (SYM && ti > w) continue;
        MF_UNROLL for (int e = 0; e < 4; ++e) {
            const int i = 16 * ti + Tr<T>::row(ln.q, e), j = 16 * w + ln.r;
            if (i < d && j < d) {
                g[i * d + j] = p.t[ti][e];
                if (SYM && ti < w) g[j * d + i] = p.t[ti][e];
            }
        }
    }
}

// ---- the workgroup's LDS ---------------------------------------------------------------------------------------------------------------
template <typename T, int NT, int MT> struct Lds {
    using G = PG<T, NT>;
    static constexpr int MP = 16 * MT;
    static constexpr int LDH = MP + (sizeof(T) == 4 ? 4 : 2);
    static constexpr int ELEMS = 3 * G::IMG + G::DP * LDH + MP * LDH + NT * G::SLOT + 3 * G::DP + MP + 64;
    static constexpr int BYTES = ELEMS * (int)sizeof(T);
    T* base;
    MF_DEV T* I(int i) const { return base + i * G::IMG; }
    MF_DEV T* IH() const { return base + 3 * G::IMG; }                        // holds H (MP x DP)
    MF_DEV T* IR() const { return IH() + G::DP * LDH; }                       // holds R^-1 (MP x MP, symmetric)
    MF_DEV T* slot(int i) const { return IR() + MP * LDH + i * G::SLOT; }     // inverse of diagonal tile i, row-major (mf_wave.hpp image)
    MF_DEV T* vec(int i) const { return slot(NT) + i * G::DP; }               // 0: rn, 1: t, 2: z
    MF_DEV T* ys() const { return vec(3); }
    MF_DEV T* red() const { return ys() + MP; }
};
enum { V_RN = 0, V_T = 1, V_Z = 2 };

template <typename T, int NT, int MT> struct Ctx {
    Lds<T, NT, MT> sm;
    Lane ln;
    int w;               // wavefront = panel index (wave-uniform)
};

// ---- C (lower triangular panels, tiles ti >= w) -> Ci = C^-1 (lower).  Barriers: 1.  Uses image 0 (holds C^T) and the slots. -----------
template <typename T, int NT, int MT>
MF_DEV void tri_inv_panel(const Panel<T, NT>& C, Panel<T, NT>& Ci, const Ctx<T, NT, MT>& c, LogAcc<T>& la, bool& bad) {
    using v4 = typename Tr<T>::v4;
    using G = PG<T, NT>;
    const Lane& ln = c.ln;
    T* I0 = c.sm.I(0);
    MF_UNROLL for (int ti = 1; ti < NT; ++ti)
        if (ti > c.w) img_put_t<T, G::LD>(I0, ti, c.w, C.t[ti], ln);
    v4 in[1] = {C.t[0]}, o[1];
    MF_UNROLL for (int j = 1; j < NT; ++j) if (is_wave(c.w, j)) in[0] = C.t[j];
    wv::tri_inv_tiles<T, 1, false>(in, o, c.sm.slot(c.w), ln, la, bad);       // the slot keeps Ci_ww row-major
    __syncthreads();
    Ci.zero();
    MF_UNROLL for (int j = 0; j < NT; ++j) if (is_wave(c.w, j)) Ci.t[j] = o[0];
    // forward substitution down the wavefront's own column: Ci(i, w) = -Ci_ii sum_{k = w}^{i - 1} C(i, k) Ci(k, w)
    MF_UNROLL for (int i = 1; i < NT; ++i) {
        if (i <= c.w) continue;
        v4 acc = {0, 0, 0, 0};
        MF_UNROLL for (int k = 0; k < i; ++k) {
            if (k < c.w) continue;
            acc = mm<T>(img_tile<T, G::LD>(I0, k, i, ln), Ci.t[k], acc);          // tile (k, i) of C^T: the A operand of C(i, k) Q
        }
        v4 dt;
        wv::image_to_tile_t<T>(dt, c.sm.slot(i), ln);                             // Ci_ii^T: the A operand of Ci_ii Q
        Ci.t[i] = -mm<T>(dt, acc, v4{0, 0, 0, 0});
    }
}

// ---- Phi (symmetric, tiles ti <= w valid; consumed) -> LiT = chol(Phi)^-T (upper: tiles ti <= w).  Barriers: 2 NT - 1. ------------------
// Uses image 1 (holds U = L^T), image 2 (holds U^T) and the slots.
template <typename T, int NT, int MT>
MF_DEV void chol_inv_panel(Panel<T, NT>& Phi, Panel<T, NT>& LiT, const Ctx<T, NT, MT>& c, LogAcc<T>& la, bool& bad) {
    using v4 = typename Tr<T>::v4;
    using G = PG<T, NT>;
    const Lane& ln = c.ln;
    T *I1 = c.sm.I(1), *I2 = c.sm.I(2);
    v4 own = {0, 0, 0, 0};
    MF_UNROLL for (int j = 0; j < NT; ++j) {
        if (is_wave(c.w, j)) {
            const v4 in[1] = {Phi.t[j]};
            v4 o[1];
            wv::chol_inv_tiles<T, 1, true>(in, o, c.sm.slot(j), ln, la, bad);     // the slot keeps Li_jj = L_jj^-1 row-major
            own = o[0];                                                           // Li_jj^T = U_jj^-1 in the accumulator layout
        }
        __syncthreads();
        if (j + 1 < NT) {
            if (c.w > j) {
                v4 lt;
                wv::image_to_tile_t<T>(lt, c.sm.slot(j), ln);                     // Li_jj^T: the A operand of Li_jj Q
                const v4 u = mm<T>(lt, Phi.t[j], v4{0, 0, 0, 0});                 // U(j, w) = Li_jj Phi(j, w)
                Phi.t[j] = u;
                img_put<T, G::LD>(I1, j, c.w, u, ln);
                img_put_t<T, G::LD>(I2, j, c.w, u, ln);
            }
            __syncthreads();
            if (c.w > j) {
                MF_UNROLL for (int i = j + 1; i < NT; ++i) {
                    if (i > c.w) continue;
                    Phi.t[i] -= mm<T>(img_tile<T, G::LD>(I1, j, i, ln), Phi.t[j], v4{0, 0, 0, 0});      // -= U(j, i)^T U(j, w)
                }
            }
        }
    }
    // U^-1, own panel: Uinv(i, w) = -Uinv_ii sum_{k = i + 1}^{w} U(i, k) Uinv(k, w), i = w - 1 ... 0
    LiT.zero();
    MF_UNROLL for (int j = 0; j < NT; ++j) if (is_wave(c.w, j)) LiT.t[j] = own;
    MF_UNROLL for (int i = NT - 2; i >= 0; --i) {
        if (i >= c.w) continue;
        v4 acc = {0, 0, 0, 0};
        MF_UNROLL for (int k = i + 1; k < NT; ++k) {
            if (k > c.w) continue;
            acc = mm<T>(img_tile<T, G::LD>(I2, k, i, ln), LiT.t[k], acc);          // tile (k, i) of U^T: the A operand of U(i, k) Q
        }
        v4 li;
        wv::image_to_tile<T>(li, c.sm.slot(i), ln);                               // Li_ii: the A operand of Li_ii^T Q = Uinv_ii Q
        LiT.t[i] = -mm<T>(li, acc, v4{0, 0, 0, 0});
    }
}

// ---- the elimination state of one chunk ------------------------------------------------------------------------------------------------
template <typename T, int NT> struct PanelElim {
    Panel<T, NT> Phi;      // symmetric (tiles ti <= w): pivot of the current block
    Panel<T, NT> X;        // coupling current block <-> the chunk's left separator
    Panel<T, NT> GU;       // symmetric (tiles ti <= w): accumulated contribution to the separator's pivot
    T t, gU;               // own entries of the right-hand sides
    T quad;                // own entries of z, squared and summed
    LogAcc<T> laL;
    bool bad;
    MF_DEV void init() {
        Phi.zero(); X.zero(); GU.zero();
        t = T(0); gU = T(0); quad = T(0);
        laL.init();
        bad = false;
    }
};

// Eliminate the block whose complete pivot is in E.Phi and whose right-hand side is E.t; then advance to the next block, whose own
// pivot / right-hand-side parts are Dn (tiles ti <= w) / rn and whose coupling to the eliminated block is W with WT = W^T = Li S^T.
//   ST_FROM_S: S is the wavefront's panel of the coupling S (level 0: transposed through image 1);
//   else:      S already holds the panel of S^T (reduction levels: the coupling is read transposed from memory).
// Images: 0 <- LiT, 1 <- S^T then WT, 2 <- V.  Barriers: 2 NT - 1 + 3.  late(): called at the end, where few panels are
// live - the place for the next step's first global loads.
template <typename T, int NT, int MT, bool ST_FROM_S, typename Late>
MF_DEV void eliminate_advance(PanelElim<T, NT>& E, Panel<T, NT>& S, const Panel<T, NT>& Dn, T rn, bool spike,
                              const Ctx<T, NT, MT>& c, Late late) {
    using G = PG<T, NT>;
    const Lane& ln = c.ln;
    const int w = c.w;
    T *I0 = c.sm.I(0), *I1 = c.sm.I(1), *I2 = c.sm.I(2);
    vec_put<T>(c.sm.vec(V_T), w, E.t, ln);
    Panel<T, NT> LiT;
    chol_inv_panel<T, NT, MT>(E.Phi, LiT, c, E.laL, E.bad);
    T z;
    {
        RV<T, NT> t_rv;
        vec_rv<T, NT>(t_rv, c.sm.vec(V_T), ln);
        z = mv_panel<T, NT>(LiT.t, t_rv.v, [&](int ti) { return ti <= w; });            // z = Li t
    }
    E.quad = __builtin_fma(z, z, E.quad);
    vec_put<T>(c.sm.vec(V_Z), w, z, ln);
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        if (ti <= w) img_put<T, G::LD>(I0, ti, w, LiT.t[ti], ln);
    if (ST_FROM_S) {
        MF_UNROLL for (int ti = 0; ti < NT; ++ti) img_put_t<T, G::LD>(I1, ti, w, S.t[ti], ln);
    }
    __syncthreads();
    Panel<T, NT> V;
    if (spike) {
        tn_img<T, NT, NT, G::LD, OP_SET>(V, I0, E.X.t, ln, [](int) { return true; }, [](int tk, int ti) { return tk <= ti; });     // V = Li X
        RV<T, NT> z_rv;
        vec_rv<T, NT>(z_rv, c.sm.vec(V_Z), ln);
        E.gU -= mv_panel<T, NT>(V.t, z_rv.v, [](int) { return true; });                                                            // gU -= V^T z
        MF_UNROLL for (int ti = 0; ti < NT; ++ti) img_put<T, G::LD>(I2, ti, w, V.t[ti], ln);
    }
    Panel<T, NT> WT;
    {
        Panel<T, NT> ST;
        if (ST_FROM_S) {
            MF_UNROLL for (int tk = 0; tk < NT; ++tk) ST.t[tk] = img_tile<T, G::LD>(I1, tk, w, ln);
        } else {
            ST = S;
        }
        tn_img<T, NT, NT, G::LD, OP_SET>(WT, I0, ST.t, ln, [](int) { return true; }, [](int tk, int ti) { return tk <= ti; });     // W^T = Li S^T
    }
    __syncthreads();
    MF_UNROLL for (int ti = 0; ti < NT; ++ti) img_put<T, G::LD>(I1, ti, w, WT.t[ti], ln);
    if (spike)
        tn_img<T, NT, NT, G::LD, OP_SUB>(E.GU, I2, V.t, ln, [&](int ti) { return ti <= w; }, [](int, int) { return true; });         // GU -= V^T V
    __syncthreads();
    {
        RV<T, NT> z2;
        vec_rv<T, NT>(z2, c.sm.vec(V_Z), ln);
        E.t = rn - mv_panel<T, NT>(WT.t, z2.v, [](int) { return true; });                                                          // t = rn - W z
    }
    MF_UNROLL for (int ti = 0; ti < NT; ++ti) E.Phi.t[ti] = Dn.t[ti];
    tn_img<T, NT, NT, G::LD, OP_SUB>(E.Phi, I1, WT.t, ln, [&](int ti) { return ti <= w; }, [](int, int) { return true; });           // Phi = Dn - W W^T
    if (spike) tn_img<T, NT, NT, G::LD, OP_NEG>(E.X, I1, V.t, ln, [](int) { return true; }, [](int, int) { return true; });         // X = -W V
    late();
}

// The last block of a final reduction: factor, z, nothing to advance to.
template <typename T, int NT, int MT> MF_DEV void eliminate_last(PanelElim<T, NT>& E, const Ctx<T, NT, MT>& c) {
    const int w = c.w;
    vec_put<T>(c.sm.vec(V_T), w, E.t, c.ln);
    Panel<T, NT> LiT;
    chol_inv_panel<T, NT, MT>(E.Phi, LiT, c, E.laL, E.bad);
    RV<T, NT> t_rv;
    vec_rv<T, NT>(t_rv, c.sm.vec(V_T), c.ln);
    const T z = mv_panel<T, NT>(LiT.t, t_rv.v, [&](int ti) { return ti <= w; });
    E.quad = __builtin_fma(z, z, E.quad);
    __syncthreads();
}

// sum over the workgroup of a per-wavefront value (uniform within the wavefront); every thread gets the total.  Barriers: 2.
template <typename T, int NT> MF_DEV T wg_sum(T x, T* red, int w) {
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = x;
    __syncthreads();
    T s = T(0);
    MF_UNROLL for (int i = 0; i < NT; ++i) s += red[i];
    return s;
}

template <typename T, int NT, int MT>
MF_DEV void store_chunk_panel(const RedSys<T>& out, long idx, int d, const PanelElim<T, NT>& E, T scalar, const Ctx<T, NT, MT>& c) {
    const long dd = long(d) * d;
    store_panel<T, NT, true>(out.Dv + idx * dd, E.Phi, d, c.w, c.ln);
    store_panel<T, NT, true>(out.GU + idx * dd, E.GU, d, c.w, c.ln);
    store_panel<T, NT, false>(out.F + idx * dd, E.X, d, c.w, c.ln);
    const int j = 16 * c.w + c.ln.r;
    if (c.ln.q == 0 && j < d) {
        out.tv[idx * d + j] = E.t;
        out.gU[idx * d + j] = E.gU;
    }
    if (threadIdx.x == 0) out.sc[idx] = scalar;
}

template <typename T, int NT, int MT> constexpr int panel_wpe() { return sizeof(T) == 4 ? 2 : 1; }

// Level 0: workgroup (s, c) eliminates the transitions [c L, min((c+1) L, T-1)) of series s.
template <typename T, int NT, int MT, bool EX>
__global__ void __launch_bounds__(64 * NT) __attribute__((amdgpu_waves_per_eu(panel_wpe<T, NT, MT>(), panel_wpe<T, NT, MT>())))
panel_kf_chunk_kernel(wv::WvArgs<T> a, RedSys<T> out) {
    using v4 = typename Tr<T>::v4;
    using G = PG<T, NT>;
    using L = Lds<T, NT, MT>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Ctx<T, NT, MT> c{L{reinterpret_cast<T*>(smem_raw)}, Lane{(int)(threadIdx.x & 15), (int)((threadIdx.x >> 4) & 3)},
                     __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))};
    Lane& ln = c.ln;
    const int w = c.w;
    const long id = blockIdx.x, s = id / a.P, ch = id % a.P;
    int d = EX ? 16 * NT : a.d;
    const int m = a.m;
    const long nt = a.Tn - 1, tau0 = ch * a.L;
    long len = nt - tau0;
    if (len > a.L) len = a.L;
    if (len < 0) len = 0;
    const bool spike = ch > 0;
    const long dd = long(d) * d;
    T *I0 = c.sm.I(0), *I1 = c.sm.I(1), *I2 = c.sm.I(2), *IH = c.sm.IH(), *IR = c.sm.IR();

    PanelElim<T, NT> E;
    E.init();
    LogAcc<T> laC;
    laC.init();
    T acc_ww = T(0), acc_yry = T(0);

    // R^-1 into its image (symmetric: the image of the matrix is the image of its transpose)
    auto stage_rinv = [&](const T* __restrict__ R) __attribute__((always_inline)) {
        for (int e = threadIdx.x; e < L::MP * L::MP; e += 64 * NT) {
            const int k = e / L::MP, cc = e % L::MP;
            IR[cc * L::LDH + 16 * (k >> 4) + pos16<T>(k & 15)] = (k < m && cc < m) ? R[k * m + cc] : T(0);
        }
    };
    if (!a.rinv_per_step) stage_rinv(a.Rinv);

    Panel<T, NT> Dn, Am;
    typename Tr<T>::v4 Hp[MT];
    T rn = T(0);
    // The block's own terms from its Cholesky factor C: Dn = Q^-1 (all tiles), rn = Q^-1 mvec; image 2 <- Q^-1; the observation rows
    // (-> Hp) and the transition (Ag != NULL, -> Am) are loaded behind the inversion of C - their latency is covered by the product that
    // follows - and go into their images.  Ends with the barrier that publishes them.
    auto own_terms = [&](const Panel<T, NT>& C, const T* __restrict__ mvec, long blk, const T* __restrict__ Ag) __attribute__((always_inline)) {
        if (a.rinv_per_step) stage_rinv(a.Rinv + (s * a.Tn + blk) * m * m);
        if (threadIdx.x < L::MP) c.sm.ys()[threadIdx.x] = (int)threadIdx.x < m ? a.y[(s * a.Tn + blk) * m + threadIdx.x] : T(0);
        Panel<T, NT> Ci;
        tri_inv_panel<T, NT, MT>(C, Ci, c, laC, E.bad);
        if (Ag) load_panel<T, NT, EX>(Am, Ag, d, w, false, false, ln);
        load_rows_panel<T, MT>(Hp, a.H + (s * a.Tn + blk) * m * d, m, d, w, ln);
        MF_UNROLL for (int ti = 0; ti < NT; ++ti)
            if (ti >= w) img_put<T, G::LD>(I1, ti, w, Ci.t[ti], ln);
        __syncthreads();
        tn_img<T, NT, NT, G::LD, OP_SET>(Dn, I1, Ci.t, ln, [](int) { return true; },
                                         [&](int tk, int ti) { return tk >= ti && tk >= w; });                  // Q^-1 = Ci^T Ci
        {
            RV<T, NT> mv;
            load_rv_g<T, NT>(mv.v, mvec, d, ln);
            rn = mv_panel<T, NT>(Dn.t, mv.v, [](int) { return true; });                                         // Q^-1 mvec
            acc_ww = __builtin_fma(rn, load_cv_g<T>(mvec, d, w, ln), acc_ww);                                   // mvec^T Q^-1 mvec
        }
        vec_put<T>(c.sm.vec(V_RN), w, rn, ln);
        MF_UNROLL for (int ti = 0; ti < NT; ++ti) img_put<T, G::LD>(I2, ti, w, Dn.t[ti], ln);
        MF_UNROLL for (int to = 0; to < MT; ++to) img_put<T, L::LDH>(IH, to, w, Hp[to], ln);
        if (Ag) {
            MF_UNROLL for (int ti = 0; ti < NT; ++ti) img_put<T, G::LD>(I0, ti, w, Am.t[ti], ln);
        }
        __syncthreads();
    };
    // the observation terms on top: Dn += H^T R^-1 H, rn += H^T R^-1 y, y^T R^-1 y (kalman_filter.py:86-101)
    auto obs_terms = [&](long blk) __attribute__((always_inline)) {
        v4 Gp[MT];
        {
            Panel<T, MT> Gq;
            tn_img<T, MT, MT, L::LDH, OP_SET>(Gq, IR, Hp, ln, [](int) { return true; }, [](int, int) { return true; });           // G = R^-1 H
            MF_UNROLL for (int to = 0; to < MT; ++to) Gp[to] = Gq.t[to];
        }
        tn_img<T, NT, MT, L::LDH, OP_ADD>(Dn, IH, Gp, ln, [](int) { return true; }, [](int, int) { return true; });                // += H^T G
        {
            T yv[MT][4];
            load_rv_g<T, MT>(yv, a.y + (s * a.Tn + blk) * m, m, ln);
            rn += mv_panel<T, MT>(Gp, yv, [](int) { return true; });                                                              // += G^T y
        }
        {   // y^T R^-1 y: thread (o, part) takes the terms p = part, part + NPART, ... of row o (R^-1 symmetric: read down a column)
            constexpr int NPART = 64 * NT / L::MP;
            const int o = threadIdx.x % L::MP, part = threadIdx.x / L::MP;
            if (part < NPART) {
                T acc = T(0);
                for (int p = part; p < L::MP; p += NPART) acc = __builtin_fma(IR[p * L::LDH + 16 * (o >> 4) + pos16<T>(o & 15)], c.sm.ys()[p], acc);
                acc_yry = __builtin_fma(acc, c.sm.ys()[o], acc_yry);
            }
        }
    };

    Panel<T, NT> Cn;
    if (ch == 0) {   // block 0: the prior
        load_panel<T, NT, EX>(Cn, a.cholP0 + s * dd, d, w, true, true, ln);
        own_terms(Cn, a.mu0 + s * d, 0, nullptr);
        obs_terms(0);
        MF_UNROLL for (int ti = 0; ti < NT; ++ti) E.Phi.t[ti] = Dn.t[ti];
        E.t = rn;
        __syncthreads();
    }
    if (len > 0) load_panel<T, NT, EX>(Cn, a.cholQ + (s * nt + tau0) * dd, d, w, true, true, ln);
    for (long j = 0; j < len; ++j) {
        const long tau = tau0 + j, blk = tau + 1;
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        if constexpr (!EX) asm volatile("" : "+s"(d));
        own_terms(Cn, a.b + (s * nt + tau) * d, blk, a.A + (s * nt + tau) * dd);
        Panel<T, NT> S;
        tn_img<T, NT, NT, G::LD, OP_NEG>(S, I2, Am.t, ln, [](int) { return true; }, [](int, int) { return true; });               // S = -Q^-1 A
        T btw;
        {
            RV<T, NT> rn_rv;
            vec_rv<T, NT>(rn_rv, c.sm.vec(V_RN), ln);
            btw = mv_panel<T, NT>(Am.t, rn_rv.v, [](int) { return true; });                                                       // A^T Q^-1 mvec
        }
        obs_terms(blk);
        if (j == 0 && spike) {
            // the block on the left is the chunk's separator: its coupling seeds the spike
            tn_img<T, NT, NT, G::LD, OP_NEG>(E.GU, I0, S.t, ln, [&](int ti) { return ti <= w; }, [](int, int) { return true; });   // GU = A^T Q^-1 A
            E.X = S;
            E.gU = -btw;
            MF_UNROLL for (int ti = 0; ti < NT; ++ti) E.Phi.t[ti] = Dn.t[ti];
            E.t = rn;
            if (j + 1 < len) load_panel<T, NT, EX>(Cn, a.cholQ + (s * nt + tau + 1) * dd, d, w, true, true, ln);
            __syncthreads();
        } else {
            tn_img<T, NT, NT, G::LD, OP_SUB>(E.Phi, I0, S.t, ln, [&](int ti) { return ti <= w; }, [](int, int) { return true; });  // D_{k-1} += A^T Q^-1 A
            E.t -= btw;
            // the next transition's Cholesky factor opens the next step: its loads are issued behind this step's last barrier
            eliminate_advance<T, NT, MT, true>(E, S, Dn, rn, spike, c, [&]() __attribute__((always_inline)) {
                if (j + 1 < len) load_panel<T, NT, EX>(Cn, a.cholQ + (s * nt + tau + 1) * dd, d, w, true, true, ln);
            });
        }
    }
    // the chunk's scalar: every wavefront's share, summed
    T part = T(-0.5) * wv::sum16<T>(acc_ww) + T(0.5) * wv::sum16<T>(E.quad) - laC.value() - T(0.5) * E.laL.value();
    part += T(-0.5) * wv::sum16<T>(wv::xor_rows<T>(acc_yry));
    const T scalar = wg_sum<T, NT>(part, c.sm.red(), w);
    store_chunk_panel<T, NT, MT>(out, id, d, E, scalar, c);
    if (__any(E.bad) && (threadIdx.x & 63) == 0 && a.info) raise_info(a.info);
}

// Reduction level: RedSys(n) -> RedSys(P) (FINAL: P = 1, the last block is eliminated too and out_scalar written).  Same block
// sequence as big_red_kernel (mf_big_impl.hpp).
template <typename T, int NT, bool FINAL, bool EX>
__global__ void __launch_bounds__(64 * NT) __attribute__((amdgpu_waves_per_eu(panel_wpe<T, NT, 1>(), panel_wpe<T, NT, 1>())))
panel_red_kernel(RedSys<T> in, RedSys<T> out, long B, long P, int d_, T add_const, T* __restrict__ out_scalar, int* info) {
    using L = Lds<T, NT, 1>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Ctx<T, NT, 1> c{L{reinterpret_cast<T*>(smem_raw)}, Lane{(int)(threadIdx.x & 15), (int)((threadIdx.x >> 4) & 3)},
                    __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))};
    Lane& ln = c.ln;
    const int w = c.w;
    int d = EX ? 16 * NT : d_;
    const long id = blockIdx.x, s = id / P, ch = id % P;
    const long k0 = (ch * in.n) / P, k1 = ((ch + 1) * in.n) / P;
    const bool spike = !FINAL && k0 > 0;
    const long dd = long(d) * d;
    PanelElim<T, NT> E;
    E.init();
    T acc_sc = T(0);
    for (long k = k0; k < k1; ++k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        if constexpr (!EX) asm volatile("" : "+s"(d));
        const long idx = s * in.n + k;
        const bool has_next = in.GU && (k + 1 < in.n);
        if (in.sc) acc_sc += in.sc[idx];
        // the block's own pivot / right-hand-side parts (padded diagonal = 1 keeps the padded states harmless)
        Panel<T, NT> Dn;
        load_panel<T, NT, EX>(Dn, in.Dv + idx * dd, d, w, false, true, ln);
        T rn = load_cv_g<T>(in.tv + idx * d, d, w, ln);
        if (has_next) {
            Panel<T, NT> G2;
            load_panel<T, NT, EX>(G2, in.GU + (idx + 1) * dd, d, w, false, false, ln);
            MF_UNROLL for (int ti = 0; ti < NT; ++ti) Dn.t[ti] += G2.t[ti];
            rn += load_cv_g<T>(in.gU + (idx + 1) * d, d, w, ln);
        }
        const T* Fk = in.F + (s * in.f_stride + k + in.f_off) * dd;
        if (k > k0) {
            Panel<T, NT> FT;
            load_panel_t<T, NT, EX>(FT, Fk, d, w, ln);
            eliminate_advance<T, NT, 1, false>(E, FT, Dn, rn, spike, c, [] {});
        } else {
            if (k > 0 && !FINAL) load_panel<T, NT, EX>(E.X, Fk, d, w, false, false, ln);
            MF_UNROLL for (int ti = 0; ti < NT; ++ti) E.Phi.t[ti] = Dn.t[ti];
            E.t = rn;
        }
    }
    if (FINAL) eliminate_last<T, NT, 1>(E, c);
    T part = T(0.5) * wv::sum16<T>(E.quad) - T(0.5) * E.laL.value();
    const T tot = wg_sum<T, NT>(part, c.sm.red(), w);
    if (FINAL) {
        if (threadIdx.x == 0) out_scalar[s] = add_const + acc_sc + tot;
    } else {
        store_chunk_panel<T, NT, 1>(out, id, d, E, acc_sc + tot, c);
    }
    if (__any(E.bad) && (threadIdx.x & 63) == 0 && info) raise_info(info);
}

}  // namespace pn
}  // namespace mf
