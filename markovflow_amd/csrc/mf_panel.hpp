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
#include "mf_wave_ops.hpp"   // chol_fact_tile (a diagonal tile factored with the factor itself kept: the emit pass)

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

MF_DEV float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
MF_DEV double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Phase timing of a level-0 step (diagnostic builds only: -DMF_PANEL_STAMP; the first wavefront of workgroup 1 prints the table).
// s_memtime between the phases, accumulated over the chunk.
#ifdef MF_PANEL_STAMP
struct Stamp {
    unsigned long long* acc;          // 12 words of LDS, written by thread 0
    unsigned long long prev;
    MF_DEV void init(void* lds) {
        acc = static_cast<unsigned long long*>(lds);
        if (threadIdx.x == 0) for (int i = 0; i < 12; ++i) acc[i] = 0;
        prev = __builtin_readcyclecounter();
    }
    MF_DEV void at(int i) {
        const unsigned long long now = __builtin_readcyclecounter();
        if (threadIdx.x == 0) acc[i] += now - prev;
        prev = now;
    }
    MF_DEV void print(long steps) {
        if (blockIdx.x == 1 && threadIdx.x == 0) {
            printf("panel step phases (memtime ticks per step, %ld steps):", steps);
            for (int i = 0; i < 12; ++i) printf(" [%d] %.0f", i, (double)acc[i] / (double)(steps > 0 ? steps : 1));
            printf("\n");
        }
    }
};
#define MF_PSTAMP(st, i) (st).at(i);
#else
struct Stamp {
    MF_DEV void init(void*) {}
    MF_DEV void print(long) {}
};
#define MF_PSTAMP(st, i)
#endif

template <typename T> MF_DEV typename Tr<T>::v4 mm(const typename Tr<T>::v4& a, const typename Tr<T>::v4& b, typename Tr<T>::v4 acc) {
    MF_UNROLL for (int e = 0; e < 4; ++e) acc = Tr<T>::mfma(a[e], b[e], acc);
    return acc;
}

enum { OP_SET = 0, OP_ADD = 1, OP_SUB = 2, OP_NEG = 3 };
enum { P_FULL = 0, P_UPPER = 1, P_LOWER = 2 };      // tile structure of P: all tiles, tk <= ti only, tk >= ti only
// out[ti] (OP) sum_tk P(tk, ti)^T Q[tk], P from the image.  Straight-line code - NO wavefront-dependent skipping: a symmetric result is
// formed in full (a branch per tile costs the schedule more than the instructions it would save: every tile product becomes a block of
// its own - LDS read, wait, four dependent matrix instructions).  The NT output tiles are independent accumulator chains, issued
// round-robin so that no instruction waits for the one before it on its own chain.
template <typename T, int NT, int KT, int LD, int OP, int PS>
MF_DEV void tn_img(Panel<T, NT>& out, const T* img, const typename Tr<T>::v4 (&Q)[KT], const Lane& ln) {
    using v4 = typename Tr<T>::v4;
    auto on = [](int tk, int ti) { return PS == P_FULL || (PS == P_UPPER ? tk <= ti : tk >= ti); };
    v4 acc[NT], a[2][NT];
    MF_UNROLL for (int ti = 0; ti < NT; ++ti) acc[ti] = (OP == OP_ADD) ? out.t[ti] : v4{0, 0, 0, 0};
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        if (on(0, ti)) a[0][ti] = img_tile<T, LD>(img, 0, ti, ln);
    // the A tiles of K-tile tk + 1 are read while the matrix instructions of K-tile tk issue; the scheduling fence keeps the compiler
    // from hoisting ALL the reads of a product to its top (4 NT^2 registers of operands in flight)
    MF_UNROLL for (int tk = 0; tk < KT; ++tk) {
        if (tk + 1 < KT) {
            MF_UNROLL for (int ti = 0; ti < NT; ++ti)
                if (on(tk + 1, ti)) a[(tk + 1) & 1][ti] = img_tile<T, LD>(img, tk + 1, ti, ln);
        }
        MF_UNROLL for (int e = 0; e < 4; ++e)
            MF_UNROLL for (int ti = 0; ti < NT; ++ti)
                if (on(tk, ti)) acc[ti] = Tr<T>::mfma(a[tk & 1][ti][e], Q[tk][e], acc[ti]);
        __builtin_amdgcn_sched_barrier(0);
    }
    MF_UNROLL for (int ti = 0; ti < NT; ++ti) {
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
template <typename T, int KT> MF_DEV T mv_panel(const typename Tr<T>::v4 (&M)[KT], const T (&v)[KT][4]) {
    T acc = T(0);
    MF_UNROLL for (int ti = 0; ti < KT; ++ti) {
        MF_UNROLL for (int e = 0; e < 4; ++e) acc = fma_t(M[ti][e], v[ti][e], acc);
    }
    return wv::xor_rows<T>(acc);
}

// ---- global memory <-> panels ------------------------------------------------------------------------------------------------------
// The lane coordinates behind an optimisation barrier: the per-lane offsets of a group of loads are then computed where the loads are
// issued.  Without it the compiler hoists them out of the step loop (they are loop-invariant), keeps 2 x 16 registers of 64-bit offsets
// alive across the whole step, spills them - and every reload waits (vmcnt is in order) for the global load issued just before it:
// sixteen loads of a panel went out ONE AT A TIME, each paying the full memory latency (profiles/r06_panel_phases.txt).
MF_DEV Lane opaque_lane(const Lane& ln) {
    Lane l = ln;
    asm volatile("" : "+v"(l.r), "+v"(l.q));
    return l;
}
// panel w of a d x d row-major matrix.  lower: the strict upper triangle reads as zero; idpad: ones on the padded diagonal.
// EX (d = 16 NT): every address is in range - the loads are unconditional (one base, immediate offsets) and the triangle is masked
// on the VALUES.
template <typename T, int NT, bool EX>
MF_DEV void load_panel(Panel<T, NT>& p, const T* __restrict__ g, int d, int w, bool lower, bool idpad, const Lane& ln_) {
    const Lane ln = opaque_lane(ln_);
    if constexpr (EX) {
        constexpr int D = 16 * NT;
        const T* __restrict__ base = g + Tr<T>::row(ln.q, 0) * D + 16 * w + ln.r;
        MF_UNROLL for (int ti = 0; ti < NT; ++ti) {
            if (lower && ti < w) { p.t[ti] = typename Tr<T>::v4{0, 0, 0, 0}; continue; }
            MF_UNROLL for (int e = 0; e < 4; ++e) p.t[ti][e] = base[(16 * ti + Tr<T>::row(0, e)) * D];
        }
        if (lower) {
            MF_UNROLL for (int ti = 0; ti < NT; ++ti)
                MF_UNROLL for (int e = 0; e < 4; ++e)
                    if (16 * ti + Tr<T>::row(ln.q, e) < 16 * w + ln.r) p.t[ti][e] = T(0);
        }
        return;
    }
    MF_UNROLL for (int ti = 0; ti < NT; ++ti) {
        if (lower && ti < w) { p.t[ti] = typename Tr<T>::v4{0, 0, 0, 0}; continue; }
        MF_UNROLL for (int e = 0; e < 4; ++e) {
            const int i = 16 * ti + Tr<T>::row(ln.q, e), j = 16 * w + ln.r;
            const bool in = (i < d && j < d) && (!lower || j <= i);
            const T v = g[in ? i * d + j : 0];
            p.t[ti][e] = in ? v : ((idpad && i == j && i >= d) ? T(1) : T(0));
        }
    }
}
// panel w of g^T (g: d x d row-major): 16 contiguous bytes per lane and tile in fp32
template <typename T, int NT, bool EX>
MF_DEV void load_panel_t(Panel<T, NT>& p, const T* __restrict__ g, int d, int w, const Lane& ln_) {
    const Lane ln = opaque_lane(ln_);
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
MF_DEV void load_rows_panel(typename Tr<T>::v4 (&p)[MT], const T* __restrict__ g, int mo, int d, int w, const Lane& ln_) {
    const Lane ln = opaque_lane(ln_);
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

// ---- LDS-DMA (buffer_load_dword ... lds): global memory -> LDS without a register in between ---------------------------------------
// One instruction moves 256 consecutive bytes (lane l: 4 bytes at voffset + 4 l) to the 256 LDS bytes at M0 + imm + 4 l.  Used to
// fetch the NEXT step's inputs while this step's last products run: nothing is loop-carried in registers (a loaded value that
// crosses the back edge of the step loop is waited for at the back edge).  The instructions sit in asm statements, so the compiler's
// own s_waitcnt bookkeeping does not know them: memory operations complete in order, its waits are therefore never too short, and the
// consumers below wait with an explicit s_waitcnt vmcnt(0).  Lanes beyond the descriptor's range write ZEROS (padding for free).
typedef int pn_v4i __attribute__((ext_vector_type(4)));
MF_DEV pn_v4i dma_srd(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    pn_v4i srd;
    srd.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    srd.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));
    srd.z = __builtin_amdgcn_readfirstlane((int)bytes);
    srd.w = 0x00020000;
    return srd;
}
MF_DEV unsigned lds_addr(const void* p) { return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)p); }
// sixteen rows of ROWB <= 256 bytes: global row stride ROWB, LDS row stride ROWB + PAD (the immediate offset advances both sides by
// ROWB, M0 by PAD); the lanes beyond a row are switched off
template <int ROWB, int PAD> MF_DEV void dma_rows16(pn_v4i srd, unsigned lds, unsigned voff) {
    unsigned keep;
#define MF_PN_ROW(i) "buffer_load_dword %3, %1, 0 offen offset:%" #i " lds\n\ts_add_u32 m0, m0, %4\n\ts_nop 0\n\t"
    if (voff < (unsigned)ROWB)
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     MF_PN_ROW(5) MF_PN_ROW(6) MF_PN_ROW(7) MF_PN_ROW(8) MF_PN_ROW(9) MF_PN_ROW(10) MF_PN_ROW(11) MF_PN_ROW(12)
                     MF_PN_ROW(13) MF_PN_ROW(14) MF_PN_ROW(15) MF_PN_ROW(16) MF_PN_ROW(17) MF_PN_ROW(18) MF_PN_ROW(19) MF_PN_ROW(20)
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(srd), "s"(lds), "v"(voff), "n"(PAD),
                       "n"(0 * ROWB), "n"(1 * ROWB), "n"(2 * ROWB), "n"(3 * ROWB), "n"(4 * ROWB), "n"(5 * ROWB), "n"(6 * ROWB), "n"(7 * ROWB),
                       "n"(8 * ROWB), "n"(9 * ROWB), "n"(10 * ROWB), "n"(11 * ROWB), "n"(12 * ROWB), "n"(13 * ROWB), "n"(14 * ROWB), "n"(15 * ROWB)
                     : "memory", "scc");
#undef MF_PN_ROW
}
// `bytes` (<= 512, a multiple of 4) consecutive bytes, of which the first `valid` come from memory and the rest are zeros
MF_DEV void dma_vec(const void* g, unsigned valid, void* dst, unsigned bytes) {
    const pn_v4i srd = dma_srd(g, valid);
    const unsigned lds = lds_addr(dst), voff = 4u * (threadIdx.x & 63);
    unsigned keep;
    if (voff < bytes)
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %3, %1, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(srd), "s"(lds), "v"(voff) : "memory");
    if (voff + 256u < bytes)
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %3, %1, 0 offen offset:256 lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(srd), "s"(lds), "v"(voff) : "memory");
}
MF_DEV void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// C_ww (lower triangular 16 x 16 tile, given by ROWS: lane r holds a[k] = C_ww[r][k]) -> its inverse: row-major in the slot (for the
// other wavefronts) and in the accumulator layout (out).  wv::tri_inv_tiles without the accumulator -> image -> rows detour.
template <typename T>
MF_DEV void tri_inv_rows(T (&a)[16], typename Tr<T>::v4& out, T* slot, const Lane& ln, LogAcc<T>& la, bool& bad) {
    using D = wv::Dpp<T>;
    using wv::sfor;
    using wv::sfor2;
    T x[16], dinv[16];
    wv::fence(a);
    sfor<16>([&](auto k) {
        constexpr int kk = decltype(k)::value;
        const T cc = D::template bcast<kk>(a[kk]);
        bad |= !(cc != T(0));
        dinv[kk] = t_rcp<T>(cc);
        la.mul(cc);
        if constexpr (kk == 7) la.renorm();
    });
    la.renorm();
    sfor<16>([&](auto i) { x[decltype(i)::value] = (ln.r == decltype(i)::value) ? T(1) : T(0); });
    sfor<16>([&](auto k) {
        constexpr int kk = decltype(k)::value;
        x[kk] *= dinv[kk];
        sfor2<kk + 1, 16>([&](auto i) { D::template fnmac<decltype(i)::value>(x[decltype(i)::value], a[kk], x[kk]); });
    });
    typename Tr<T>::v4 o[1];
    wv::cols_out<T, 1, false>(x, o, slot, ln);
    out = o[0];
}

// ---- the workgroup's LDS ---------------------------------------------------------------------------------------------------------------
template <typename T, int NT, int MT> struct Lds {
    using G = PG<T, NT>;
    static constexpr int MP = 16 * MT;
    static constexpr int LDH = MP + (sizeof(T) == 4 ? 4 : 2);
    static constexpr int VECS = 4 * G::DP + MP + 64;      // the vectors come first: low LDS addresses for the DMA destinations
    static constexpr int ELEMS = VECS + 3 * G::IMG + G::DP * LDH + MP * LDH + NT * G::SLOT;
    static constexpr int BYTES = ELEMS * (int)sizeof(T);
    T* base;
    MF_DEV T* vec(int i) const { return base + i * G::DP; }                   // 0: rn, 1: t, 2: z, 3: mvec
    MF_DEV T* ys() const { return vec(4); }
    MF_DEV T* red() const { return ys() + MP; }
    MF_DEV T* I(int i) const { return base + VECS + i * G::IMG; }
    MF_DEV T* IH() const { return I(3); }                                     // holds H (MP x DP)
    MF_DEV T* IR() const { return IH() + G::DP * LDH; }                       // holds R^-1 (MP x MP, symmetric)
    MF_DEV T* slot(int i) const { return IR() + MP * LDH + i * G::SLOT; }     // inverse of diagonal tile i, row-major (mf_wave.hpp image)
};
enum { V_RN = 0, V_T = 1, V_Z = 2, V_M = 3 };

template <typename T, int NT, int MT> struct Ctx {
    Lds<T, NT, MT> sm;
    Lane ln;
    int w;               // wavefront = panel index (wave-uniform)
};

// ---- C (lower triangular panels, tiles ti >= w) -> Ci = C^-1 (lower).  Barriers: 1.  Uses image 1 (holds C^T) and the slots. -----------
// FROM_IMG: C^T is already in image 1 - every wavefront fetched its OWN sixteen rows of C there by LDS-DMA (an image row of C^T is a
// row of C) - and C is not in registers at all; the wavefront's diagonal tile is read by rows straight from the image.
template <typename T, int NT, int MT, bool FROM_IMG>
MF_DEV void tri_inv_panel(const Panel<T, NT>& C, Panel<T, NT>& Ci, const Ctx<T, NT, MT>& c, LogAcc<T>& la, bool& bad) {
    using v4 = typename Tr<T>::v4;
    using G = PG<T, NT>;
    const Lane& ln = c.ln;
    T* I1 = c.sm.I(1);
    v4 o[1];
    if constexpr (FROM_IMG) {
        dma_wait();
        T a[16];
        const T* row = I1 + (16 * c.w + ln.r) * G::LD + 16 * c.w;
        MF_UNROLL for (int k4 = 0; k4 < 4; ++k4) {
            const v4 t4 = *reinterpret_cast<const typename V4A<T>::type*>(row + 4 * k4);
            MF_UNROLL for (int e = 0; e < 4; ++e) a[4 * k4 + e] = t4[e];
        }
        tri_inv_rows<T>(a, o[0], c.sm.slot(c.w), ln, la, bad);
    } else {
        MF_UNROLL for (int ti = 1; ti < NT; ++ti)
            if (ti > c.w) img_put_t<T, G::LD>(I1, ti, c.w, C.t[ti], ln);
        v4 in[1] = {C.t[0]};
        MF_UNROLL for (int j = 1; j < NT; ++j) if (is_wave(c.w, j)) in[0] = C.t[j];
        wv::tri_inv_tiles<T, 1, false>(in, o, c.sm.slot(c.w), ln, la, bad);       // the slot keeps Ci_ww row-major
    }
    __syncthreads();
    Ci.zero();
    MF_UNROLL for (int j = 0; j < NT; ++j) if (is_wave(c.w, j)) Ci.t[j] = o[0];
    // forward substitution down the wavefront's own column: Ci(i, w) = -Ci_ii sum_{k = w}^{i - 1} C(i, k) Ci(k, w)
    MF_UNROLL for (int i = 1; i < NT; ++i) {
        if (i <= c.w) continue;
        v4 acc = {0, 0, 0, 0};
        MF_UNROLL for (int k = 0; k < i; ++k) {
            if (k < c.w) continue;
            acc = mm<T>(img_tile<T, G::LD>(I1, k, i, ln), Ci.t[k], acc);          // tile (k, i) of C^T: the A operand of C(i, k) Q
        }
        v4 dt;
        wv::image_to_tile_t<T>(dt, c.sm.slot(i), ln);                             // Ci_ii^T: the A operand of Ci_ii Q
        Ci.t[i] = -mm<T>(dt, acc, v4{0, 0, 0, 0});
    }
}

// ---- Phi (symmetric, tiles ti <= w valid; consumed) -> LiT = chol(Phi)^-T (upper: tiles ti <= w).  Barriers: 2 NT - 1. ------------------
// Uses image 1 (holds U = L^T), image 2 (holds U^T) and the slots.  first(): called behind the first barrier (whatever the workgroup
// read from the images before the call has been read by then).
// want_l (the emit pass of the factorisation): also the wavefront's diagonal tile of the factor itself, L_ww in the
// accumulator layout; with it and Phi.t[j] = U(j, w) = L(w, j)^T, j < w, the wavefront holds block row w of L on return.
template <typename T, int NT, int MT, typename First>
MF_DEV void chol_inv_panel(Panel<T, NT>& Phi, Panel<T, NT>& LiT, const Ctx<T, NT, MT>& c, LogAcc<T>& la, bool& bad, First first,
                           bool want_l, typename Tr<T>::v4& Ldiag) {
    using v4 = typename Tr<T>::v4;
    using G = PG<T, NT>;
    const Lane& ln = c.ln;
    T *I1 = c.sm.I(1), *I2 = c.sm.I(2);
    v4 own = {0, 0, 0, 0};
    MF_UNROLL for (int j = 0; j < NT; ++j) {
        if (is_wave(c.w, j)) {
            if (want_l) {
                wv::chol_fact_tile<T>(Phi.t[j], Ldiag, own, c.sm.slot(j), ln, bad);   // (same image afterwards; no log-determinant)
            } else {
            const v4 in[1] = {Phi.t[j]};
            v4 o[1];
            wv::chol_inv_tiles<T, 1, true>(in, o, c.sm.slot(j), ln, la, bad);     // the slot keeps Li_jj = L_jj^-1 row-major
            own = o[0];                                                           // Li_jj^T = U_jj^-1 in the accumulator layout
            }
        }
        __syncthreads();
        if (j == 0) first();
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
    Panel<T, NT> Phi;      // symmetric, all tiles formed (the factorisation reads the tiles ti <= w): pivot of the current block
    Panel<T, NT> X;        // coupling current block <-> the chunk's left separator
    Panel<T, NT> GU;       // symmetric, all tiles formed: accumulated contribution to the separator's pivot
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
//   ST_FROM_S: S is the wavefront's panel of the coupling S (level 0: transposed through image 0);
//   else:      S already holds the panel of S^T (reduction levels: the coupling is read transposed from memory).
// Images: 0 <- S^T, then WT;  1 <- U, then LiT;  2 <- U^T, then V.  A panel that has gone into its image is re-read from there as the B
// operand of a later product instead of staying in registers (S, V, WT: 48 registers at the step's widest point).
// Barriers: 2 NT - 1 + 3.  mid(): called behind the last barrier, in front of the last two products: image 1 and the vectors are
// free from there on - the place where the next step's inputs are requested (LDS-DMA), two products ahead of their use.
// block row w of a matrix whose TRANSPOSED column panel w the wavefront holds (tiles t[tj] = M^T[16 tj ..][16 w ..]): M[16 w + r][16 tj + row]
template <typename T, int NT>
MF_DEV void store_panel_rows_t(T* __restrict__ g, const typename Tr<T>::v4 (&t)[NT], int d, int w, const Lane& ln, int skip = -1) {
    MF_UNROLL for (int tj = 0; tj < NT; ++tj) {
        if (tj == skip) continue;
        MF_UNROLL for (int e = 0; e < 4; ++e) {
            const int i = 16 * w + ln.r, j = 16 * tj + Tr<T>::row(ln.q, e);
            if (i < d && j < d) g[i * d + j] = t[tj][e];
        }
    }
}
// gl / gw (the emit pass of SymmetricBlockTriDiagonal.cholesky, block_tri_diag.py:423-436): the factor of the pivot that is
// eliminated goes to gl (lower triangular, zeros above), W = S L^-T to gw; either may be NULL.
template <typename T, int NT, int MT, bool ST_FROM_S, typename Mid>
MF_DEV void eliminate_advance(PanelElim<T, NT>& E, Panel<T, NT>& S, const Panel<T, NT>& Dn, T rn, bool spike,
                              const Ctx<T, NT, MT>& c, Mid mid, Stamp& stamp, T* __restrict__ gl = nullptr, T* __restrict__ gw = nullptr,
                              int d_emit = 0) {
    using G = PG<T, NT>;
    const Lane& ln = c.ln;
    const int w = c.w;
    T *I0 = c.sm.I(0), *I1 = c.sm.I(1), *I2 = c.sm.I(2);
    vec_put<T>(c.sm.vec(V_T), w, E.t, ln);
    T z;
    {
        Panel<T, NT> LiT;
        typename Tr<T>::v4 Ld = {0, 0, 0, 0};
        chol_inv_panel<T, NT, MT>(E.Phi, LiT, c, E.laL, E.bad, [&]() __attribute__((always_inline)) {
            if (ST_FROM_S) {
                MF_UNROLL for (int ti = 0; ti < NT; ++ti) img_put_t<T, G::LD>(I0, ti, w, S.t[ti], ln);
            }
        }, gl != nullptr, Ld);
        if (gl) {   // block row w of L: U(j, w)^T for j < w, the diagonal tile, zeros to the right
            typename Tr<T>::v4 rowt[NT];
            MF_UNROLL for (int tj = 0; tj < NT; ++tj) rowt[tj] = tj < w ? E.Phi.t[tj] : typename Tr<T>::v4{0, 0, 0, 0};
            store_panel_rows_t<T, NT>(gl, rowt, d_emit, w, ln, w);
            MF_UNROLL for (int e = 0; e < 4; ++e) {
                const int i = 16 * w + Tr<T>::row(ln.q, e), j = 16 * w + ln.r;
                if (i < d_emit && j < d_emit) gl[i * d_emit + j] = Ld[e];
            }
        }
        MF_PSTAMP(stamp, 5)
        RV<T, NT> t_rv;
        vec_rv<T, NT>(t_rv, c.sm.vec(V_T), ln);
        z = mv_panel<T, NT>(LiT.t, t_rv.v);            // z = Li t
        MF_UNROLL for (int ti = 0; ti < NT; ++ti)
            if (ti <= w) img_put<T, G::LD>(I1, ti, w, LiT.t[ti], ln);
    }
    E.quad = fma_t(z, z, E.quad);
    vec_put<T>(c.sm.vec(V_Z), w, z, ln);
    __syncthreads();
    MF_PSTAMP(stamp, 6)
    Panel<T, NT> WT;
    {
        Panel<T, NT> ST;
        if (ST_FROM_S) {
            MF_UNROLL for (int tk = 0; tk < NT; ++tk) ST.t[tk] = img_tile<T, G::LD>(I0, tk, w, ln);
        } else {
            ST = S;
        }
        tn_img<T, NT, NT, G::LD, OP_SET, P_UPPER>(WT, I1, ST.t, ln);     // W^T = Li S^T
    }
    if (gw) store_panel_rows_t<T, NT>(gw, WT.t, d_emit, w, ln);
    if (spike) {
        Panel<T, NT> V;
        tn_img<T, NT, NT, G::LD, OP_SET, P_UPPER>(V, I1, E.X.t, ln);     // V = Li X
        RV<T, NT> z_rv;
        vec_rv<T, NT>(z_rv, c.sm.vec(V_Z), ln);
        E.gU -= mv_panel<T, NT>(V.t, z_rv.v);                                                            // gU -= V^T z
        MF_UNROLL for (int ti = 0; ti < NT; ++ti) img_put<T, G::LD>(I2, ti, w, V.t[ti], ln);
    }
    __syncthreads();
    MF_PSTAMP(stamp, 7)
    MF_UNROLL for (int ti = 0; ti < NT; ++ti) img_put<T, G::LD>(I0, ti, w, WT.t[ti], ln);
    {
        RV<T, NT> z_rv;
        vec_rv<T, NT>(z_rv, c.sm.vec(V_Z), ln);
        E.t = rn - mv_panel<T, NT>(WT.t, z_rv.v);                                                        // t = rn - W z
    }
    if (spike) {
        Panel<T, NT> Vq;
        MF_UNROLL for (int tk = 0; tk < NT; ++tk) Vq.t[tk] = img_tile<T, G::LD>(I2, tk, w, ln);
        tn_img<T, NT, NT, G::LD, OP_SUB, P_FULL>(E.GU, I2, Vq.t, ln);      // GU -= V^T V
    }
    __syncthreads();
    MF_PSTAMP(stamp, 8)
    mid();
    MF_UNROLL for (int ti = 0; ti < NT; ++ti) E.Phi.t[ti] = Dn.t[ti];
    {
        Panel<T, NT> Wq;
        MF_UNROLL for (int tk = 0; tk < NT; ++tk) Wq.t[tk] = img_tile<T, G::LD>(I0, tk, w, ln);
        tn_img<T, NT, NT, G::LD, OP_SUB, P_FULL>(E.Phi, I0, Wq.t, ln);     // Phi = Dn - W W^T
    }
    if (spike) {
        Panel<T, NT> Vq;
        MF_UNROLL for (int tk = 0; tk < NT; ++tk) Vq.t[tk] = img_tile<T, G::LD>(I2, tk, w, ln);
        tn_img<T, NT, NT, G::LD, OP_NEG, P_FULL>(E.X, I0, Vq.t, ln);              // X = -W V
    }
    MF_PSTAMP(stamp, 9)
}

// The last block of a final reduction: factor, z, nothing to advance to.
template <typename T, int NT, int MT> MF_DEV void eliminate_last(PanelElim<T, NT>& E, const Ctx<T, NT, MT>& c) {
    const int w = c.w;
    vec_put<T>(c.sm.vec(V_T), w, E.t, c.ln);
    Panel<T, NT> LiT;
    typename Tr<T>::v4 unused = {0, 0, 0, 0};
    chol_inv_panel<T, NT, MT>(E.Phi, LiT, c, E.laL, E.bad, [] {}, false, unused);
    RV<T, NT> t_rv;
    vec_rv<T, NT>(t_rv, c.sm.vec(V_T), c.ln);
    const T z = mv_panel<T, NT>(LiT.t, t_rv.v);
    E.quad = fma_t(z, z, E.quad);
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
    store_panel<T, NT, false>(out.Dv + idx * dd, E.Phi, d, c.w, c.ln);
    store_panel<T, NT, false>(out.GU + idx * dd, E.GU, d, c.w, c.ln);
    store_panel<T, NT, false>(out.F + idx * dd, E.X, d, c.w, c.ln);
    const int j = 16 * c.w + c.ln.r;
    if (out.tv && c.ln.q == 0 && j < d) {
        out.tv[idx * d + j] = E.t;
        out.gU[idx * d + j] = E.gU;
    }
    if (out.sc && threadIdx.x == 0) out.sc[idx] = scalar;
}

template <typename T, int NT, int MT> constexpr int panel_wpe() { return NT <= 2 ? (sizeof(T) == 4 ? 4 : 2) : (sizeof(T) == 4 ? 2 : 1); }

// Level 0: workgroup (s, c) eliminates the transitions [c L, min((c+1) L, T-1)) of series s.
// PREC: the same per-block terms WITHOUT the elimination - StateSpaceModel._build_precision (+ H^T R^-1 H, + the information vector;
// state_space_model.py:431-483, kalman_filter.py:86-101,153-156) for a chunk of blocks: diag_k = Q_k^-1 + A_{k+1}^T Q_{k+1}^-1 A_{k+1}
// (+ H^T R^-1 H), sub_k = -Q_{k+1}^-1 A_{k+1}, eta_k = Q_k^-1 m_k - A_{k+1}^T Q_{k+1}^-1 m_{k+1} (+ H^T R^-1 y).  The chunk inverts the
// factor of its first block itself (one inversion more per chunk); a.H == NULL: the prior precision, po.eta == NULL: no vector,
// a.y == NULL: no observation term in it.  (Rounds 2-5: a workgroup per block on LDS tiles, two inversions per block: 2.3 ms at
// config 5's shape.)
template <typename T> struct PrecOut {
    T *diag, *sub, *eta;
};
template <typename T, int NT, int MT, bool EX, bool PREC = false>
__global__ void __launch_bounds__(64 * NT) __attribute__((amdgpu_waves_per_eu(panel_wpe<T, NT, MT>(), panel_wpe<T, NT, MT>())))
panel_kf_chunk_kernel(wv::WvArgs<T> a, RedSys<T> out, PrecOut<T> po = PrecOut<T>{nullptr, nullptr, nullptr}) {
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
    if (!a.rinv_per_step && (!PREC || a.H)) stage_rinv(a.Rinv);

    Panel<T, NT> Dn, Am;
    typename Tr<T>::v4 Hp[MT];
    if constexpr (PREC) { MF_UNROLL for (int to = 0; to < MT; ++to) Hp[to] = typename Tr<T>::v4{0, 0, 0, 0}; }
    T rn = T(0);
    // The block's own terms from its Cholesky factor C: Dn = Q^-1 (all tiles), rn = Q^-1 mvec; image 1 <- Q^-1; the observation rows
    // (-> Hp) and the transition (Ag != NULL, -> Am) are loaded behind the inversion of C - their latency is covered by the product that
    // follows - and go into their images.  Ends with the barrier that publishes them.
    // DMA: the next block's chol(Q) goes straight into image 1 (= the image of C^T), fp32 and d = 16 NT only (an image row is then a
    // row of C, at most 256 bytes; the fp64 image permutes its columns); the offset and observation vectors go by DMA in every case.
    constexpr bool DMA = EX && sizeof(T) == 4;
    // request the inputs that open the step of transition tau (block blk = tau + 1; tau < 0: the prior block 0).  Image 1, V_M and ys
    // must be free.  Wavefront w fetches the rows 16 w ... 16 w + 15 of C; wavefront 0 the offset vector, the last one the observations.
    auto request = [&](long tau) __attribute__((always_inline)) {
        const T* Cg = tau < 0 ? a.cholP0 + s * dd : a.cholQ + (s * nt + tau) * dd;
        const T* mv = tau < 0 ? a.mu0 + s * d : a.b + (s * nt + tau) * d;
        const long blk = tau + 1;
        if constexpr (DMA)
            dma_rows16<G::DP * (int)sizeof(T), (G::LD - G::DP) * (int)sizeof(T)>(dma_srd(Cg + 16 * w * G::DP, 16 * G::DP * sizeof(T)),
                                                         lds_addr(I1 + 16 * w * G::LD), 4u * (threadIdx.x & 63));
        if (w == 0 && (!PREC || po.eta)) dma_vec(mv, d * sizeof(T), c.sm.vec(V_M), G::DP * sizeof(T));
        if (w == NT - 1 && (!PREC || (a.H && a.y))) dma_vec(a.y + (s * a.Tn + blk) * m, m * sizeof(T), c.sm.ys(), L::MP * sizeof(T));
    };
    // The block's own terms from its Cholesky factor: Dn = Q^-1 (all tiles), rn = Q^-1 mvec; image 1 <- Q^-1; the observation rows
    // (-> Hp) and the transition (Ag != NULL, -> Am) are loaded behind the inversion of C - their latency is covered by the product that
    // follows - and go into their images.  Cg: the factor in memory (read here unless it came by DMA).  Ends with the barrier that
    // publishes the images.
    auto own_terms = [&](const T* __restrict__ Cg, long blk, const T* __restrict__ Ag, Stamp& stamp) __attribute__((always_inline)) {
        if (a.rinv_per_step && (!PREC || a.H)) stage_rinv(a.Rinv + (s * a.Tn + blk) * m * m);
        Panel<T, NT> Ci;
        if constexpr (DMA) {
            tri_inv_panel<T, NT, MT, true>(Ci, Ci, c, laC, E.bad);
        } else {
            Panel<T, NT> C;
            load_panel<T, NT, EX>(C, Cg, d, w, true, true, ln);
            dma_wait();                                       // (the vectors; C's loads are the compiler's)
            tri_inv_panel<T, NT, MT, false>(C, Ci, c, laC, E.bad);
        }
        MF_PSTAMP(stamp, 0)
        if (Ag) load_panel<T, NT, EX>(Am, Ag, d, w, false, false, ln);
        if (!PREC || a.H) load_rows_panel<T, MT>(Hp, a.H + (s * a.Tn + blk) * m * d, m, d, w, ln);
        MF_UNROLL for (int ti = 0; ti < NT; ++ti)
            if (ti >= w) img_put<T, G::LD>(I2, ti, w, Ci.t[ti], ln);
        __syncthreads();
        MF_PSTAMP(stamp, 1)
        tn_img<T, NT, NT, G::LD, OP_SET, P_LOWER>(Dn, I2, Ci.t, ln);                                            // Q^-1 = Ci^T Ci
        {
            RV<T, NT> mv;
            vec_rv<T, NT>(mv, c.sm.vec(V_M), ln);
            rn = mv_panel<T, NT>(Dn.t, mv.v);                                                                   // Q^-1 mvec
            acc_ww = fma_t(rn, c.sm.vec(V_M)[16 * w + ln.r], acc_ww);                                           // mvec^T Q^-1 mvec
        }
        vec_put<T>(c.sm.vec(V_RN), w, rn, ln);
        MF_UNROLL for (int ti = 0; ti < NT; ++ti) img_put<T, G::LD>(I1, ti, w, Dn.t[ti], ln);
        MF_UNROLL for (int to = 0; to < MT; ++to) img_put<T, L::LDH>(IH, to, w, Hp[to], ln);
        if (Ag) {
            MF_UNROLL for (int ti = 0; ti < NT; ++ti) img_put<T, G::LD>(I0, ti, w, Am.t[ti], ln);
        }
        __syncthreads();
        MF_PSTAMP(stamp, 2)
    };
    // the observation terms on top: Dn += H^T R^-1 H, rn += H^T R^-1 y, y^T R^-1 y (kalman_filter.py:86-101)
    auto obs_terms = [&](long blk) __attribute__((always_inline)) {
        v4 Gp[MT];
        {
            Panel<T, MT> Gq;
            tn_img<T, MT, MT, L::LDH, OP_SET, P_FULL>(Gq, IR, Hp, ln);           // G = R^-1 H
            MF_UNROLL for (int to = 0; to < MT; ++to) Gp[to] = Gq.t[to];
        }
        tn_img<T, NT, MT, L::LDH, OP_ADD, P_FULL>(Dn, IH, Gp, ln);                // += H^T G
        {
            T yv[MT][4];
            MF_UNROLL for (int to = 0; to < MT; ++to)
                MF_UNROLL for (int e = 0; e < 4; ++e) yv[to][e] = c.sm.ys()[16 * to + Tr<T>::row(ln.q, e)];
            if (!PREC || a.y) rn += mv_panel<T, MT>(Gp, yv);                                            // += G^T y
        }
        {   // y^T R^-1 y: thread (o, part) takes the terms p = part, part + NPART, ... of row o (R^-1 symmetric: read down a column)
            constexpr int NPART = 64 * NT / L::MP;
            const int o = threadIdx.x % L::MP, part = threadIdx.x / L::MP;
            if (part < NPART) {
                T acc = T(0);
                for (int p = part; p < L::MP; p += NPART) acc = fma_t(IR[p * L::LDH + 16 * (o >> 4) + pos16<T>(o & 15)], c.sm.ys()[p], acc);
                acc_yry = fma_t(acc, c.sm.ys()[o], acc_yry);
            }
        }
    };

    if constexpr (PREC) {
        Stamp st0;
        st0.init(c.sm.red() + 16);
        const int jv = 16 * w + ln.r;
        if (ch == 0) {
            request(-1);
            own_terms(a.cholP0 + s * dd, 0, nullptr, st0);
        } else {
            request(tau0 - 1);
            own_terms(a.cholQ + (s * nt + tau0 - 1) * dd, tau0, nullptr, st0);
        }
        if (a.H) obs_terms(tau0);
        Panel<T, NT> Dp = Dn;                      // the diagonal block in the making and its share of the vector
        T ep = rn;
        __syncthreads();
        if (len > 0) request(tau0);
        for (long j = 0; j < len; ++j) {
            const long tau = tau0 + j, blk = tau + 1;
            asm volatile("" : "+v"(ln.r), "+v"(ln.q));
            if constexpr (!EX) asm volatile("" : "+s"(d));
            own_terms(a.cholQ + (s * nt + tau) * dd, blk, a.A + (s * nt + tau) * dd, st0);
            Panel<T, NT> S;
            tn_img<T, NT, NT, G::LD, OP_NEG, P_FULL>(S, I1, Am.t, ln);           // S = -Q^-1 A
            T btw;
            {
                RV<T, NT> rn_rv;
                vec_rv<T, NT>(rn_rv, c.sm.vec(V_RN), ln);
                btw = mv_panel<T, NT>(Am.t, rn_rv.v);                            // A^T Q^-1 m
            }
            if (a.H) obs_terms(blk);
            tn_img<T, NT, NT, G::LD, OP_SUB, P_FULL>(Dp, I0, S.t, ln);           // + A^T Q^-1 A
            ep -= btw;
            store_panel<T, NT, false>(po.diag + (s * a.Tn + tau) * dd, Dp, d, w, ln);
            store_panel<T, NT, false>(po.sub + (s * nt + tau) * dd, S, d, w, ln);
            if (po.eta && ln.q == 0 && jv < d) po.eta[(s * a.Tn + tau) * d + jv] = ep;
            Dp = Dn;
            ep = rn;
            __syncthreads();
            if (j + 1 < len) request(tau + 1);
        }
        if (tau0 + len == nt) {                    // the last block of the series: its own terms are all it has
            store_panel<T, NT, false>(po.diag + (s * a.Tn + nt) * dd, Dp, d, w, ln);
            if (po.eta && ln.q == 0 && jv < d) po.eta[(s * a.Tn + nt) * d + jv] = ep;
        }
        return;
    }
    if (ch == 0) {   // block 0: the prior
        Stamp st0;
        st0.init(c.sm.red() + 16);
        request(-1);
        own_terms(a.cholP0 + s * dd, 0, nullptr, st0);
        obs_terms(0);
        MF_UNROLL for (int ti = 0; ti < NT; ++ti) E.Phi.t[ti] = Dn.t[ti];
        E.t = rn;
        __syncthreads();
    }
    if (len > 0) request(tau0);
    Stamp stamp;
    stamp.init(c.sm.red() + 16);
    long first_bad = -1;          // the block whose elimination step first met a non-positive pivot (per wavefront)
    for (long j = 0; j < len; ++j) {
        const long tau = tau0 + j, blk = tau + 1;
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        if constexpr (!EX) asm volatile("" : "+s"(d));
        MF_PSTAMP(stamp, 11)
        own_terms(a.cholQ + (s * nt + tau) * dd, blk, a.A + (s * nt + tau) * dd, stamp);
        Panel<T, NT> S;
        tn_img<T, NT, NT, G::LD, OP_NEG, P_FULL>(S, I1, Am.t, ln);               // S = -Q^-1 A
        T btw;
        {
            RV<T, NT> rn_rv;
            vec_rv<T, NT>(rn_rv, c.sm.vec(V_RN), ln);
            btw = mv_panel<T, NT>(Am.t, rn_rv.v);                                                       // A^T Q^-1 mvec
        }
        MF_PSTAMP(stamp, 3)
        obs_terms(blk);
        MF_PSTAMP(stamp, 4)
        if (j == 0 && spike) {
            // the block on the left is the chunk's separator: its coupling seeds the spike
            tn_img<T, NT, NT, G::LD, OP_NEG, P_FULL>(E.GU, I0, S.t, ln);   // GU = A^T Q^-1 A
            E.X = S;
            E.gU = -btw;
            MF_UNROLL for (int ti = 0; ti < NT; ++ti) E.Phi.t[ti] = Dn.t[ti];
            E.t = rn;
            __syncthreads();
            if (j + 1 < len) request(tau + 1);
        } else {
            tn_img<T, NT, NT, G::LD, OP_SUB, P_FULL>(E.Phi, I0, S.t, ln);  // D_{k-1} += A^T Q^-1 A
            E.t -= btw;
            eliminate_advance<T, NT, MT, true>(E, S, Dn, rn, spike, c, [&]() __attribute__((always_inline)) {
                if (j + 1 < len) request(tau + 1);
            }, stamp);
            if (E.bad && first_bad < 0) first_bad = tau;
        }
    }
    stamp.print(len);
    // the chunk's scalar: every wavefront's share, summed
    T part = T(-0.5) * wv::sum16<T>(acc_ww) + T(0.5) * wv::sum16<T>(E.quad) - laC.value() - T(0.5) * E.laL.value();
    part += T(-0.5) * wv::sum16<T>(wv::xor_rows<T>(acc_yry));
    const T scalar = wg_sum<T, NT>(part, c.sm.red(), w);
    store_chunk_panel<T, NT, MT>(out, id, d, E, scalar, c);
    if (__any(E.bad) && (threadIdx.x & 63) == 0 && a.info) raise_pivot(a.info, s * a.Tn + (first_bad < 0 ? tau0 : first_bad));
}

// Reduction level: RedSys(n) -> RedSys(P) (FINAL: P = 1, the last block is eliminated too and out_scalar written).  Same block
// sequence as big_red_kernel (mf_big_impl.hpp).
template <typename T, int NT, bool FINAL, bool EX>
__global__ void __launch_bounds__(64 * NT) __attribute__((amdgpu_waves_per_eu(panel_wpe<T, NT, 1>(), panel_wpe<T, NT, 1>())))
panel_red_kernel(RedSys<T> in, RedSys<T> out, long B, long P, int d_, T add_const, T* __restrict__ out_scalar, int* info,
                 long Lc = 0, int rev = 0, T* __restrict__ pivs = nullptr) {
    using L = Lds<T, NT, 1>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Ctx<T, NT, 1> c{L{reinterpret_cast<T*>(smem_raw)}, Lane{(int)(threadIdx.x & 15), (int)((threadIdx.x >> 4) & 3)},
                    __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))};
    Lane& ln = c.ln;
    const int w = c.w;
    int d = EX ? 16 * NT : d_;
    const long id = blockIdx.x, s = id / P, ch = id % P;
    // Lc > 0: OPERATOR mode - the up-sweep of the time-partitioned Cholesky / U D U^T factorisation of a user matrix (in.Dv = diag,
    // in.F = sub with f_stride = n - 1, f_off = -1, no vectors): chunks of Lc blocks as the tile engine's boundary and emit passes
    // expect them (mf_bigpar_impl.hpp), rev: the blocks in reversed order with the couplings transposed (upper_diagonal_lower)
    const long k0 = Lc > 0 ? ch * Lc : (ch * in.n) / P;
    const long k1 = Lc > 0 ? (k0 + Lc < in.n ? k0 + Lc : in.n) : ((ch + 1) * in.n) / P;
    const bool spike = !FINAL && k0 > 0;
    if (Lc > 0 && k0 >= in.n) return;             // (an empty last chunk of the operator partition: the whole workgroup leaves)
    const long dd = long(d) * d;
    PanelElim<T, NT> E;
    E.init();
    T acc_sc = T(0);
    // block k's inputs: own pivot / right-hand-side parts (Dv[k] + GU[k + 1]: the next chunk's contribution to its separator) and the
    // coupling to block k - 1, transposed.  They are requested one block AHEAD, in front of the elimination of block k - 1.
    struct Blk {
        Panel<T, NT> Dn, G2, FT;
        T rn, sc;
    };
    auto load_blk = [&](Blk& bk, long k, bool coupling) __attribute__((always_inline)) {
        const long idx = s * in.n + (rev ? in.n - 1 - k : k);
        const bool has_next = in.GU && (k + 1 < in.n);
        bk.sc = in.sc ? in.sc[idx] : T(0);
        load_panel<T, NT, EX>(bk.Dn, in.Dv + idx * dd, d, w, false, true, ln);
        bk.rn = in.tv ? load_cv_g<T>(in.tv + idx * d, d, w, ln) : T(0);
        if (has_next) {
            load_panel<T, NT, EX>(bk.G2, in.GU + (idx + 1) * dd, d, w, false, false, ln);
            if (in.gU) bk.rn += load_cv_g<T>(in.gU + (idx + 1) * d, d, w, ln);
        } else {
            bk.G2.zero();
        }
        if (coupling) {
            if (rev) load_panel<T, NT, EX>(bk.FT, in.F + (s * in.f_stride + in.n - 1 - k) * dd, d, w, false, false, ln);
            else load_panel_t<T, NT, EX>(bk.FT, in.F + (s * in.f_stride + k + in.f_off) * dd, d, w, ln);
        }
    };
    {
        // the chunk's first block: nothing to eliminate yet; its coupling (k0 > 0) is the one to the chunk's left separator
        Blk b0;
        load_blk(b0, k0, false);
        if (k0 > 0 && !FINAL) {
            if (rev) load_panel_t<T, NT, EX>(E.X, in.F + (s * in.f_stride + in.n - 1 - k0) * dd, d, w, ln);
            else load_panel<T, NT, EX>(E.X, in.F + (s * in.f_stride + k0 + in.f_off) * dd, d, w, false, false, ln);
        }
        MF_UNROLL for (int ti = 0; ti < NT; ++ti) E.Phi.t[ti] = b0.Dn.t[ti] + b0.G2.t[ti];
        E.t = b0.rn;
        acc_sc += b0.sc;
        // pivs (operator mode, one workgroup per series over the chunk ends): the pivot of every block WITHOUT the next block's
        // contribution - the natural-order pivots the emit passes restart from (bigpar_chol_boundary_kernel's output)
        if (pivs) store_panel<T, NT, false>(pivs + (s * in.n + k0) * dd, b0.Dn, d, w, ln);
    }
    Blk nx;
    if (k0 + 1 < k1) load_blk(nx, k0 + 1, true);
    for (long k = k0 + 1; k < k1; ++k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        if constexpr (!EX) asm volatile("" : "+s"(d));
        Panel<T, NT> Dn, FT = nx.FT, G2k = nx.G2;
        MF_UNROLL for (int ti = 0; ti < NT; ++ti) Dn.t[ti] = nx.Dn.t[ti] + nx.G2.t[ti];
        const T rn = nx.rn;
        acc_sc += nx.sc;
        if (k + 1 < k1) load_blk(nx, k + 1, true);
        Stamp stamp;
        stamp.init(c.sm.red() + 16);
        eliminate_advance<T, NT, 1, false>(E, FT, Dn, rn, spike, c, [] {}, stamp);
        if (pivs) {
            MF_UNROLL for (int ti = 0; ti < NT; ++ti) G2k.t[ti] = E.Phi.t[ti] - G2k.t[ti];
            store_panel<T, NT, false>(pivs + (s * in.n + k) * dd, G2k, d, w, ln);
        }
    }
    if (FINAL) eliminate_last<T, NT, 1>(E, c);
    T part = T(0.5) * wv::sum16<T>(E.quad) - T(0.5) * E.laL.value();
    const T tot = wg_sum<T, NT>(part, c.sm.red(), w);
    if (FINAL) {
        if (threadIdx.x == 0) out_scalar[s] = add_const + acc_sc + tot;
    } else if (out.Dv) {
        store_chunk_panel<T, NT, 1>(out, id, d, E, acc_sc + tot, c);
    }
    if (__any(E.bad) && (threadIdx.x & 63) == 0 && info) raise_info(info);
}

// The emit pass of the time-partitioned SymmetricBlockTriDiagonal.cholesky (block_tri_diag.py:423-436) for 32 < d <= 64: workgroup
// (series, chunk) restarts the textbook recursion from the natural-order pivot of the block in front of its chunk (piv, the pass
// over the chunk ends) and writes L_k and W_{k-1} = S_{k-1} L_{k-1}^-T for its blocks - eliminate_advance without a spike, with the
// factor and W taken out of it on the way (rounds 2-5: bigpar_chol_emit_kernel on LDS tiles, 1.66 ms at config 5's shape).
template <typename T, int NT, bool EX>
__global__ void __launch_bounds__(64 * NT) __attribute__((amdgpu_waves_per_eu(panel_wpe<T, NT, 1>(), panel_wpe<T, NT, 1>())))
panel_chol_emit_kernel(long B, long n, int d_, long P, long Lc, const T* __restrict__ diag, const T* __restrict__ sub,
                       const T* __restrict__ piv, T* __restrict__ ldiag, T* __restrict__ lsub, int* info) {
    using L = Lds<T, NT, 1>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Ctx<T, NT, 1> c{L{reinterpret_cast<T*>(smem_raw)}, Lane{(int)(threadIdx.x & 15), (int)((threadIdx.x >> 4) & 3)},
                    __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))};
    Lane& ln = c.ln;
    const int w = c.w;
    int d = EX ? 16 * NT : d_;
    const long s = blockIdx.x / P, ch = blockIdx.x % P;
    const long k0 = ch * Lc, k1 = k0 + Lc < n ? k0 + Lc : n, dd = long(d) * d;
    if (k0 >= n) return;
    PanelElim<T, NT> E;
    E.init();
    Stamp stamp;
    stamp.init(c.sm.red() + 16);
    // step j = k0 - 1 (ch > 0: the pivot in front of the chunk, only W is written), then k0 ... k1 - 2 (factor and W); block k1 - 1: factor
    const long j0 = ch > 0 ? k0 - 1 : k0;
    if (ch > 0) load_panel<T, NT, EX>(E.Phi, piv + (s * P + ch - 1) * dd, d, w, false, true, ln);
    else load_panel<T, NT, EX>(E.Phi, diag + (s * n) * dd, d, w, false, true, ln);
    Panel<T, NT> Dn, FT;
    auto load_step = [&](long j) __attribute__((always_inline)) {     // the inputs of eliminating block j: D_{j+1} and S_j^T
        load_panel<T, NT, EX>(Dn, diag + (s * n + j + 1) * dd, d, w, false, true, ln);
        load_panel_t<T, NT, EX>(FT, sub + (s * (n - 1) + j) * dd, d, w, ln);
    };
    if (j0 < k1 - 1) load_step(j0);
    for (long j = j0; j < k1 - 1; ++j) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        if constexpr (!EX) asm volatile("" : "+s"(d));
        Panel<T, NT> Dc = Dn, Fc = FT;
        if (j + 1 < k1 - 1) load_step(j + 1);
        const bool own = j >= k0;                    // (the factor of the pivot in front of the chunk belongs to the chunk before)
        eliminate_advance<T, NT, 1, false>(E, Fc, Dc, T(0), false, c, [] {}, stamp, own ? ldiag + (s * n + j) * dd : nullptr,
                                           lsub + (s * (n - 1) + j) * dd, d);
    }
    {   // the chunk's last block: its factor alone
        Panel<T, NT> LiT;
        typename Tr<T>::v4 Ld = {0, 0, 0, 0};
        chol_inv_panel<T, NT, 1>(E.Phi, LiT, c, E.laL, E.bad, [] {}, true, Ld);
        T* gl = ldiag + (s * n + k1 - 1) * dd;
        typename Tr<T>::v4 rowt[NT];
        MF_UNROLL for (int tj = 0; tj < NT; ++tj) rowt[tj] = tj < w ? E.Phi.t[tj] : typename Tr<T>::v4{0, 0, 0, 0};
        store_panel_rows_t<T, NT>(gl, rowt, d, w, ln, w);
        MF_UNROLL for (int e = 0; e < 4; ++e) {
            const int i = 16 * w + Tr<T>::row(ln.q, e), jj = 16 * w + ln.r;
            if (i < d && jj < d) gl[i * d + jj] = Ld[e];
        }
    }
    if (__any(E.bad) && (threadIdx.x & 63) == 0 && info) raise_info(info);
}

}  // namespace pn
}  // namespace mf
