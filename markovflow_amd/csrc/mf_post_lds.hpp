// posterior_state_space_model (kalman_filter.py:109-182) for FEW, LONG series, fused and partitioned in time, with the
// memory side of the log-likelihood kernel (mf_kf_lds.hpp: every lane's next transition brought in by LDS-DMA, consecutive
// lanes reading consecutive 16-B pieces of a row).  Three passes, arithmetic in mf_post_math.hpp:
//
//   1. post_lds_kernel<MODE = 0>      a lane per (series, chunk) walks its transitions from the last to the first, assembles
//      the posterior precision on the way (state_space_model.py:431-483, kalman_filter.py:86-101,149-156) and eliminates
//      with the fill-in carried towards the block on the chunk's right: one summary per chunk.  Reads (2 d^2 + d + m d + m) s
//      bytes per step, writes nothing but the summaries.
//   2. post_scan_kernel               a wavefront per series composes the summaries (Kogge-Stone over the lanes): the state of
//      the backward recursion (Psi, psi) at every chunk boundary.
//   3. post_lds_kernel<MODE = 1>      every chunk restarts the textbook backward recursion (block_tri_diag.py:438-545) from
//      its boundary and writes the posterior chain: reads the same bytes again, writes (2 d^2 + d) s per step.
//
// Against the route it replaces for B < 2048 (mf_ssm_precision -> parallel-in-time U D U^T -> affine scan -> emit kernels:
// precision and factor written and re-read, ~6x the algorithmic traffic through per-lane row loads) the inputs are read
// twice, coalesced, and every output once.
#pragma once
#include <type_traits>

#include "mf_kf_lds.hpp"
#include "mf_post_math.hpp"

namespace mf {

template <typename T> struct PostOut {
    T* a_post; T* mu0_post; T* b_post; T* cp0_post; T* cq_post;   // the posterior chain (EMIT)
    const T* bPsi; const T* bpsi;                                 // boundary state per consumer chunk [B, P, D, D] / [B, P, D] (EMIT)
};

// Workspace of the three passes: chunk summaries (Dv, GU, F: D*D each; tv, gU: D each; sc) and the boundary states (Psi: D*D,
// psi: D) per (series, chunk).  (mf_grad_lds.hpp reads both after the passes have run.)
template <typename T, int D> struct PostWs {
    RedSys<T> sum; T* bPsi; T* bpsi;
    static size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }
    static size_t bytes(long B, long P) {
        const size_t nb = size_t(B) * P;
        return align_up(nb * (3 * D * D + 2 * D + 1) * sizeof(T)) + align_up(nb * D * D * sizeof(T)) + align_up(nb * D * sizeof(T));
    }
    static PostWs carve(void* ws, long B, long P) {
        PostWs w;
        char* p = static_cast<char*>(ws);
        T* base = reinterpret_cast<T*>(p);
        const long nb = B * P;
        w.sum.Dv = base; w.sum.GU = w.sum.Dv + nb * D * D; w.sum.F = w.sum.GU + nb * D * D; w.sum.tv = w.sum.F + nb * D * D;
        w.sum.gU = w.sum.tv + nb * D; w.sum.sc = w.sum.gU + nb * D;
        w.sum.n = P; w.sum.f_stride = P; w.sum.f_off = 0;
        p += align_up(size_t(nb) * (3 * D * D + 2 * D + 1) * sizeof(T));
        w.bPsi = reinterpret_cast<T*>(p); p += align_up(size_t(B) * P * D * D * sizeof(T));
        w.bpsi = reinterpret_cast<T*>(p);
        return w;
    }
};

// ---- coalesced output rows, one store at a time ---------------------------------------------------------------------------
// A lane produces whole rows of the posterior chain (A'_t: d x d, cholQ'_t: d x d, b'_t: d) for ITS chunk; stored directly,
// one store instruction of the wave touches 64 different 128-B lines.  Instead the rows go through an LDS staging buffer:
// every lane writes (half of) its row, then store instruction i of the wave moves unit 64 i + lane of the row-major image,
// i.e. consecutive lanes store consecutive 16-B units of a row - the mirror image of the LDS-DMA loads.  `buffer_store` with
// the row offsets of the input streams (same shapes); a row that must not be written (lane without a chunk, chunk not yet
// active) gets an out-of-range offset, which the buffer range check drops.
//
// WHEN the stores are issued (measured on MI355X, B=1024, T=10000, d=6 fp64; profiles/r04_post_store_path.txt,
// scripts/micro/store_rate.hip): a SIMD's store path takes one 1-KB store instruction per write round trip - 50-85 cycles on
// an idle chip, ~330 while every CU streams 3 TB/s of reads - and the issuing wave is held only when its NEXT store finds the
// path busy; arithmetic between two stores hides it (64 fp64 FMAs between stores: the store costs nothing).  Issued in
// bursts where the rows become available, a step's 39 stores cost the wave ~13 k cycles on top of 15 k of arithmetic (emit
// pass 3.9 ms against 1.8 ms with every store dropped by the range check).  So the emit step (post_emit_step) calls
// `tick<SITE>()` every ~30 multiply-adds and the sink issues ONE store per tick from the piece that is staged: the first half
// of chol(Delta^-1) and b' while the products up to A' run, its second half after them, the first half of A' during the rest
// of the step and the second half during the NEXT step's factorisation (its rows wait in the staging buffer across the loop's
// back edge).  Tried and measured slower: `nt` stores (4.6 ms), 256-B aligned rows (3.6 ms: partial lines are not the
// cost), all stores of a CU issued by a fourth, storing wavefront fed through LDS (4.3-4.8 ms: one SIMD's store path carries
// a third of what three carry).
// what DmaStream needs to know of a piece: NU units of UNIT bytes per row, all kept
template <int NU, int UNIT_> struct OutPiece {
    static constexpr int U = NU, NI = NU, UNIT = UNIT_, UG = NU;
    static constexpr bool ALL = true;
};
template <int UNIT> struct OutWord;
template <> struct OutWord<16> { typedef int type __attribute__((ext_vector_type(4))); };
template <> struct OutWord<8> { typedef int type __attribute__((ext_vector_type(2))); };
template <> struct OutWord<4> { typedef int type; };
// The read-back of the staged image goes to ACCUMULATION registers and is stored from there ("a" operands; gfx950 has one
// unified register file, ds_read and buffer_store take AGPRs): the emit step keeps all 256 VGPRs busy.
// Hazards hipcc cannot see inside asm: the descriptor may just have been written by v_readfirstlane (5 wait states before a
// VMEM instruction reads an SGPR a VALU instruction wrote: leading s_nop 4), and a store of more than 64 bits needs one wait
// state before its data registers are overwritten (trailing s_nop 0).
MF_DEV void lds_read_a(OutWord<16>::type& v, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=a"(v) : "v"(addr) : "memory"); }
MF_DEV void lds_read_a(OutWord<8>::type& v, unsigned addr) { asm volatile("ds_read_b64 %0, %1" : "=a"(v) : "v"(addr) : "memory"); }
MF_DEV void lds_read_a(OutWord<4>::type& v, unsigned addr) { asm volatile("ds_read_b32 %0, %1" : "=a"(v) : "v"(addr) : "memory"); }
MF_DEV void buf_store(OutWord<16>::type v, mf_v4i srd, unsigned voff) {
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 0" :: "a"(v), "v"(voff), "s"(srd) : "memory");
}
MF_DEV void buf_store(OutWord<8>::type v, mf_v4i srd, unsigned voff) {
    asm volatile("s_nop 4\n\tbuffer_store_dwordx2 %0, %1, %2, 0 offen\n\ts_nop 0" :: "a"(v), "v"(voff), "s"(srd) : "memory");
}
MF_DEV void buf_store(OutWord<4>::type v, mf_v4i srd, unsigned voff) {
    // (the dword form serves the odd fp32 cases; in the fused GPR emit at d = 5 the compiler carried the descriptor in VGPRs across the
    // step and handed them to the "s" operand as they were: made scalar again here)
    mf_v4i u;
    u.x = __builtin_amdgcn_readfirstlane(srd.x); u.y = __builtin_amdgcn_readfirstlane(srd.y);
    u.z = __builtin_amdgcn_readfirstlane(srd.z); u.w = __builtin_amdgcn_readfirstlane(srd.w);
    asm volatile("s_nop 4\n\tbuffer_store_dword %0, %1, %2, 0 offen\n\ts_nop 0" :: "a"(v), "v"(voff), "s"(u) : "memory");
}

// Geometry of the outputs of one step: a d x d row is handled as two halves of H0 and D - H0 matrix rows, a d row whole.
template <typename T, int D, int M, bool RSTEP, int BG = 1> struct PostLds {
    using Cfg = KfLdsCfg<T, D, M, RSTEP, BG>;
    static constexpr int S = (int)sizeof(T);
    static constexpr int H0 = (D + 1) / 2;
    static constexpr int B0 = H0 * D * S, B1 = (D - H0) * D * S, Bv = D * S;      // bytes of the two halves and of a vector row
    static constexpr int unit() { for (int u = 16; u > 4; u /= 2) if (B0 % u == 0 && B1 % u == 0 && Bv % u == 0) return u; return 4; }
    static constexpr int UNIT = unit();
    static constexpr int U0 = B0 / UNIT, U1 = B1 / UNIT, Uv = Bv / UNIT;          // store instructions per piece
    // The chain as the streamed backward of log_likelihood reads it (MODE 2 of post_lds_kernel, mf_grad_lds.hpp): ONE record per
    // transition, [chol(Q') lower triangle, row-major | b' | zeros up to a 16-B unit] - 224 B at d = 6 fp64 instead of rows of 288 + 48 B
    // in two tensors: two or three lines touched per step instead of four or five.  Staged and stored in two pieces.
    static constexpr int NG = D * (D + 1) / 2, NR = NG + D;                        // elements of the factor, of the record
    static constexpr int REC = ((NR * S + 15) / 16) * 16, RU = REC / 16;           // bytes, 16-B units per record
    static constexpr int RUa = (RU + 1) / 2, RUb = RU / 2, REa = RUa * 16 / S, REb = RUb * 16 / S;
    static constexpr int OFF_stageM = ((Cfg::LDS_TOTAL + 15) / 16) * 16;
    static constexpr int OFF_stagev = OFF_stageM + 64 * B0;
    static constexpr int STAGE = (64 * B0 + ((64 * Bv + 15) / 16) * 16) > 64 * RUa * 16 ? (64 * B0 + ((64 * Bv + 15) / 16) * 16) : 64 * RUa * 16;
    static constexpr int OFF_len = OFF_stageM + STAGE;
    static constexpr int OFF_relP = OFF_len + 256;                   // row offsets of the records
    static constexpr int TOTAL = OFF_relP + 256;                    // one wavefront's image + staging
};

// One staged piece: 64 rows x NU units in an LDS buffer; unit u of the image (u = 64 i + lane for store instruction i) is read
// back one instruction ahead of its store.
template <typename T, int NU, int UNIT> struct StagedPiece {
    using W = typename OutWord<UNIT>::type;
    static constexpr int EPU = UNIT / (int)sizeof(T);
    // validity of the row that instruction i moves, at position e (slow path: some chunk of the wave is not active yet)
    static MF_DEV unsigned guarded(const char* smem, int off_len, int lane, int i, long e, unsigned vo) {
        const int q0 = lane / NU, c0 = lane - q0 * NU;
        const int a = (64 * i) / NU, b = (64 * i) % NU;
        const int row = q0 + a + ((c0 + b >= NU) ? 1 : 0);
        const int len = *reinterpret_cast<const int*>(smem + off_len + row * 4);
        return e < (long)len ? vo : MF_DMA_INVALID;
    }
    // this lane's NU * EPU elements into the buffer, then the read of unit 0 is started
    static MF_DEV void stage(char* smem, int off_stage, int lane, const T* row, W& q0) {
        if constexpr (NU > 0) {
            T* dst = reinterpret_cast<T*>(smem + off_stage + lane * (NU * UNIT));
            MF_UNROLL for (int k = 0; k < NU * EPU; ++k) dst[k] = row[k];
            // (asm volatile with a memory clobber: the stores above are issued first, and the LDS executes one wave's operations in order)
            lds_read_a(q0, (unsigned)(size_t)smem + (unsigned)off_stage + (unsigned)lane * UNIT);
        }
    }
    // store unit I (already on its way into q[I & 1]) and start the read of unit I + 1
    template <int I> static MF_DEV void unit(const char* smem, int off_stage, int off_len, int lane, const unsigned (&vo)[NU > 0 ? NU : 1],
                                             mf_v4i srd, bool fast, long e, W& qa, W& qb) {
        if constexpr (I < NU) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned off = fast ? vo[I] : guarded(smem, off_len, lane, I, e, vo[I]);
            buf_store((I & 1) ? qb : qa, srd, off);
            if constexpr (I + 1 < NU)
                lds_read_a(((I + 1) & 1) ? qb : qa, (unsigned)(size_t)smem + (unsigned)off_stage + (unsigned)(64 * (I + 1) + lane) * UNIT);
        }
    }
};

// The emit step's sink (mf_post_math.hpp: post_emit_step) on the device.  It lives across the steps of the loop: the second half
// of A' of step j is stored during step j + 1.
// TRANS = false: the step hands over no transitions (post_emit_step<TRANS = false>), their store windows stay empty.
template <typename T, int D, int M, bool RSTEP, bool TRANS = true, int BG = 1> struct PostSink {
    using PL = PostLds<T, D, M, RSTEP, BG>;
    static constexpr int H0 = PL::H0, U0 = PL::U0, U1 = PL::U1, Uv = PL::Uv, UNIT = PL::UNIT;
    using W = typename OutWord<UNIT>::type;
    using P0 = StagedPiece<T, U0, UNIT>;
    using P1 = StagedPiece<T, U1, UNIT>;
    using Pv = StagedPiece<T, Uv, UNIT>;
    char* smem; int lane;
    DmaStream<OutPiece<U0, UNIT>> d0;              // row offsets per store instruction: first / second half of a matrix row, vector row
    DmaStream<OutPiece<(U1 > 0 ? U1 : 1), UNIT>> d1;
    DmaStream<OutPiece<Uv, UNIT>> dv;
    unsigned long long qA, qC, qb, fA, fC, fb;     // this position's rows of a_post, cholQ_post, b_post; the tensors' ends
    long e, minlen;
    bool have_prev;                                // the previous step left the second half of its A' in the staging buffer
    W ma, mb, va, vb;                              // read-back registers: matrix pieces, vector piece
    mf_v4i sM, sv;                                 // descriptors of the staged matrix piece / vector piece

    MF_DEV void init(char* smem_, int lane_, int rel_mat, int rel_vec) {
        smem = smem_; lane = lane_;
        d0.init(smem, lane, rel_mat, 0);
        if constexpr (U1 > 0) d1.init(smem, lane, rel_mat, 0);
        dv.init(smem, lane, rel_vec, 0);
        have_prev = false;
    }
    // windows of tick sites (post_emit_step's map) and the units a site stores: units [i U / W, (i + 1) U / W) at site S0 + i
    template <int SITE, int S0, int S1, int U, typename F> MF_DEV void window(F&& f) {
        if constexpr (SITE >= S0 && SITE < S1 && U > 0) {
            constexpr int i = SITE - S0, Wd = S1 - S0;
            static_for<(i * U) / Wd, ((i + 1) * U) / Wd>(f);
        }
    }
    template <int SITE> MF_DEV void tick(bool) {
        constexpr int C0a = 9, C0b = 10 + D, Cva = C0b, Cvb = 10 + 2 * D, C1a = Cvb, C1b = 10 + 4 * D, A0a = 34, A0b = 34 + (D - H0) + D + 1;
        // the second half of the PREVIOUS step's A' (position e + 1; its descriptor was built when it was staged)
        if (TRANS && have_prev)
            window<SITE, 0, 9, U1>([&](auto ic) {
                P1::template unit<decltype(ic)::value>(smem, PL::OFF_stageM, PL::OFF_len, lane, d1.vo, sM, e + 1 < minlen, e + 1, ma, mb);
            });
        window<SITE, C0a, C0b, U0>([&](auto ic) {
            P0::template unit<decltype(ic)::value>(smem, PL::OFF_stageM, PL::OFF_len, lane, d0.vo, sM, e < minlen, e, ma, mb);
        });
        window<SITE, Cva, Cvb, Uv>([&](auto ic) {
            Pv::template unit<decltype(ic)::value>(smem, PL::OFF_stagev, PL::OFF_len, lane, dv.vo, sv, e < minlen, e, va, vb);
        });
        window<SITE, C1a, C1b, U1>([&](auto ic) {
            P1::template unit<decltype(ic)::value>(smem, PL::OFF_stageM, PL::OFF_len, lane, d1.vo, sM, e < minlen, e, ma, mb);
        });
        if constexpr (TRANS)
            window<SITE, A0a, A0b, U0>([&](auto ic) {
                P0::template unit<decltype(ic)::value>(smem, PL::OFF_stageM, PL::OFF_len, lane, d0.vo, sM, e < minlen, e, ma, mb);
            });
    }
    MF_DEV void stage_factor(const T (&Gi)[D][D], const T (&mean)[D], bool) {
        T row[H0 * D];
        MF_UNROLL for (int i = 0; i < H0; ++i) MF_UNROLL for (int j = 0; j < D; ++j) row[i * D + j] = (j <= i) ? Gi[i][j] : T(0);
        sM = make_srd(qC, fC);
        P0::stage(smem, PL::OFF_stageM, lane, row, ma);
        sv = make_srd(qb, fb);
        Pv::stage(smem, PL::OFF_stagev, lane, mean, va);
    }
    MF_DEV void stage_factor_rest(const T (&Gi)[D][D], const T (&)[D], bool) {
        if constexpr (U1 > 0) {
            T row[(D - H0) * D];
            MF_UNROLL for (int i = H0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) row[(i - H0) * D + j] = (j <= i) ? Gi[i][j] : T(0);
            sM = make_srd(qC + PL::B0, fC);
            P1::stage(smem, PL::OFF_stageM, lane, row, ma);
        }
    }
    template <int HALF, int R> MF_DEV void stage_transition(const T (&Ap)[R][D], bool) {
        T row[R * D];
        MF_UNROLL for (int i = 0; i < R; ++i) MF_UNROLL for (int j = 0; j < D; ++j) row[i * D + j] = Ap[i][j];
        if constexpr (HALF == 0) {
            sM = make_srd(qA, fA);
            P0::stage(smem, PL::OFF_stageM, lane, row, ma);
        } else {
            sM = make_srd(qA + PL::B0, fA);
            P1::stage(smem, PL::OFF_stageM, lane, row, ma);
            have_prev = true;
        }
    }
    // after the last step: the second half of its A' is still staged (position `e` of that step)
    MF_DEV void flush() {
        if constexpr (U1 > 0) {
            if (have_prev)
                static_for<0, U1>([&](auto ic) {
                    P1::template unit<decltype(ic)::value>(smem, PL::OFF_stageM, PL::OFF_len, lane, d1.vo, sM, e < minlen, e, ma, mb);
                });
        }
        have_prev = false;
    }
};

// The sink of MODE 2: the packed records (PostLds: REC).  Piece A (the record's first RUa units) is staged when the factor
// exists and stored during the window of the factor's first half, piece B with the rest of the factor; no transitions.
// (PL: any layout with the record geometry RUa ... NR and the offsets OFF_stageM, OFF_len, OFF_relP - PostLds here, GprBwdLds in
// mf_gpr_grad.hpp)
template <typename T, int D, typename PL> struct PackedSinkT {
    static constexpr int RUa = PL::RUa, RUb = PL::RUb, REa = PL::REa, REb = PL::REb, NG = PL::NG, NR = PL::NR;
    using W = typename OutWord<16>::type;
    using PA = StagedPiece<T, RUa, 16>;
    using PB = StagedPiece<T, (RUb > 0 ? RUb : 1), 16>;
    char* smem; int lane;
    DmaStream<OutPiece<RUa, 16>> da;
    DmaStream<OutPiece<(RUb > 0 ? RUb : 1), 16>> db;
    unsigned long long qR, fR;                     // this position's records; the end of the record array
    unsigned long long qA, qC, qb, fA, fC, fb;     // (the members the kernel sets for either sink)
    long e, minlen;
    bool have_prev;
    W ma, mb;
    mf_v4i sM;

    MF_DEV void init(char* smem_, int lane_, int, int) {
        smem = smem_; lane = lane_;
        da.init(smem, lane, PL::OFF_relP, 0);
        if constexpr (RUb > 0) db.init(smem, lane, PL::OFF_relP, 0);
        have_prev = false;
    }
    template <int SITE, int S0, int S1, int U, typename F> MF_DEV void window(F&& f) {
        if constexpr (SITE >= S0 && SITE < S1 && U > 0) {
            constexpr int i = SITE - S0, Wd = S1 - S0;
            static_for<(i * U) / Wd, ((i + 1) * U) / Wd>(f);
        }
    }
    template <int SITE> MF_DEV void tick(bool) {
        constexpr int C0a = 9, C0b = 10 + 2 * D, C1a = C0b, C1b = 10 + 4 * D;
        window<SITE, C0a, C0b, RUa>([&](auto ic) {
            PA::template unit<decltype(ic)::value>(smem, PL::OFF_stageM, PL::OFF_len, lane, da.vo, sM, e < minlen, e, ma, mb);
        });
        if constexpr (RUb > 0)
            window<SITE, C1a, C1b, RUb>([&](auto ic) {
                PB::template unit<decltype(ic)::value>(smem, PL::OFF_stageM, PL::OFF_len, lane, db.vo, sM, e < minlen, e, ma, mb);
            });
    }
    // element q of the record
    static MF_DEV T element(int q, const T (&Gi)[D][D], const T (&mean)[D]) {
        T v = T(0);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            MF_UNROLL for (int j = 0; j <= i; ++j) if (q == i * (i + 1) / 2 + j) v = Gi[i][j];
            if (q == NG + i) v = mean[i];
        }
        return v;
    }
    MF_DEV void stage_factor(const T (&Gi)[D][D], const T (&mean)[D], bool) {
        T row[REa];
        MF_UNROLL for (int q = 0; q < REa; ++q) row[q] = element(q, Gi, mean);
        sM = make_srd(qR, fR);
        PA::stage(smem, PL::OFF_stageM, lane, row, ma);
    }
    MF_DEV void stage_factor_rest(const T (&Gi)[D][D], const T (&mean)[D], bool) {
        if constexpr (RUb > 0) {
            T row[REb];
            MF_UNROLL for (int q = 0; q < REb; ++q) row[q] = element(REa + q, Gi, mean);
            sM = make_srd(qR + RUa * 16, fR);
            PB::stage(smem, PL::OFF_stageM, lane, row, ma);
        }
    }
    template <int HALF, int R> MF_DEV void stage_transition(const T (&)[R][D], bool) {}
    MF_DEV void flush() {}
};

// rows of b and H per DMA batch in the passes of the streamed backward (three outputs: the image of one row is as much as fits)
constexpr int backward_row_group(int m) { return m <= 2 ? 2 : 1; }

// KfPump (mf_kf_lds.hpp) with the rows of b and H fetched Cfg::BGRP steps at a time: their batch goes out only on the steps
// that leave a group (bfetch), like the y rows.
template <typename Cfg> struct PostPump {
    const DmaStream<typename Cfg::StA>& dA; const DmaStream<typename Cfg::StC>& dC;
    const DmaStream<typename Cfg::Stb>& db; const DmaStream<typename Cfg::StH>& dH;
    const DmaStream<typename Cfg::Sty>& dy; const DmaStream<typename Cfg::StR>& dR;
    mf_v4i sA, sC, sb, sH, sy, sR;
    unsigned lds0;
    bool more;
    bool yfetch = true;
    bool bfetch = true;
    template <int K> MF_DEV void small() const {
        if (!more) return;
        constexpr int HC = (Cfg::StC::NI + 1) / 2;
        if (K == 0) dC.template issue<0, HC>(sC, lds0 + Cfg::OFF_C);
        else {
            dC.template issue<HC, 64>(sC, lds0 + Cfg::OFF_C);
            if (bfetch) {
                db.template issue<0, 64>(sb, lds0 + Cfg::OFF_b);
                dH.template issue<0, 64>(sH, lds0 + Cfg::OFF_H);
            }
            if (yfetch) dy.template issue<0, 64>(sy, lds0 + Cfg::OFF_y);
            if (Cfg::RS) dR.template issue<0, 64>(sR, lds0 + Cfg::OFF_R);
        }
    }
    template <int K> MF_DEV void all() const {
        dC.template issue<0, 64>(sC, lds0 + Cfg::OFF_C);
        db.template issue<0, 64>(sb, lds0 + Cfg::OFF_b);
        dH.template issue<0, 64>(sH, lds0 + Cfg::OFF_H);
        dy.template issue<0, 64>(sy, lds0 + Cfg::OFF_y);
        if (Cfg::RS) dR.template issue<0, 64>(sR, lds0 + Cfg::OFF_R);
        dA.template issue<0, 64>(sA, lds0 + Cfg::OFF_A);
    }
    template <int K> MF_DEV void big() const {
        if (!more) return;
        constexpr int Q = (Cfg::StA::NI + 3) / 4;
        dA.template issue<K * Q, (K + 1) * Q>(sA, lds0 + Cfg::OFF_A);
    }
    MF_DEV void unpumped() const {}
};

// Passes 1 and 3: one wavefront per workgroup = 64 (series, chunk) lanes.  KfArgs::P = chunks per series, L = transitions per
// chunk.  Position e of a chunk = transition tau0 + e; the wave walks e = nsteps-1 ... 0 (a chunk shorter than the wave's
// longest idles FIRST, so that all lanes end on their chunk's first transition and every DMA address is >= the tensor's start).
// MODE: 0 = pass 1, 1 = pass 3, 2 = pass 3 for the streamed backward: no transitions, chol(Q') and b' as packed records
// (PostLds::REC bytes per transition) at po.cq_post; po.a_post, po.b_post are not touched.
template <typename T, int D, int M, bool RSTEP, int MODE>
__global__ void __launch_bounds__(64) post_lds_kernel(KfArgs<T> a, long L, RedSys<T> out, PostOut<T> po) {
    constexpr bool EMIT = MODE != 0;
    constexpr int BG = (MODE == 2) ? backward_row_group(M) : 1;       // MODE 2 runs two wavefronts per CU: LDS for pairs of b and H rows
    using Cfg = KfLdsCfg<T, D, M, RSTEP, BG>;
    using PL = PostLds<T, D, M, RSTEP, BG>;
    using Sink = std::conditional_t<MODE == 2, PackedSinkT<T, D, PL>, PostSink<T, D, M, RSTEP, true, BG>>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const long total = a.B * a.P;
    const long id_raw = (long)blockIdx.x * 64 + lane;
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;
    const long s = id / a.P, c = id % a.P;
    const long nt = a.Tn - 1;                       // transitions per series
    const long tau0 = c * L;
    long len = nt - tau0;
    if (len > L) len = L;
    if (len < 0 || !valid) len = 0;
    constexpr int S = sizeof(T);

    // wave-uniform trip count: the longest chunk in this wave
    long nsteps = len;
    MF_UNROLL for (int off = 32; off > 0; off >>= 1) {
        const long o = __shfl_xor((long long)nsteps, off);
        nsteps = o > nsteps ? o : nsteps;
    }
    nsteps = __builtin_amdgcn_readfirstlane((int)nsteps);
    // shortest chunk of the wave (lanes without one count as 0): from position minlen - 1 down every row of the wave is stored
    long minlen = len;
    MF_UNROLL for (int off = 32; off > 0; off >>= 1) {
        const long o = __shfl_xor((long long)minlen, off);
        minlen = o < minlen ? o : minlen;
    }
    minlen = __builtin_amdgcn_readfirstlane((int)minlen);

    // ---- DMA set-up: per-row offsets into LDS tables, wave-uniform stream pointers ---------------------
    const unsigned long long offA = (unsigned long long)(s * nt + tau0) * (D * D * S);
    const unsigned long long offb = (unsigned long long)(s * nt + tau0) * (D * S);
    const unsigned long long offH = (unsigned long long)(s * a.Tn + tau0 + 1) * (M * D * S);
    const unsigned long long offy = (unsigned long long)(s * a.Tn + tau0 + 1) * (M * S);
    const unsigned long long offR = (unsigned long long)(s * a.Tn + tau0 + 1) * (M * M * S);
    const unsigned long long offA0 = uniform64(offA), offb0 = uniform64(offb);
    const unsigned long long offH0 = uniform64(offH), offy0 = uniform64(offy), offR0 = uniform64(offR);
    const bool rowok = valid && len > 0;
    {
        unsigned* tab = reinterpret_cast<unsigned*>(smem);
        tab[Cfg::OFF_relA / 4 + lane] = rowok ? (unsigned)(offA - offA0) : MF_DMA_INVALID;
        tab[Cfg::OFF_relb / 4 + lane] = rowok ? (unsigned)(offb - offb0) : MF_DMA_INVALID;
        tab[Cfg::OFF_relH / 4 + lane] = rowok ? (unsigned)(offH - offH0) : MF_DMA_INVALID;
        tab[Cfg::OFF_rely / 4 + lane] = rowok ? (unsigned)(offy - offy0) : MF_DMA_INVALID;
        tab[Cfg::OFF_relR / 4 + lane] = rowok ? (unsigned)(offR - offR0) : MF_DMA_INVALID;
        if (EMIT) reinterpret_cast<int*>(smem)[PL::OFF_len / 4 + lane] = rowok ? (int)len : 0;
        if (MODE == 2) tab[PL::OFF_relP / 4 + lane] = rowok ? (unsigned)((offA - offA0) / (D * D * S) * PL::REC) : MF_DMA_INVALID;
        if (lane < Cfg::StC::U) {
            unsigned g = 0;
            MF_UNROLL for (int cc = 0; cc < Cfg::StC::U; ++cc) if (lane == cc) g = (unsigned)Cfg::StC::global_unit(cc);
            tab[Cfg::OFF_gtabC / 4 + lane] = g * Cfg::StC::UNIT;
        }
    }
    DmaStream<typename Cfg::StA> dA;
    DmaStream<typename Cfg::StC> dC;
    DmaStream<typename Cfg::Stb> db;
    DmaStream<typename Cfg::StH> dH;
    DmaStream<typename Cfg::Sty> dy;
    DmaStream<typename Cfg::StR> dR;
    // stream pointers at the wave's LAST position; they walk downwards
    const long e_top = nsteps > 0 ? nsteps - 1 : 0;
    unsigned long long pA = (unsigned long long)a.A + offA0 + (unsigned long long)e_top * (D * D * S);
    unsigned long long pC = (unsigned long long)a.cholQ + offA0 + (unsigned long long)e_top * (D * D * S);
    unsigned long long pb = (unsigned long long)a.b + offb0 + (unsigned long long)(e_top / BG) * (BG * D * S);
    unsigned long long pH = (unsigned long long)a.H + offH0 + (unsigned long long)(e_top / BG) * (BG * M * D * S);
    unsigned long long py = (unsigned long long)a.y + offy0 + (unsigned long long)(e_top / Cfg::YG) * (Cfg::YG * M * S);
    unsigned long long pR = (unsigned long long)a.Rinv + (RSTEP ? offR0 + (unsigned long long)e_top * (M * M * S) : 0ull);
    const unsigned long long eA = (unsigned long long)a.A + (unsigned long long)a.B * nt * (D * D * S);
    const unsigned long long eC = (unsigned long long)a.cholQ + (unsigned long long)a.B * nt * (D * D * S);
    const unsigned long long eb = (unsigned long long)a.b + (unsigned long long)a.B * nt * (D * S);
    const unsigned long long eH = (unsigned long long)a.H + (unsigned long long)a.B * a.Tn * (M * D * S);
    const unsigned long long ey = (unsigned long long)a.y + (unsigned long long)a.B * a.Tn * (M * S);
    const unsigned long long eR = (unsigned long long)a.Rinv + (RSTEP ? (unsigned long long)a.B * a.Tn * (M * M * S) : 0ull);
    const unsigned lds0 = (unsigned)(size_t)smem;

    // ---- state of the recursion --------------------------------------------------------------------------
    Elim<T, D, true> E;            // EMIT uses Phi, t, bad only
    E.init();
    T Rsh[M * M];                  // observation precision: shared (kept in registers) or this step's
    MF_UNROLL for (int i = 0; i < M * M; ++i) Rsh[i] = RSTEP ? T(0) : a.Rinv[i];
    if (EMIT && valid && c + 1 < a.P) {          // restart from the chunk's right boundary (the last chunk starts from zero)
        load_lower<T, D>(po.bPsi + id * D * D, E.Phi);
        load_vec<T, D>(po.bpsi + id * D, E.t);
    }
    // the LDS tables must be visible to every lane before the first DMA address is formed (one wave: a wait on the LDS
    // counter is enough) and the plain loads above must be done before DMAs are counted
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (EMIT) {
        // hipcc's wait-count pass does not look into asm: for it the boundary loads above are still in flight, and since their
        // first use is INSIDE the loop (whose header merges the pre-header state back in on every iteration) it put a
        // `s_waitcnt vmcnt(0)` in front of four groups of instructions of every step - each one a full drain of the next step's
        // DMA prefetch (and of the output stores).  Touching the values here makes it place that wait once, before the loop.
        MF_UNROLL for (int i = 0; i < D; ++i) {
            asm volatile("" : "+v"(E.t[i]));
            MF_UNROLL for (int j = 0; j <= i; ++j) asm volatile("" : "+v"(E.Phi[i][j]));
        }
    }
    dA.init(smem, lane, Cfg::OFF_relA, 0);
    dC.init(smem, lane, Cfg::OFF_relA, Cfg::OFF_gtabC);
    db.init(smem, lane, Cfg::OFF_relb, 0);
    dH.init(smem, lane, Cfg::OFF_relH, 0);
    dy.init(smem, lane, Cfg::OFF_rely, 0);
    if (RSTEP) dR.init(smem, lane, Cfg::OFF_relR, 0);
    // output rows (EMIT): row offsets of the input streams of the same shape
    Sink sink;
    if (EMIT) sink.init(smem, lane, Cfg::OFF_relA, Cfg::OFF_relb);

    const RowReader<T, typename Cfg::StA> rA(smem, Cfg::OFF_A, lane);
    const RowReader<T, typename Cfg::StC> rC(smem, Cfg::OFF_C, lane);
    const RowReader<T, typename Cfg::Stb> rb(smem, Cfg::OFF_b, lane);
    const RowReader<T, typename Cfg::StH> rH(smem, Cfg::OFF_H, lane);
    const RowReader<T, typename Cfg::Sty> ry(smem, Cfg::OFF_y, lane);
    const RowReader<T, typename Cfg::StR> rR(smem, Cfg::OFF_R, lane);

    using Pump = PostPump<Cfg>;
    if (nsteps > 0) {   // prologue: fetch the last position
        Pump p0{dA, dC, db, dH, dy, dR, make_srd(pA, eA), make_srd(pC, eC), make_srd(pb, eb), make_srd(pH, eH),
                make_srd(py, ey), make_srd(pR, eR), lds0, true};
        p0.template all<0>();
    }
    // output rows (EMIT): wave-uniform pointers to the wave's rows at the current position, walking downwards like the inputs
    unsigned long long qA = (unsigned long long)po.a_post + offA0 + (unsigned long long)e_top * (D * D * S);
    unsigned long long qC = (unsigned long long)po.cq_post + offA0 + (unsigned long long)e_top * (D * D * S);
    unsigned long long qb = (unsigned long long)po.b_post + offb0 + (unsigned long long)e_top * (D * S);
    const unsigned long long fA = (unsigned long long)po.a_post + (unsigned long long)a.B * nt * (D * D * S);
    const unsigned long long fC = (unsigned long long)po.cq_post + (unsigned long long)a.B * nt * (D * D * S);
    const unsigned long long fb = (unsigned long long)po.b_post + (unsigned long long)a.B * nt * (D * S);
    if (EMIT) { sink.fA = fA; sink.fC = fC; sink.fb = fb; sink.minlen = minlen; sink.e = 0; }
    // MODE 2: po.cq_post is the record array [B, T-1, REC bytes]
    unsigned long long qR = (unsigned long long)po.cq_post + offA0 / (D * D * S) * PL::REC + (unsigned long long)e_top * PL::REC;
    if constexpr (MODE == 2) sink.fR = (unsigned long long)po.cq_post + (unsigned long long)a.B * nt * PL::REC;

#define MF_POST_LDS_STEP(FIRST)                                                                                       \
    {                                                                                                                 \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
        const long e = nsteps - 1 - j;                                                                                \
        const bool more = e > 0;                                                                                      \
        const bool yfetch = (e % Cfg::YG) == 0;        /* position e-1 lies in the previous group of y rows */        \
        const bool bfetch = (e % BG) == 0;             /* ... of b and H rows */                                      \
        pA -= D * D * S; pC -= D * D * S; if (RSTEP) pR -= M * M * S;                                                 \
        if (bfetch) { pb -= BG * D * S; pH -= BG * M * D * S; }                                                       \
        if (yfetch) py -= Cfg::YG * M * S;                                                                            \
        T C[D][D], mvec[D], hk[M * D], yk[M];                                                                         \
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int jj = 0; jj <= i; ++jj) C[i][jj] = rC.at(i * D + jj); \
        if constexpr (BG == 1) {                                                                                      \
            MF_UNROLL for (int i = 0; i < D; ++i) mvec[i] = rb.at(i);                                                 \
            MF_UNROLL for (int i = 0; i < M * D; ++i) hk[i] = rH.at(i);                                               \
        } else {                                                                                                      \
            const int gb = (int)(e % BG);                                                                             \
            MF_UNROLL for (int i = 0; i < D; ++i)                                                                     \
                mvec[i] = *reinterpret_cast<const T*>(rb.row + (gb * D + i) * (int)sizeof(T));                        \
            MF_UNROLL for (int i = 0; i < M * D; ++i)                                                                 \
                hk[i] = *reinterpret_cast<const T*>(rH.row + (gb * M * D + i) * (int)sizeof(T));                      \
        }                                                                                                             \
        MF_UNROLL for (int i = 0; i < M; ++i)                                                                         \
            yk[i] = *reinterpret_cast<const T*>(ry.row + ((int)(e % Cfg::YG) * M + i) * (int)sizeof(T));              \
        if (RSTEP) { MF_UNROLL for (int i = 0; i < M * M; ++i) Rsh[i] = rR.at(i); }                                    \
        T Bm[D][D];                                                                                                   \
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int jj = 0; jj < D; ++jj) Bm[i][jj] = rA.at(i * D + jj); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
        const Pump pump{dA, dC, db, dH, dy, dR, make_srd(pA, eA), make_srd(pC, eC), make_srd(pb, eb),                 \
                        make_srd(pH, eH), make_srd(py, ey), make_srd(pR, eR), lds0, more, yfetch, bfetch};            \
        const bool active = e < len;                                                                                  \
        if constexpr (EMIT) {                                                                                         \
            sink.qA = qA; sink.qC = qC; sink.qb = qb; sink.e = e;                                                     \
            if constexpr (MODE == 2) { sink.qR = qR; qR -= PL::REC; }                                                 \
            qA -= D * D * S; qC -= D * D * S; qb -= D * S;                                                            \
            post_emit_step<T, D, M, MODE == 1>(E.Phi, E.t, E.bad, C, mvec, hk, yk, Rsh, Bm, pump, sink, active);                 \
        } else {                                                                                                      \
            post_up_step<T, D, M, FIRST>(E, C, mvec, hk, yk, Rsh, Bm, pump, active, c + 1 < a.P);                     \
        }                                                                                                             \
    }
    long j = 0;
    if (nsteps > 0) MF_POST_LDS_STEP(true)
    for (j = 1; j < nsteps; ++j) MF_POST_LDS_STEP(false)
#undef MF_POST_LDS_STEP
    if (EMIT) sink.flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (EMIT) {
        if (valid && c == 0) {     // block 0: the prior closes the chain
            T C[D][D], mvec[D], hk[M * D], yk[M], mean[D], Gi[D][D];
            load_lower<T, D>(a.cholP0 + s * D * D, C);
            load_vec<T, D>(a.mu0 + s * D, mvec);
            MF_UNROLL for (int i = 0; i < M * D; ++i) hk[i] = a.H[(s * a.Tn) * M * D + i];
            MF_UNROLL for (int i = 0; i < M; ++i) yk[i] = a.y[(s * a.Tn) * M + i];
            if (RSTEP) { MF_UNROLL for (int i = 0; i < M * M; ++i) Rsh[i] = a.Rinv[(s * a.Tn) * M * M + i]; }
            post_emit_prior<T, D, M>(E.Phi, E.t, E.bad, C, mvec, hk, yk, Rsh, mean, Gi);
            store_vec<T, D>(po.mu0_post + s * D, mean);
            store_lower<T, D>(po.cp0_post + s * D * D, Gi);
        }
    } else if (valid) {
        store_chunk<T, D, true>(out, s * a.P + (a.P - 1 - c), E, T(0));     // mirrored: the scan runs from the last chunk
    }
    if (valid && E.bad && a.info) raise_info(a.info);
}

// ---- pass 2 ----------------------------------------------------------------------------------------------------------
// One wavefront per series.  Mirrored summary j stems from chunk P-1-j; the inclusive scan leaves in j the composition of
// summaries 0 .. j, whose (Dv, tv) is the state (Psi, psi) of the backward recursion at the block that separates chunk
// P-1-j from chunk P-2-j: it is written where the latter (the consumer) looks for it.  More than 64 chunks per series: a
// lane folds q = ceil(P / 64) consecutive summaries first and re-walks them after the scan.
template <typename T, int D> MF_DEV void post_summary_load(const RedSys<T>& in, long idx, PostSummary<T, D>& o) {
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) { o.Dv[i][j] = T(0); o.GU[i][j] = T(0); }
    load_lower<T, D>(in.Dv + idx * D * D, o.Dv);
    load_lower<T, D>(in.GU + idx * D * D, o.GU);
    load_mat<T, D, D>(in.F + idx * D * D, o.F);
    load_vec<T, D>(in.tv + idx * D, o.tv);
    load_vec<T, D>(in.gU + idx * D, o.gU);
}
template <typename T, int D> struct PostScanLds {
    static constexpr int NE = D * (D + 1) + D * D + 2 * D;      // Dv, GU (lower), F, tv, gU
    static constexpr int BYTES = NE * 64 * (int)sizeof(T);
    T* base; int lane;
    MF_DEV void put(const PostSummary<T, D>& o) const {
        int e = 0;
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) { base[(e++) * 64 + lane] = o.Dv[i][j]; }
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) { base[(e++) * 64 + lane] = o.GU[i][j]; }
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) { base[(e++) * 64 + lane] = o.F[i][j]; }
        MF_UNROLL for (int i = 0; i < D; ++i) { base[(e++) * 64 + lane] = o.tv[i]; }
        MF_UNROLL for (int i = 0; i < D; ++i) { base[(e++) * 64 + lane] = o.gU[i]; }
    }
    MF_DEV void get(int from, PostSummary<T, D>& o) const {
        int e = 0;
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) { o.Dv[i][j] = base[(e++) * 64 + from]; }
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j <= i; ++j) { o.GU[i][j] = base[(e++) * 64 + from]; }
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) { o.F[i][j] = base[(e++) * 64 + from]; }
        MF_UNROLL for (int i = 0; i < D; ++i) { o.tv[i] = base[(e++) * 64 + from]; }
        MF_UNROLL for (int i = 0; i < D; ++i) { o.gU[i] = base[(e++) * 64 + from]; }
    }
};

template <typename T, int D>
__global__ void __launch_bounds__(64) post_scan_kernel(RedSys<T> in, long B, T* __restrict__ bPsi, T* __restrict__ bpsi,
                                                       int* info) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const long s = blockIdx.x;
    const long P = in.n;
    const long q = (P + 63) / 64;
    const long j0 = lane * q;
    long j1 = j0 + q;
    if (j1 > P) j1 = P;
    const bool has = j0 < P;
    bool bad = false;
    const PostScanLds<T, D> lds{reinterpret_cast<T*>(smem), lane};
    auto emit = [&](long j, const PostSummary<T, D>& o) {
        if (j + 1 < P) {                                   // consumer: chunk P-2-j
            const long idx = s * P + (P - 2 - j);
            store_sym<T, D>(bPsi + idx * D * D, o.Dv);
            store_vec<T, D>(bpsi + idx * D, o.tv);
        }
    };
    PostSummary<T, D> acc;
    if (has) {
        post_summary_load<T, D>(in, s * P + j0, acc);
        for (long j = j0 + 1; j < j1; ++j) {
            PostSummary<T, D> nx;
            post_summary_load<T, D>(in, s * P + j, nx);
            post_combine<T, D>(acc, nx, bad);
            acc = nx;
        }
    }
    const int nl = (int)((P + q - 1) / q);                  // lanes that hold a run
    for (int off = 1; off < nl; off <<= 1) {
        if (has) lds.put(acc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave per workgroup: LDS operations execute in order
        __builtin_amdgcn_wave_barrier();
        if (has && lane >= off) {
            PostSummary<T, D> prev;
            lds.get(lane - off, prev);
            post_combine<T, D>(prev, acc, bad);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave per workgroup: LDS operations execute in order
        __builtin_amdgcn_wave_barrier();
    }
    if (q == 1) {
        if (has) emit(j0, acc);
    } else {
        // exclusive prefix of this lane's run = the scanned value of the lane before it; re-walk the run from there
        if (has) lds.put(acc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave per workgroup: LDS operations execute in order
        __builtin_amdgcn_wave_barrier();
        if (has) {
            PostSummary<T, D> run;
            if (lane > 0) lds.get(lane - 1, run);
            for (long j = j0; j < j1; ++j) {
                PostSummary<T, D> nx;
                post_summary_load<T, D>(in, s * P + j, nx);
                if (lane > 0 || j > j0) post_combine<T, D>(run, nx, bad);
                run = nx;
                emit(j, run);
            }
        }
    }
    if (bad && info) raise_info(info);
}

}  // namespace mf
