// K0, LDS-staged: the streaming form of kf_chunk_kernel for gfx950.
//
// Same math as kf_chunk_kernel (one lane = one (series, time-chunk) sub-problem, register-resident
// partitioned elimination), but the HBM side is re-designed around the memory system:
//
//   * every lane's next transition (A_k, cholQ_k, b_k, H_k, y_k) is brought in by LDS-DMA
//     (`buffer_load_dword[x4] ... lds`): no VGPR staging, and the load of step k+1 is in flight
//     during almost all of step k, so one wavefront per SIMD is enough to cover HBM latency;
//   * one DMA wave-instruction moves 64 x 16 B as contiguous pieces of consecutive rows
//     (lane -> (row, unit) = divmod(64 i + lane, units per row)), i.e. full 128-B lines of the
//     row-major [B, T, d, d] tensors, instead of 64 lanes touching 64 different lines;
//   * the LDS image is row-major with an ODD row stride (in 16-B units) so the per-lane row reads
//     (`ds_read_b128`, lane r reads row r) are bank-conflict free; the pad slot of every row is
//     filled by an out-of-range DMA lane (the buffer range check returns zeros);
//   * only the 16-B units of cholQ_k that hold lower-triangular entries are fetched into LDS.
//
// A workgroup is ONE wavefront (no barriers anywhere); LDS per workgroup is ~39 KB at d=6 fp64, so
// four workgroups (one per SIMD) share a CU's 160 KB.
//
// Chunk convention here: chunk c of a series owns transitions [c L, min((c+1) L, T-1)), transition t
// leads from block t to block t+1; chunk 0 additionally owns block 0 (the prior).  The separator a
// chunk leaves behind is the block its last transition leads to.
#pragma once
#include "mf_kernels.hpp"
#ifndef MF_PUMPMASK
#define MF_PUMPMASK 63   // bit k: DMA batch k of the next step is issued between arithmetic phases (else up front)
#endif

namespace mf {

typedef int mf_v4i __attribute__((ext_vector_type(4)));

// NOTE the leading `s_nop 4`: hipcc pads no hazards inside an asm string.  When the descriptor (or the LDS
// address) has just been produced by a VALU instruction - v_readfirstlane, or a v_readlane restoring a spilled
// SGPR - a VMEM instruction reading it needs 5 wait states; without them the DMA ran with a stale descriptor
// and silently fetched another stream's rows.
// Cache policy of the wide (A, cholQ) and narrow (b, H, y) streams: experiment knobs, "" = default policy.
#define MF_POLICY_STR_0 ""
#define MF_POLICY_STR_1 " nt"
#define MF_POLICY_STR_2 " sc1"
#define MF_POLICY_STR_3 " sc0 sc1"
#define MF_POLICY_STR_4 " sc0"
#define MF_POLICY_CAT(n) MF_POLICY_STR_##n
#define MF_POLICY_OF(n) MF_POLICY_CAT(n)
#ifndef MF_POLW
#define MF_POLW 1   // A, cholQ: nt - streamed past the L2 so that it keeps the partially used b, H, y lines (measured -6 %)
#endif
#ifndef MF_POLN
#define MF_POLN 0
#endif
#define MF_POLICY_WIDE MF_POLICY_OF(MF_POLW)
#define MF_POLICY_NARROW MF_POLICY_OF(MF_POLN)
// Up to four consecutive DMA instructions of one stream in ONE asm statement (round 4): the leading `s_nop 4`, the save /
// restore of m0 are paid once per group instead of once per instruction (6 -> 3.5 instructions per DMA, 56 DMAs per step at
// d = 6 fp64), and m0 walks from one 1-KB (dwordx4) or 256-B (dword) slice of the LDS image to the next by `s_add_u32` (SCC is
// declared clobbered; one wait state between a write of m0 and the LDS-DMA that reads it, as before).
#define MF_DMA_HEAD "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
#define MF_DMA_TAIL "s_mov_b32 m0, %0"
#define MF_DMA_X4(v, pol) "buffer_load_dwordx4 " v ", %1, 0 offen" pol " lds\n\t"
#define MF_DMA_X1(v, pol) "buffer_load_dword " v ", %1, 0 offen" pol " lds\n\t"
#define MF_DMA_STEP4 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
#define MF_DMA_STEP1 "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\t"
template <bool WIDE, int N> MF_DEV void dma_b128_group(mf_v4i srd, unsigned lds_addr, const unsigned* v) {
    unsigned keep;
    static_assert(N >= 1 && N <= 4, "group size");
#define MF_DMA_B128_BODY(POL)                                                                                                       \
    if constexpr (N == 1)                                                                                                           \
        asm volatile(MF_DMA_HEAD MF_DMA_X4("%3", POL) MF_DMA_TAIL : "=&s"(keep) : "s"(srd), "s"(lds_addr), "v"(v[0]) : "memory", "scc"); \
    else if constexpr (N == 2)                                                                                                      \
        asm volatile(MF_DMA_HEAD MF_DMA_X4("%3", POL) MF_DMA_STEP4 MF_DMA_X4("%4", POL) MF_DMA_TAIL                                 \
                     : "=&s"(keep) : "s"(srd), "s"(lds_addr), "v"(v[0]), "v"(v[1]) : "memory", "scc");                              \
    else if constexpr (N == 3)                                                                                                      \
        asm volatile(MF_DMA_HEAD MF_DMA_X4("%3", POL) MF_DMA_STEP4 MF_DMA_X4("%4", POL) MF_DMA_STEP4 MF_DMA_X4("%5", POL) MF_DMA_TAIL \
                     : "=&s"(keep) : "s"(srd), "s"(lds_addr), "v"(v[0]), "v"(v[1]), "v"(v[2]) : "memory", "scc");                   \
    else                                                                                                                            \
        asm volatile(MF_DMA_HEAD MF_DMA_X4("%3", POL) MF_DMA_STEP4 MF_DMA_X4("%4", POL) MF_DMA_STEP4 MF_DMA_X4("%5", POL)           \
                     MF_DMA_STEP4 MF_DMA_X4("%6", POL) MF_DMA_TAIL                                                                  \
                     : "=&s"(keep) : "s"(srd), "s"(lds_addr), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "memory", "scc");
    if (WIDE) { MF_DMA_B128_BODY(MF_POLICY_WIDE) } else { MF_DMA_B128_BODY(MF_POLICY_NARROW) }
#undef MF_DMA_B128_BODY
}
template <int N> MF_DEV void dma_b32_group(mf_v4i srd, unsigned lds_addr, const unsigned* v) {
    unsigned keep;
    static_assert(N >= 1 && N <= 4, "group size");
    if constexpr (N == 1)
        asm volatile(MF_DMA_HEAD MF_DMA_X1("%3", MF_POLICY_NARROW) MF_DMA_TAIL : "=&s"(keep) : "s"(srd), "s"(lds_addr), "v"(v[0]) : "memory", "scc");
    else if constexpr (N == 2)
        asm volatile(MF_DMA_HEAD MF_DMA_X1("%3", MF_POLICY_NARROW) MF_DMA_STEP1 MF_DMA_X1("%4", MF_POLICY_NARROW) MF_DMA_TAIL
                     : "=&s"(keep) : "s"(srd), "s"(lds_addr), "v"(v[0]), "v"(v[1]) : "memory", "scc");
    else if constexpr (N == 3)
        asm volatile(MF_DMA_HEAD MF_DMA_X1("%3", MF_POLICY_NARROW) MF_DMA_STEP1 MF_DMA_X1("%4", MF_POLICY_NARROW) MF_DMA_STEP1
                     MF_DMA_X1("%5", MF_POLICY_NARROW) MF_DMA_TAIL
                     : "=&s"(keep) : "s"(srd), "s"(lds_addr), "v"(v[0]), "v"(v[1]), "v"(v[2]) : "memory", "scc");
    else
        asm volatile(MF_DMA_HEAD MF_DMA_X1("%3", MF_POLICY_NARROW) MF_DMA_STEP1 MF_DMA_X1("%4", MF_POLICY_NARROW) MF_DMA_STEP1
                     MF_DMA_X1("%5", MF_POLICY_NARROW) MF_DMA_STEP1 MF_DMA_X1("%6", MF_POLICY_NARROW) MF_DMA_TAIL
                     : "=&s"(keep) : "s"(srd), "s"(lds_addr), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "memory", "scc");
}

// One streamed array: ROWB bytes per (row, step), of which only the units flagged by Keep are fetched.
// UNIT is the DMA granule (16 when ROWB is a multiple of 16, else 4).  The LDS image of a stream is
// row-major [64 rows][U units], filled by U wave-instructions: instruction i, lane l carries unit
// p = 64 i + l, i.e. (row, unit) = divmod(p, U) - consecutive lanes read consecutive 16-B pieces of a row.
// (Rows are NOT padded to an odd stride: the per-lane row reads then take a 2..4-way LDS bank conflict, but
// the kernel reads ~600 B per lane per step, nowhere near LDS bandwidth, while padding would push the
// fp64 d=6 image over a quarter of the CU's 160 KB and cost a wave of occupancy.)
template <int ROWB, typename Keep> struct Stream {
    // 16-B granules whenever a row holds at least one; a row that is not a whole number of them (odd d) is fetched up
    // to the next 16-B boundary: the extra 4..12 bytes belong to the next row (or lie past the end of the tensor, where
    // the buffer range check returns zeros) and are never read.  Rows then start 4- or 8-byte aligned, which the DMA
    // (dword-aligned dwordx4) accepts.
    static constexpr int UNIT = (ROWB >= 16) ? 16 : 4;
    static constexpr int UG = (ROWB + UNIT - 1) / UNIT;          // units per row in global memory
    static constexpr int count_kept() { int n = 0; for (int u = 0; u < UG; ++u) n += Keep::keep(u, UNIT) ? 1 : 0; return n; }
    static constexpr int U = count_kept();                        // units per row kept in LDS
    static constexpr int NI = U;                                  // DMA wave-instructions per step
    static constexpr int LDS_BYTES = 64 * U * UNIT;
    static constexpr bool ALL = (U == UG);
    // global unit index of the c-th kept unit
    static constexpr int global_unit(int c) {
        int n = 0;
        for (int u = 0; u < UG; ++u) if (Keep::keep(u, UNIT)) { if (n == c) return u; ++n; }
        return -1;
    }
    // compact index of global unit u (must be kept)
    static constexpr int compact_unit(int gu) {
        int n = 0;
        for (int u = 0; u < gu; ++u) n += Keep::keep(u, UNIT) ? 1 : 0;
        return n;
    }
};

struct KeepAll { static constexpr bool keep(int, int) { return true; } };
// keep the units of a row-major D x D matrix (element size S) that contain an entry of the lower triangle
template <int D, int S> struct KeepLower {
    static constexpr bool keep(int u, int unit) {
        const int lo = u * unit, hi = lo + unit;                 // byte range of the unit
        for (int i = 0; i < D; ++i) {
            const int a = (i * D) * S, b = (i * D + i + 1) * S;   // bytes of row i's lower part
            if (a < hi && lo < b) return true;
        }
        return false;
    }
};

constexpr unsigned MF_DMA_INVALID = 0xF0000000u;   // row offset of an invalid row: lands out of range
constexpr unsigned long long MF_DMA_MAXREC = 0xE0000000ull;

// DMA source addressing of one stream.  The per-row byte offsets (relative to the wave's descriptor
// base) live in a 64-entry LDS table; each lane derives its (row, unit) for instruction i from two
// per-lane constants and compile-time (64 i) / U, (64 i) % U - a few VALU ops, no per-instruction
// registers (a register table per instruction was spilled to scratch, and every scratch reload waits
// `vmcnt(0)`, i.e. for every DMA in flight).
template <typename St> struct DmaStream {
    unsigned vo[St::NI];           // this lane's source byte offset for DMA instruction i (registers / AGPRs)
    // rel_tab: LDS byte address of this stream's row-offset table; gtab: LDS byte address of the
    // compact-unit -> global byte offset table (only read when the stream drops units)
    MF_DEV void init(const char* smem, int lane, int rel_tab, int gtab) {
        const int q0 = lane / St::U, c0 = lane - q0 * St::U;
        MF_UNROLL for (int i = 0; i < St::NI; ++i) {
            const int a = (64 * i) / St::U, b = (64 * i) % St::U;
            int cu = c0 + b;
            const int carry = cu >= St::U ? 1 : 0;
            cu -= carry * St::U;
            const int row = q0 + a + carry;
            const unsigned rel = *reinterpret_cast<const unsigned*>(smem + rel_tab + row * 4);
            unsigned off;
            if (St::ALL) off = (unsigned)cu * St::UNIT;
            else off = *reinterpret_cast<const unsigned*>(smem + gtab + cu * 4);
            vo[i] = rel + off;
#ifdef MF_EXPERIMENT
            // ablation "what would a line ring buy" (MF_KF_DEBUG bit 3): bits 24..28 carry the unit's index in its row, bits
            // 29..30 the 32-byte phase of the row's first step (valid offsets stay below 2^24)
            if (pack_base != 0xffffffffu && rel < MF_DMA_INVALID)
                vo[i] |= ((((pack_base + rel) >> 5) & 3u) << 29) | ((unsigned)(off / St::UNIT) << 24);
#endif
        }
    }
#ifdef MF_EXPERIMENT
    unsigned pack_base = 0xffffffffu;     // low 32 bits of the stream's base address when the ablation is on
    unsigned step_next = 0;               // index (within the chunk) of the step whose rows are being fetched
    bool skip_carried = false;
    // the offset actually issued: a unit that lies in the row's first 128-B line, when that line is shared with the previous
    // row (row start not on a line boundary), is NOT fetched (results are garbage - timing / traffic experiment only)
    MF_DEV unsigned exp_offset(unsigned v) const {
        if (pack_base == 0xffffffffu || v >= MF_DMA_INVALID) return v;
        const unsigned phase = ((v >> 29) + step_next) & 3u, gu = (v >> 24) & 31u, off = v & 0xffffffu;
        const bool carried = skip_carried && phase != 0u && gu < 8u - 2u * phase;
        return carried ? MF_DMA_INVALID : off;
    }
#endif
    // issue DMA instructions [i0, i1) of this stream, up to four per asm statement
    template <int I0, int I1> MF_DEV void issue(mf_v4i srd, unsigned lds_base) const {
        constexpr int HI = I1 < St::NI ? I1 : St::NI;
        if constexpr (I0 < HI) {
            constexpr int N = (HI - I0) < 4 ? (HI - I0) : 4;
            unsigned voff[N];
            MF_UNROLL for (int k = 0; k < N; ++k) {
#ifdef MF_EXPERIMENT
                voff[k] = exp_offset(vo[I0 + k]);
#else
                voff[k] = vo[I0 + k];
#endif
            }
            if (St::UNIT == 16) dma_b128_group<(St::UG >= 8), N>(srd, lds_base + I0 * 1024, voff);
            else dma_b32_group<N>(srd, lds_base + I0 * 256, voff);
            issue<I0 + N, I1>(srd, lds_base);
        }
    }
};

// value of lane 0 as a wave-uniform (SGPR) 64-bit quantity
MF_DEV unsigned long long uniform64(unsigned long long x) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// wave-uniform buffer descriptor over [base, end)
MF_DEV mf_v4i make_srd(unsigned long long base, unsigned long long end, int debug = 0) {
    unsigned long long rem = end > base ? end - base : 0ull;
    if (debug & 1) rem = 0;
    if (rem > MF_DMA_MAXREC) rem = MF_DMA_MAXREC;
    mf_v4i srd;
    srd.x = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    srd.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu));
    srd.z = __builtin_amdgcn_readfirstlane((int)(unsigned)rem);
    srd.w = 0x00020000;
    return srd;
}

// read this lane's row from an LDS stream image
template <typename T, typename St> struct RowReader {
    const char* row;
    MF_DEV RowReader(const char* smem, int lds_off, int lane) : row(smem + lds_off + lane * (St::U * St::UNIT)) {}
    // element e of the ORIGINAL row (must lie in a kept unit)
    MF_DEV T at(int e) const {
        const int byte = e * (int)sizeof(T);
        const int gu = byte / St::UNIT;
        const int off = St::compact_unit(gu) * St::UNIT + (byte - gu * St::UNIT);
        return *reinterpret_cast<const T*>(row + off);
    }
};

// RSTEP: the observation precision is a per-step stream [B, T, M, M] (KalmanFilterWithSites / WithSparseSites) instead of
// one shared [M, M] matrix held in registers.
// BG: rows of b and H fetched per DMA batch (the same remedy as YG below, for kernels that have the LDS: the passes of the streamed
// backward, mf_post_lds.hpp MODE 2 and mf_grad_lds.hpp, run two wavefronts per CU and take pairs - a 48-B row touches 1.375 lines,
// a 96-B pair 1.5).
template <typename T, int D, int M, bool RSTEP = false, int BG = 1> struct KfLdsCfg {
    static constexpr int S = sizeof(T);
    static constexpr int BGRP = BG;
    using StA = Stream<D * D * S, KeepAll>;
    using StC = Stream<D * D * S, KeepLower<D, S>>;
    using Stb = Stream<BG * D * S, KeepAll>;
    using StH = Stream<BG * M * D * S, KeepAll>;
    // y rows are tiny (M S bytes): with one output they are fetched YG steps at a time, so a 128-B line of y is
    // touched every YG-th step instead of every step (the memory side moves whole lines whatever part is used)
    static constexpr int YG = (M == 1) ? 4 : 1;
    using Sty = Stream<YG * M * S, KeepAll>;
    using StR = Stream<M * M * S, KeepAll>;
    static constexpr bool RS = RSTEP;
    static constexpr int OFF_A = 0;
    static constexpr int OFF_C = OFF_A + StA::LDS_BYTES;
    static constexpr int OFF_b = OFF_C + StC::LDS_BYTES;
    static constexpr int OFF_H = OFF_b + Stb::LDS_BYTES;
    static constexpr int OFF_y = OFF_H + StH::LDS_BYTES;
    static constexpr int OFF_R = OFF_y + ((Sty::LDS_BYTES + 15) / 16) * 16;
    static constexpr int OFF_relA = OFF_R + (RSTEP ? ((StR::LDS_BYTES + 15) / 16) * 16 : 0);   // row offsets of A and cholQ
    static constexpr int OFF_relb = OFF_relA + 256;
    static constexpr int OFF_relH = OFF_relb + 256;
    static constexpr int OFF_rely = OFF_relH + 256;
    static constexpr int OFF_relR = OFF_rely + 256;
    static constexpr int OFF_gtabC = OFF_relR + 256;
    static constexpr int LDS_TOTAL = OFF_gtabC + ((StC::U * 4 + 15) / 16) * 16;
    // the streaming kernel is instantiated only where matrix rows are whole 16-B units, the per-step DMA count
    // fits the 6-bit vm counter and the image fits 64 KB of LDS
    static constexpr bool SUPPORTED =
                                      (StA::NI + StC::NI + Stb::NI + StH::NI + Sty::NI + (RSTEP ? StR::NI : 0)) < 64 &&
                                      LDS_TOTAL <= 64 * 1024;
};

// DMA batches of one step.  All of a step's data is pulled into registers at the top of the step behind ONE
// `s_waitcnt vmcnt(0)`; the batches of the next step are then issued between the arithmetic phases.
// (Counted waits - "all but the youngest n have landed" - were tried and are NOT safe here: with dword and
// dwordx4 LDS-DMA mixed in one stream of requests the landing order did not follow the issue order.)
template <typename Cfg> struct KfPump {
    static constexpr int N_SMALL = Cfg::StC::NI + Cfg::Stb::NI + Cfg::StH::NI + Cfg::Sty::NI + (Cfg::RS ? Cfg::StR::NI : 0);
    static constexpr int N_BIG = Cfg::StA::NI;
    const DmaStream<typename Cfg::StA>& dA; const DmaStream<typename Cfg::StC>& dC;
    const DmaStream<typename Cfg::Stb>& db; const DmaStream<typename Cfg::StH>& dH;
    const DmaStream<typename Cfg::Sty>& dy; const DmaStream<typename Cfg::StR>& dR;
    mf_v4i sA, sC, sb, sH, sy, sR;
    unsigned lds0;
    bool more;
    bool yfetch = true;     // this step's batch includes the next group of y rows
    template <int K> MF_DEV void small() const {
        if (!((MF_PUMPMASK >> K) & 1)) return;
        small_do<K>();
    }
    template <int K> MF_DEV void small_do() const {
        if (!more) return;
        constexpr int HC = (Cfg::StC::NI + 1) / 2;
        if (K == 0) dC.template issue<0, HC>(sC, lds0 + Cfg::OFF_C);
        else {
            dC.template issue<HC, 64>(sC, lds0 + Cfg::OFF_C);
            db.template issue<0, 64>(sb, lds0 + Cfg::OFF_b);
            dH.template issue<0, 64>(sH, lds0 + Cfg::OFF_H);
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
        if (!((MF_PUMPMASK >> (K + 2)) & 1)) return;
        big_do<K>();
    }
    // the batches that are NOT pumped between the arithmetic phases are issued up front
    MF_DEV void unpumped() const {
        if (!((MF_PUMPMASK >> 0) & 1)) small_do<0>();
        if (!((MF_PUMPMASK >> 1) & 1)) small_do<1>();
        if (!((MF_PUMPMASK >> 2) & 1)) big_do<0>();
        if (!((MF_PUMPMASK >> 3) & 1)) big_do<1>();
        if (!((MF_PUMPMASK >> 4) & 1)) big_do<2>();
        if (!((MF_PUMPMASK >> 5) & 1)) big_do<3>();
    }
    template <int K> MF_DEV void big_do() const {
        if (!more) return;
        constexpr int Q = (Cfg::StA::NI + 3) / 4;
        dA.template issue<K * Q, (K + 1) * Q>(sA, lds0 + Cfg::OFF_A);
    }
};

// One transition of the chain for this lane.  FIRST_SEP: the block on the left is this chunk's separator
// (it is not eliminated here; its coupling seeds the spike).  Register discipline: the only large live
// temporaries are Ci (lower), and ONE D x D array that is A -> B = C^-1 A -> Y = B L^-T -> W in turn;
// Q_k^-1 + H^T R^-1 H is formed after the elimination, when B is no longer needed.
// `pump` issues the LDS-DMA of the NEXT step in small batches between the arithmetic phases: a burst of
// ~40 wave-instructions back to back fills the CU's address queue and stalls the (only) wave of the SIMD
// at issue, while one batch every few hundred VALU instructions is absorbed for free.
template <typename T, int D, int M, bool SPIKE, bool FIRST, typename Pump>
MF_DEV void kf_lds_step(Elim<T, D, SPIKE>& E, LogAcc<T>& laC, T& acc_yry, T& acc_ww, const T (&C)[D][D],
                        const T (&mvec)[D], const T (&hk)[M * D], const T (&yk)[M], const T (&Rsh)[M * M],
                        T (&Bm)[D][D], const Pump& pump, bool active, bool sep_lane) {
    // FIRST (step 0 of a chunk): lanes of chunks c > 0 take the separator form, lanes of chunk 0 the ordinary
    // one - both inside the SAME call, because the DMA batches must be issued once, by all lanes, in order.
    const bool sep = FIRST && sep_lane, ord = active && !sep;
    // Every arithmetic phase is predicated on `active` (a chunk shorter than the wave's trip count idles),
    // the DMA batches in between are issued by ALL lanes: an LDS-DMA lane carries another row's data.
    T Ci[D][D], w[D], btw[D];
    if (active) {
        tri_inv_lower<T, D>(C, Ci, laC, E.bad);
        laC.renorm();
    }
    pump.template small<0>();
    if (active) {
        trimul_lower_vec<T, D>(Ci, mvec, w);
        acc_ww += dot_self<T, D>(w);
    }
    pump.template small<1>();
    pump.template big<0>();
    if (active) {
        trimul_lower_inplace<T, D, D>(Ci, Bm);            // B = C^-1 A
        gemv_t<T, D, D>(Bm, w, btw);
    }
    pump.template big<1>();
    if (FIRST && active && sep) {
        syrk_tn_lower<T, D, D>(Bm, E.GU, T(1));
        MF_UNROLL for (int i = 0; i < D; ++i) E.gU[i] = -btw[i];
    }
    if (ord) {
        syrk_tn_lower<T, D, D>(Bm, E.Phi, T(1));
        MF_UNROLL for (int i = 0; i < D; ++i) E.t[i] -= btw[i];
    }
    pump.template big<2>();
    if (FIRST && active && sep) {
        neg_trimulT_lower_inplace<T, D, D>(Ci, Bm);       // S = -C^-T B
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) E.X[i][j] = Bm[i][j];
    }
    if (ord) E.eliminate_main();
    pump.template big<3>();
    if (FIRST && active && sep) {
        trimulT_self_lower<T, D>(Ci, E.Phi);
        trimulT_lower_vec<T, D>(Ci, w, E.t);
        acc_yry += Obs<T, D, M>::apply(hk, yk, Rsh, M, E.Phi, E.t);
    }
    if (ord) {
        E.eliminate_spike();
        trsm_right_lower_t<T, D, D>(E.Phi, E.Li, Bm);     // Y = B L^-T
        neg_trimulT_lower_inplace<T, D, D>(Ci, Bm);       // W = -C^-T Y
        T Dn[D][D], rn[D];
        trimulT_self_lower<T, D>(Ci, Dn);
        trimulT_lower_vec<T, D>(Ci, w, rn);
        acc_yry += Obs<T, D, M>::apply(hk, yk, Rsh, M, Dn, rn);
        E.advance(Bm, Dn, rn);
    }
}

// KfArgs::P = chunks per series, L = transitions per chunk.
template <typename T, int D, int M, bool SPIKE, bool RSTEP = false>
__global__ void __launch_bounds__(64) kf_chunk_lds_kernel(KfArgs<T> a, long L, RedSys<T> out) {
    using Cfg = KfLdsCfg<T, D, M, RSTEP>;
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
    unsigned long long pA = (unsigned long long)a.A + offA0, pC = (unsigned long long)a.cholQ + offA0;
    unsigned long long pb = (unsigned long long)a.b + offb0, pH = (unsigned long long)a.H + offH0;
    unsigned long long py = (unsigned long long)a.y + offy0;
    unsigned long long pR = (unsigned long long)a.Rinv + (RSTEP ? offR0 : 0ull);
    const unsigned long long eA = (unsigned long long)a.A + (unsigned long long)a.B * nt * (D * D * S);
    const unsigned long long eC = (unsigned long long)a.cholQ + (unsigned long long)a.B * nt * (D * D * S);
    const unsigned long long eb = (unsigned long long)a.b + (unsigned long long)a.B * nt * (D * S);
    const unsigned long long eH = (unsigned long long)a.H + (unsigned long long)a.B * a.Tn * (M * D * S);
    const unsigned long long ey = (unsigned long long)a.y + (unsigned long long)a.B * a.Tn * (M * S);
    const unsigned long long eR = (unsigned long long)a.Rinv + (RSTEP ? (unsigned long long)a.B * a.Tn * (M * M * S) : 0ull);
    const unsigned lds0 = (unsigned)(size_t)smem;

    // ---- block 0 of chunk 0: the prior (plain loads; no DMA in flight yet) ------------------------------
    Elim<T, D, SPIKE> E;
    E.init();
    LogAcc<T> laC;
    laC.init();
    T acc_yry = T(0), acc_ww = T(0);
    T Rsh[M * M];                                  // observation precision: shared (kept in registers) or this step's
    MF_UNROLL for (int i = 0; i < M * M; ++i) Rsh[i] = RSTEP ? (valid ? a.Rinv[(s * a.Tn) * M * M + i] : T(0)) : a.Rinv[i];
    if (valid && c == 0) {
        T C[D][D], Ci[D][D], mvec[D], w[D];
        load_lower<T, D>(a.cholP0 + s * D * D, C);
        load_vec<T, D>(a.mu0 + s * D, mvec);
        tri_inv_lower<T, D>(C, Ci, laC, E.bad);
        laC.renorm();
        trimul_lower_vec<T, D>(Ci, mvec, w);
        acc_ww += dot_self<T, D>(w);
        trimulT_self_lower<T, D>(Ci, E.Phi);
        trimulT_lower_vec<T, D>(Ci, w, E.t);
        acc_yry += Obs<T, D, M>::apply(a.H + (s * a.Tn) * M * D, a.y + (s * a.Tn) * M, Rsh, M, E.Phi, E.t);
    }

    // wave-uniform trip count: the longest chunk in this wave
    long nsteps = len;
    MF_UNROLL for (int off = 32; off > 0; off >>= 1) {
        const long o = __shfl_xor((long long)nsteps, off);
        nsteps = o > nsteps ? o : nsteps;
    }
    nsteps = __builtin_amdgcn_readfirstlane((int)nsteps);
    // the LDS tables must be visible to every lane before the first DMA address is formed (one wave: a
    // wait on the LDS counter is enough) and the plain loads above must be done before DMAs are counted
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef MF_EXPERIMENT
    if (a.debug & 8) {
        dA.pack_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pA);
        dC.pack_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pC);
    }
#endif
    dA.init(smem, lane, Cfg::OFF_relA, 0);
    dC.init(smem, lane, Cfg::OFF_relA, Cfg::OFF_gtabC);
    db.init(smem, lane, Cfg::OFF_relb, 0);
    dH.init(smem, lane, Cfg::OFF_relH, 0);
    dy.init(smem, lane, Cfg::OFF_rely, 0);
    if (RSTEP) dR.init(smem, lane, Cfg::OFF_relR, 0);

    const RowReader<T, typename Cfg::StA> rA(smem, Cfg::OFF_A, lane);
    const RowReader<T, typename Cfg::StC> rC(smem, Cfg::OFF_C, lane);
    const RowReader<T, typename Cfg::Stb> rb(smem, Cfg::OFF_b, lane);
    const RowReader<T, typename Cfg::StH> rH(smem, Cfg::OFF_H, lane);
    const RowReader<T, typename Cfg::Sty> ry(smem, Cfg::OFF_y, lane);
    const RowReader<T, typename Cfg::StR> rR(smem, Cfg::OFF_R, lane);

    using Pump = KfPump<Cfg>;

    if (nsteps > 0) {   // prologue: fetch step 0
        Pump p0{dA, dC, db, dH, dy, dR, make_srd(pA, eA, a.debug), make_srd(pC, eC, a.debug | ((a.debug >> 2) & 1)),
                make_srd(pb, eb, a.debug | ((a.debug >> 1) & 1)), make_srd(pH, eH, a.debug | ((a.debug >> 1) & 1)),
                make_srd(py, ey, a.debug | ((a.debug >> 1) & 1)), make_srd(pR, eR, a.debug | ((a.debug >> 1) & 1)), lds0, true};
        p0.template all<0>();
    }

    // One streamed step: wait for this step's cholQ/b/H/y (A may still be in flight), move them to registers,
    // then run the arithmetic with the next step's DMA pumped in between.  FIRST distinguishes step 0 (whose
    // left block is the separator for every chunk but the first) so that the steady-state loop body holds a
    // single code path.
#ifdef MF_STAMP
#define MF_STAMP_AT(var) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); var = t_; }
    unsigned long long st_wait = 0, st_io = 0, st_comp = 0, ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
#define MF_STAMP_ACC { asm volatile("" :: "v"(E.quad), "v"(E.Phi[0][0]) : "memory"); MF_STAMP_AT(ts3) st_wait += ts1 - ts0; st_io += ts2 - ts1; st_comp += ts3 - ts2; }
#else
#define MF_STAMP_AT(var)
#define MF_STAMP_ACC
#endif
#ifdef MF_CHECKSUM
    T cs_A = 0, cs_C = 0, cs_b = 0, cs_H = 0, cs_y = 0;
#define MF_CHECKSUM_ACC if (j < len) { MF_UNROLL for (int i = 0; i < D; ++i) { cs_b += mvec[i]; MF_UNROLL for (int jj = 0; jj < D; ++jj) { cs_A += Bm[i][jj] * T(1 + i * D + jj); if (jj <= i) cs_C += C[i][jj] * T(1 + i * D + jj); } } \
        MF_UNROLL for (int i = 0; i < M * D; ++i) cs_H += hk[i] * T(1 + i); MF_UNROLL for (int i = 0; i < M; ++i) cs_y += yk[i]; }
#else
#define MF_CHECKSUM_ACC
#endif
#ifdef MF_EXPERIMENT
#define MF_EXP_STEP_NEXT { dA.step_next = dC.step_next = (unsigned)(j + 1); dA.skip_carried = dC.skip_carried = (a.debug & 8) != 0; }
#else
#define MF_EXP_STEP_NEXT
#endif
#define MF_NOPUMP_ISSUE pump.unpumped();
#define MF_KF_LDS_STEP(FIRST)                                                                                         \
    {                                                                                                                 \
        MF_STAMP_AT(ts0)                                                                                              \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
        MF_STAMP_AT(ts1)                                                                                              \
        const bool more = (j + 1 < nsteps);                                                                           \
        const bool yfetch = ((j + 1) % Cfg::YG) == 0;                                                                 \
        pA += D * D * S; pC += D * D * S; pb += D * S; pH += M * D * S; if (RSTEP) pR += M * M * S;                   \
        if (yfetch) py += Cfg::YG * M * S;                                                                            \
        T C[D][D], mvec[D], hk[M * D], yk[M];                                                                         \
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int jj = 0; jj <= i; ++jj) C[i][jj] = rC.at(i * D + jj); \
        MF_UNROLL for (int i = 0; i < D; ++i) mvec[i] = rb.at(i);                                                     \
        MF_UNROLL for (int i = 0; i < M * D; ++i) hk[i] = rH.at(i);                                                   \
        MF_UNROLL for (int i = 0; i < M; ++i)                                                                         \
            yk[i] = *reinterpret_cast<const T*>(ry.row + ((int)(j % Cfg::YG) * M + i) * (int)sizeof(T));              \
        if (RSTEP) { MF_UNROLL for (int i = 0; i < M * M; ++i) Rsh[i] = rR.at(i); }                                    \
        T Bm[D][D];                                                                                                   \
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int jj = 0; jj < D; ++jj) Bm[i][jj] = rA.at(i * D + jj); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
        MF_CHECKSUM_ACC                                                                                               \
        MF_STAMP_AT(ts2)                                                                                              \
        const Pump pump{dA, dC, db, dH, dy, dR, make_srd(pA, eA, a.debug),                                            \
                        make_srd(pC, eC, a.debug | ((a.debug >> 2) & 1)),                                             \
                        make_srd(pb, eb, a.debug | ((a.debug >> 1) & 1)),                                             \
                        make_srd(pH, eH, a.debug | ((a.debug >> 1) & 1)),                                             \
                        make_srd(py, ey, a.debug | ((a.debug >> 1) & 1)),                                             \
                        make_srd(pR, eR, a.debug | ((a.debug >> 1) & 1)), lds0, more, yfetch};                        \
        MF_EXP_STEP_NEXT                                                                                              \
        MF_NOPUMP_ISSUE                                                                                               \
        const bool active = j < len;                                                                                  \
        kf_lds_step<T, D, M, SPIKE, FIRST>(E, laC, acc_yry, acc_ww, C, mvec, hk, yk, Rsh, Bm, pump, active, c > 0);    \
        if (E.bad && first_bad < 0) first_bad = tau0 + j;                                                             \
        MF_STAMP_ACC                                                                                                  \
    }
    long j = 0;
    long first_bad = -1;          // the block whose elimination step first met a non-positive pivot
    if (nsteps > 0) MF_KF_LDS_STEP(true)
    for (j = 1; j < nsteps; ++j) MF_KF_LDS_STEP(false)
#undef MF_KF_LDS_STEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (valid) {
        const T scalar = T(-0.5) * (acc_yry + acc_ww) + T(0.5) * E.quad - laC.value() - E.laL.value();
        store_chunk<T, D, SPIKE>(out, id, E, scalar);
        if (E.bad && a.info) raise_pivot(a.info, s * a.Tn + (first_bad < 0 ? tau0 : first_bad));
#ifdef MF_CHECKSUM
        out.GU[id * D * D + 0] = cs_A; out.GU[id * D * D + 1] = cs_C; out.GU[id * D * D + 2] = cs_b;
        out.gU[id * D + 0] = cs_H; out.gU[id * D + 1] = cs_y;
#endif
#ifdef MF_STAMP
        if (lane == 0) {   // diagnostic build only: overwrite this chunk's GU block with the stamp sums
            out.GU[id * D * D + 0] = (T)st_wait; out.GU[id * D * D + 1] = (T)st_io; out.GU[id * D * D + 2] = (T)st_comp;
        }
#endif
    }
}

}  // namespace mf
