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

namespace mf {

typedef int mf_v4i __attribute__((ext_vector_type(4)));

MF_DEV void dma_b128(mf_v4i srd, unsigned lds_addr, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(lds_addr) : "memory");
}
MF_DEV void dma_b32(mf_v4i srd, unsigned lds_addr, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(lds_addr) : "memory");
}

// One streamed array: ROWB bytes per (row, step), of which only the units flagged by Keep are fetched.
// UNIT is the DMA granule (16 when ROWB is a multiple of 16, else 4).  The LDS image of a stream is
// row-major [64 rows][U units], filled by U wave-instructions: instruction i, lane l carries unit
// p = 64 i + l, i.e. (row, unit) = divmod(p, U) - consecutive lanes read consecutive 16-B pieces of a row.
// (Rows are NOT padded to an odd stride: the per-lane row reads then take a 2..4-way LDS bank conflict, but
// the kernel reads ~600 B per lane per step, nowhere near LDS bandwidth, while padding would push the
// fp64 d=6 image over a quarter of the CU's 160 KB and cost a wave of occupancy.)
template <int ROWB, typename Keep> struct Stream {
    static constexpr int UNIT = (ROWB % 16 == 0) ? 16 : 4;
    static constexpr int UG = ROWB / UNIT;                       // units per row in global memory
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
    int q0, c0;                    // lane / U, lane % U
    MF_DEV void init(int lane) { q0 = lane / St::U; c0 = lane - q0 * St::U; }
    // rel_tab: LDS byte address of this stream's row-offset table; gtab: LDS byte address of the
    // compact-unit -> global byte offset table (only read when the stream drops units)
    MF_DEV void issue(const char* smem, mf_v4i srd, unsigned lds_base, int rel_tab, int gtab) const {
        unsigned vo[St::NI];
        MF_UNROLL for (int i = 0; i < St::NI; ++i) {
            constexpr int dummy = 0; (void)dummy;
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
        }
        MF_UNROLL for (int i = 0; i < St::NI; ++i) {
            if (St::UNIT == 16) dma_b128(srd, lds_base + i * 1024, vo[i]);
            else dma_b32(srd, lds_base + i * 256, vo[i]);
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
MF_DEV mf_v4i make_srd(unsigned long long base, unsigned long long end) {
    unsigned long long rem = end > base ? end - base : 0ull;
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

template <typename T, int D, int M> struct KfLdsCfg {
    static constexpr int S = sizeof(T);
    using StA = Stream<D * D * S, KeepAll>;
    using StC = Stream<D * D * S, KeepLower<D, S>>;
    using Stb = Stream<D * S, KeepAll>;
    using StH = Stream<M * D * S, KeepAll>;
    using Sty = Stream<M * S, KeepAll>;
    static constexpr int OFF_A = 0;
    static constexpr int OFF_C = OFF_A + StA::LDS_BYTES;
    static constexpr int OFF_b = OFF_C + StC::LDS_BYTES;
    static constexpr int OFF_H = OFF_b + Stb::LDS_BYTES;
    static constexpr int OFF_y = OFF_H + StH::LDS_BYTES;
    static constexpr int OFF_relA = OFF_y + ((Sty::LDS_BYTES + 15) / 16) * 16;   // row offsets of A and cholQ
    static constexpr int OFF_relb = OFF_relA + 256;
    static constexpr int OFF_relH = OFF_relb + 256;
    static constexpr int OFF_rely = OFF_relH + 256;
    static constexpr int OFF_gtabC = OFF_rely + 256;
    static constexpr int LDS_TOTAL = OFF_gtabC + ((StC::U * 4 + 15) / 16) * 16;
};

// One transition of the chain for this lane.  FIRST_SEP: the block on the left is this chunk's separator
// (it is not eliminated here; its coupling seeds the spike).  Register discipline: the only large live
// temporaries are Ci (lower), and ONE D x D array that is A -> B = C^-1 A -> Y = B L^-T -> W in turn;
// Q_k^-1 + H^T R^-1 H is formed after the elimination, when B is no longer needed.
template <typename T, int D, int M, bool SPIKE, bool FIRST_SEP>
MF_DEV void kf_lds_step(Elim<T, D, SPIKE>& E, LogAcc<T>& laC, T& acc_yry, T& acc_ww, const T (&C)[D][D],
                        T (&Bm)[D][D], const T (&mvec)[D], const T (&hk)[M * D], const T (&yk)[M],
                        const T (&Rsh)[M * M]) {
    T Ci[D][D], w[D];
    tri_inv_lower<T, D>(C, Ci, laC, E.bad);
    laC.renorm();
    trimul_lower_vec<T, D>(Ci, mvec, w);
    acc_ww += dot_self<T, D>(w);
    {
        // B = C^-1 A in place (row i only needs rows k <= i: go bottom-up)
        MF_UNROLL for (int i = D - 1; i >= 0; --i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T sacc = T(0);
                MF_UNROLL for (int k = 0; k <= i; ++k) sacc += Ci[i][k] * Bm[k][j];
                Bm[i][j] = sacc;
            }
    }
    T btw[D];
    gemv_t<T, D, D>(Bm, w, btw);
    if (FIRST_SEP) {
        syrk_tn_lower<T, D, D>(Bm, E.GU, T(1));
        MF_UNROLL for (int i = 0; i < D; ++i) E.gU[i] = -btw[i];
        neg_trimulT_lower_inplace<T, D, D>(Ci, Bm);       // S = -C^-T B
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) E.X[i][j] = Bm[i][j];
        trimulT_self_lower<T, D>(Ci, E.Phi);
        trimulT_lower_vec<T, D>(Ci, w, E.t);
        acc_yry += Obs<T, D, M>::apply(hk, yk, Rsh, M, E.Phi, E.t);
    } else {
        syrk_tn_lower<T, D, D>(Bm, E.Phi, T(1));
        MF_UNROLL for (int i = 0; i < D; ++i) E.t[i] -= btw[i];
        E.eliminate();
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
template <typename T, int D, int M, bool SPIKE>
__global__ void __launch_bounds__(64) kf_chunk_lds_kernel(KfArgs<T> a, long L, RedSys<T> out) {
    using Cfg = KfLdsCfg<T, D, M>;
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
    const unsigned long long offA0 = uniform64(offA), offb0 = uniform64(offb);
    const unsigned long long offH0 = uniform64(offH), offy0 = uniform64(offy);
    const bool rowok = valid && len > 0;
    {
        unsigned* tab = reinterpret_cast<unsigned*>(smem);
        tab[Cfg::OFF_relA / 4 + lane] = rowok ? (unsigned)(offA - offA0) : MF_DMA_INVALID;
        tab[Cfg::OFF_relb / 4 + lane] = rowok ? (unsigned)(offb - offb0) : MF_DMA_INVALID;
        tab[Cfg::OFF_relH / 4 + lane] = rowok ? (unsigned)(offH - offH0) : MF_DMA_INVALID;
        tab[Cfg::OFF_rely / 4 + lane] = rowok ? (unsigned)(offy - offy0) : MF_DMA_INVALID;
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
    dA.init(lane); dC.init(lane); db.init(lane); dH.init(lane); dy.init(lane);
    unsigned long long pA = (unsigned long long)a.A + offA0, pC = (unsigned long long)a.cholQ + offA0;
    unsigned long long pb = (unsigned long long)a.b + offb0, pH = (unsigned long long)a.H + offH0;
    unsigned long long py = (unsigned long long)a.y + offy0;
    const unsigned long long eA = (unsigned long long)a.A + (unsigned long long)a.B * nt * (D * D * S);
    const unsigned long long eC = (unsigned long long)a.cholQ + (unsigned long long)a.B * nt * (D * D * S);
    const unsigned long long eb = (unsigned long long)a.b + (unsigned long long)a.B * nt * (D * S);
    const unsigned long long eH = (unsigned long long)a.H + (unsigned long long)a.B * a.Tn * (M * D * S);
    const unsigned long long ey = (unsigned long long)a.y + (unsigned long long)a.B * a.Tn * (M * S);
    const unsigned lds0 = (unsigned)(size_t)smem;

    // ---- block 0 of chunk 0: the prior (plain loads; no DMA in flight yet) ------------------------------
    Elim<T, D, SPIKE> E;
    E.init();
    LogAcc<T> laC;
    laC.init();
    T acc_yry = T(0), acc_ww = T(0);
    T Rsh[M * M];                                  // shared observation precision, kept in registers
    MF_UNROLL for (int i = 0; i < M * M; ++i) Rsh[i] = a.Rinv[i];
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

    const RowReader<T, typename Cfg::StA> rA(smem, Cfg::OFF_A, lane);
    const RowReader<T, typename Cfg::StC> rC(smem, Cfg::OFF_C, lane);
    const RowReader<T, typename Cfg::Stb> rb(smem, Cfg::OFF_b, lane);
    const RowReader<T, typename Cfg::StH> rH(smem, Cfg::OFF_H, lane);
    const RowReader<T, typename Cfg::Sty> ry(smem, Cfg::OFF_y, lane);

    auto issue_small = [&]() {
        dC.issue(smem, make_srd(pC, eC), lds0 + Cfg::OFF_C, Cfg::OFF_relA, Cfg::OFF_gtabC);
        db.issue(smem, make_srd(pb, eb), lds0 + Cfg::OFF_b, Cfg::OFF_relb, 0);
        dH.issue(smem, make_srd(pH, eH), lds0 + Cfg::OFF_H, Cfg::OFF_relH, 0);
        dy.issue(smem, make_srd(py, ey), lds0 + Cfg::OFF_y, Cfg::OFF_rely, 0);
    };
    auto issue_A = [&]() { dA.issue(smem, make_srd(pA, eA), lds0 + Cfg::OFF_A, Cfg::OFF_relA, 0); };

    if (nsteps > 0) {
        issue_small();
        issue_A();
    }

    // One streamed step: wait for the data of step j, move it to registers, refill the LDS regions for
    // step j+1 as soon as they have been read, then do the arithmetic.  FIRST distinguishes step 0 (whose
    // left block is the separator for every chunk but the first) so that the steady-state loop body holds
    // a single code path.
#define MF_KF_LDS_STEP(FIRST)                                                                                         \
    {                                                                                                                 \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
        const bool more = (j + 1 < nsteps);                                                                           \
        pA += D * D * S; pC += D * D * S; pb += D * S; pH += M * D * S; py += M * S;                                  \
        T C[D][D], mvec[D], hk[M * D], yk[M];                                                                         \
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int jj = 0; jj <= i; ++jj) C[i][jj] = rC.at(i * D + jj); \
        MF_UNROLL for (int i = 0; i < D; ++i) mvec[i] = rb.at(i);                                                     \
        MF_UNROLL for (int i = 0; i < M * D; ++i) hk[i] = rH.at(i);                                                   \
        MF_UNROLL for (int i = 0; i < M; ++i) yk[i] = ry.at(i);                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
        if (more) issue_small();                                                                                      \
        T Bm[D][D];                                                                                                   \
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int jj = 0; jj < D; ++jj) Bm[i][jj] = rA.at(i * D + jj); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
        if (more) issue_A();                                                                                          \
        if (j < len) {                                                                                                \
            if (FIRST && c > 0)                                                                                       \
                kf_lds_step<T, D, M, SPIKE, true>(E, laC, acc_yry, acc_ww, C, Bm, mvec, hk, yk, Rsh);                 \
            else                                                                                                      \
                kf_lds_step<T, D, M, SPIKE, false>(E, laC, acc_yry, acc_ww, C, Bm, mvec, hk, yk, Rsh);                \
        }                                                                                                             \
    }
    long j = 0;
    if (nsteps > 0) MF_KF_LDS_STEP(true)
    for (j = 1; j < nsteps; ++j) MF_KF_LDS_STEP(false)
#undef MF_KF_LDS_STEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (valid) {
        const T scalar = T(-0.5) * (acc_yry + acc_ww) + T(0.5) * E.quad - laC.value() - E.laL.value();
        store_chunk<T, D, SPIKE>(out, id, E, scalar);
        if (E.bad && a.info) atomicMax(a.info, 1);
    }
}

}  // namespace mf
