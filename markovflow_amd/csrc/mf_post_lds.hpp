// posterior_state_space_model (kalman_filter.py:109-182) for FEW, LONG series, fused and partitioned in time, with the
// memory side of the log-likelihood kernel (mf_kf_lds.hpp: every lane's next transition brought in by LDS-DMA, consecutive
// lanes reading consecutive 16-B pieces of a row).  Three passes, arithmetic in mf_post_math.hpp:
//
//   1. post_lds_kernel<EMIT = false>  a lane per (series, chunk) walks its transitions from the last to the first, assembles
//      the posterior precision on the way (state_space_model.py:431-483, kalman_filter.py:86-101,149-156) and eliminates
//      with the fill-in carried towards the block on the chunk's right: one summary per chunk.  Reads (2 d^2 + d + m d + m) s
//      bytes per step, writes nothing but the summaries.
//   2. post_scan_kernel               a wavefront per series composes the summaries (Kogge-Stone over the lanes): the state of
//      the backward recursion (Psi, psi) at every chunk boundary.
//   3. post_lds_kernel<EMIT = true>   every chunk restarts the textbook backward recursion (block_tri_diag.py:438-545) from
//      its boundary and writes the posterior chain: reads the same bytes again, writes (2 d^2 + d) s per step.
//
// Against the route it replaces for B < 2048 (mf_ssm_precision -> parallel-in-time U D U^T -> affine scan -> emit kernels:
// precision and factor written and re-read, ~6x the algorithmic traffic through per-lane row loads) the inputs are read
// twice, coalesced, and every output once.
#pragma once
#include "mf_kf_lds.hpp"
#include "mf_post_math.hpp"

namespace mf {

template <typename T> struct PostOut {
    T* a_post; T* mu0_post; T* b_post; T* cp0_post; T* cq_post;   // the posterior chain (EMIT)
    const T* bPsi; const T* bpsi;                                 // boundary state per consumer chunk [B, P, D, D] / [B, P, D] (EMIT)
};

// KfArgs::P = chunks per series, L = transitions per chunk.  Position e of a chunk = transition tau0 + e; the wave walks
// e = nsteps-1 ... 0 (a chunk shorter than the wave's longest idles FIRST, so that all lanes end on their chunk's first
// transition and every DMA address is >= the tensor's start).
template <typename T, int D, int M, bool RSTEP, bool EMIT>
__global__ void __launch_bounds__(64) post_lds_kernel(KfArgs<T> a, long L, RedSys<T> out, PostOut<T> po) {
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

    // wave-uniform trip count: the longest chunk in this wave
    long nsteps = len;
    MF_UNROLL for (int off = 32; off > 0; off >>= 1) {
        const long o = __shfl_xor((long long)nsteps, off);
        nsteps = o > nsteps ? o : nsteps;
    }
    nsteps = __builtin_amdgcn_readfirstlane((int)nsteps);

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
    // stream pointers at the wave's LAST position; they walk downwards
    const long e_top = nsteps > 0 ? nsteps - 1 : 0;
    unsigned long long pA = (unsigned long long)a.A + offA0 + (unsigned long long)e_top * (D * D * S);
    unsigned long long pC = (unsigned long long)a.cholQ + offA0 + (unsigned long long)e_top * (D * D * S);
    unsigned long long pb = (unsigned long long)a.b + offb0 + (unsigned long long)e_top * (D * S);
    unsigned long long pH = (unsigned long long)a.H + offH0 + (unsigned long long)e_top * (M * D * S);
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
    if (nsteps > 0) {   // prologue: fetch the last position
        Pump p0{dA, dC, db, dH, dy, dR, make_srd(pA, eA), make_srd(pC, eC), make_srd(pb, eb), make_srd(pH, eH),
                make_srd(py, ey), make_srd(pR, eR), lds0, true};
        p0.template all<0>();
    }
    // output rows of this lane's chunk (EMIT): transition tau0 + e of series s
    T* oA = po.a_post + (s * nt + tau0) * D * D;
    T* oC = po.cq_post + (s * nt + tau0) * D * D;
    T* ob = po.b_post + (s * nt + tau0) * D;

#define MF_POST_LDS_STEP(FIRST)                                                                                       \
    {                                                                                                                 \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
        const long e = nsteps - 1 - j;                                                                                \
        const bool more = e > 0;                                                                                      \
        const bool yfetch = (e % Cfg::YG) == 0;        /* position e-1 lies in the previous group of y rows */        \
        pA -= D * D * S; pC -= D * D * S; pb -= D * S; pH -= M * D * S; if (RSTEP) pR -= M * M * S;                   \
        if (yfetch) py -= Cfg::YG * M * S;                                                                            \
        T C[D][D], mvec[D], hk[M * D], yk[M];                                                                         \
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int jj = 0; jj <= i; ++jj) C[i][jj] = rC.at(i * D + jj); \
        MF_UNROLL for (int i = 0; i < D; ++i) mvec[i] = rb.at(i);                                                     \
        MF_UNROLL for (int i = 0; i < M * D; ++i) hk[i] = rH.at(i);                                                   \
        MF_UNROLL for (int i = 0; i < M; ++i)                                                                         \
            yk[i] = *reinterpret_cast<const T*>(ry.row + ((int)(e % Cfg::YG) * M + i) * (int)sizeof(T));              \
        if (RSTEP) { MF_UNROLL for (int i = 0; i < M * M; ++i) Rsh[i] = rR.at(i); }                                    \
        T Bm[D][D];                                                                                                   \
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int jj = 0; jj < D; ++jj) Bm[i][jj] = rA.at(i * D + jj); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
        const Pump pump{dA, dC, db, dH, dy, dR, make_srd(pA, eA), make_srd(pC, eC), make_srd(pb, eb),                 \
                        make_srd(pH, eH), make_srd(py, ey), make_srd(pR, eR), lds0, more, yfetch};                    \
        pump.unpumped();                                                                                              \
        const bool active = e < len;                                                                                  \
        if (EMIT) {                                                                                                   \
            T mean[D], Gi[D][D];                                                                                      \
            post_emit_step<T, D, M>(E.Phi, E.t, E.bad, C, mvec, hk, yk, Rsh, Bm, mean, Gi, pump, active);             \
            if (active) {                                                                                             \
                store_mat<T, D, D>(oA + e * D * D, Bm);                                                               \
                store_lower<T, D>(oC + e * D * D, Gi);                                                                \
                store_vec<T, D>(ob + e * D, mean);                                                                    \
            }                                                                                                         \
        } else {                                                                                                      \
            post_up_step<T, D, M, FIRST>(E, C, mvec, hk, yk, Rsh, Bm, pump, active, c + 1 < a.P);                     \
        }                                                                                                             \
    }
    long j = 0;
    if (nsteps > 0) MF_POST_LDS_STEP(true)
    for (j = 1; j < nsteps; ++j) MF_POST_LDS_STEP(false)
#undef MF_POST_LDS_STEP
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
