// The backward of KalmanFilter.log_likelihood for FEW, LONG series, streamed and partitioned in time (arithmetic and derivation:
// mf_grad_math.hpp).  After the three passes of the posterior chain (mf_post_lds.hpp: chunk summaries, scan, emit) two more:
//
//   4. grad_start_kernel   a wavefront per series composes the SAME chunk summaries in time order (Kogge-Stone over the lanes),
//      closes every prefix with the prior, and meets the state (Psi, psi) pass 2 left at the chunk boundary: the smoothed
//      marginal (m, S) of the first block of every chunk.
//   5. grad_lds_kernel     a lane per (series, chunk) walks its transitions FORWARD from that marginal; per transition it reads
//      the model (A, cholQ, b, H, y: the LDS-DMA streams of the log-likelihood kernel) and chol(Q'), b' of the posterior chain,
//      carries (m_k, S_k) in registers and writes every gradient of the transition once, through the LDS staging buffer of
//      the emit pass (consecutive lanes store consecutive 16-B units of a row).
//
// Against the route it replaces (posterior chain -> marginal means and covariances by two scans in time -> one lane per (series,
// time point) reading them back) the moments (2 d^2 + d values per step, written and read) never exist in HBM, and the
// transitions A' of the posterior chain are not read: A' = Q' Q^-1 A costs two triangular products.
// LDS: the image of one step (~54 KB at d = 6 fp64) + the record's offset table + staging -> two wavefronts per CU.
#pragma once
#include "mf_grad_math.hpp"
#include "mf_post_lds.hpp"

namespace mf {

template <typename T> struct GradIo {
    const void* rec_post;                             // the posterior chain (pass 3): packed records [chol(Q') | b'], PostLds::REC bytes each
    T* bPsi; T* bpsi;                                 // boundary states per consumer chunk (pass 2, or k0_scan_kernel)
    T* start_m; T* start_S;                           // [B, P, D], [B, P, D, D] (pass 4 -> pass 5)
    const T* mu0_post; const T* cp0_post;             // non-NULL: block 0's marginal is the chain's (mu0', cholP0' cholP0'^T)
    T* gmu0; T* gC0; T* gA; T* gb; T* gC; T* gH; T* gy; T* gOm;
};

// ---- pass 4 -------------------------------------------------------------------------------------------------------------------
// Summary j of the mirrored array stems from chunk P-1-j (pass 1 wrote them for the scan that runs from the last chunk); here
// chunk i is read at P-1-i.  The inclusive scan leaves in i the composition of chunks 0 .. i, whose remaining block is block 0
// and whose separator is the first block of chunk i+1.
template <typename T, int D, int M>
__global__ void __launch_bounds__(64) grad_start_kernel(KfArgs<T> a, RedSys<T> in, GradIo<T> io) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const long s = blockIdx.x;
    const long P = in.n;
    const long q = (P + 63) / 64;
    const long i0 = lane * q;
    long i1 = i0 + q;
    if (i1 > P) i1 = P;
    const bool has = i0 < P;
    bool bad = false;
    const PostScanLds<T, D> lds{reinterpret_cast<T*>(smem), lane};
    // what block 0 owns
    T C0[D][D], mu0[D], Lam0[D][D], lam0[D], Rsh[M * M];
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) { C0[i][j] = T(0); Lam0[i][j] = T(0); }
    load_lower<T, D>(a.cholP0 + s * D * D, C0);
    load_vec<T, D>(a.mu0 + s * D, mu0);
    MF_UNROLL for (int i = 0; i < M * M; ++i) Rsh[i] = a.Rinv[(a.rinv_per_step ? (s * a.Tn) * M * M : 0) + i];
    grad_prior_terms<T, D, M>(C0, mu0, a.H + (s * a.Tn) * M * D, a.y + (s * a.Tn) * M, Rsh, Lam0, lam0, bad);
    // the marginal of the first block of chunk i+1 from the composition `run` of chunks 0 .. i; the composition of ALL chunks
    // gives block 0 its right-hand side (Psi_0, psi_0)
    auto emit = [&](long i, const PostSummary<T, D>& run) {
        T Lam[D][D], lam[D], Psi[D][D], psi[D], m[D], S[D][D];
        MF_UNROLL for (int r = 0; r < D; ++r) MF_UNROLL for (int c = 0; c < D; ++c) { Lam[r][c] = T(0); Psi[r][c] = T(0); S[r][c] = T(0); }
        if (i + 1 < P) {
            grad_close_prefix<T, D>(run, Lam0, lam0, Lam, lam, bad);
            load_lower<T, D>(io.bPsi + (s * P + i) * D * D, Psi);
            load_vec<T, D>(io.bpsi + (s * P + i) * D, psi);
            grad_marginal<T, D>(Lam, lam, Psi, psi, m, S, bad);
            store_vec<T, D>(io.start_m + (s * P + i + 1) * D, m);
            store_sym<T, D>(io.start_S + (s * P + i + 1) * D * D, S);
        } else {
            MF_UNROLL for (int r = 0; r < D; ++r) {
                lam[r] = lam0[r];
                psi[r] = run.tv[r];
                MF_UNROLL for (int c = 0; c <= r; ++c) { Lam[r][c] = Lam0[r][c]; Psi[r][c] = run.Dv[r][c]; }
            }
            grad_marginal<T, D>(Lam, lam, Psi, psi, m, S, bad);
            store_vec<T, D>(io.start_m + (s * P) * D, m);
            store_sym<T, D>(io.start_S + (s * P) * D * D, S);
        }
    };
    PostSummary<T, D> acc;
    if (has) {
        post_summary_load<T, D>(in, s * P + (P - 1 - i0), acc);
        for (long i = i0 + 1; i < i1; ++i) {
            PostSummary<T, D> nx;
            post_summary_load<T, D>(in, s * P + (P - 1 - i), nx);
            post_combine<T, D>(nx, acc, bad);                   // nx on the right: the result stays in acc
        }
    }
    const int nl = (int)((P + q - 1) / q);                      // lanes that hold a run
    for (int off = 1; off < nl; off <<= 1) {
        if (has) lds.put(acc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // one wave per workgroup: LDS operations execute in order
        __builtin_amdgcn_wave_barrier();
        if (has && lane >= off) {
            PostSummary<T, D> prev;
            lds.get(lane - off, prev);                          // the run on the left
            post_combine<T, D>(acc, prev, bad);
            acc = prev;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    if (q == 1) {
        if (has) emit(i0, acc);
    } else {
        if (has) lds.put(acc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (has) {
            PostSummary<T, D> run;
            if (lane > 0) lds.get(lane - 1, run);
            for (long i = i0; i < i1; ++i) {
                PostSummary<T, D> nx;
                post_summary_load<T, D>(in, s * P + (P - 1 - i), nx);
                if (lane > 0 || i > i0) post_combine<T, D>(nx, run, bad);
                else run = nx;
                emit(i, run);
            }
        }
    }
    if (bad && a.info) raise_info(a.info);
}

// ---- passes 1, 2 and 4 from the summaries the FORWARD evaluation left behind ---------------------------------------------------
// kf_chunk_lds_kernel (mf_kf_lds.hpp), the level-0 kernel of mf_kf_loglik, writes one summary per (series, chunk) in the same
// five-field form, for the forward elimination: (Dv, tv) the partial pivot / right-hand side of the chunk's LAST block including
// that block's own terms, (GU, gU) what the chunk's interior adds to its FIRST block (the separator on its left, own terms
// excluded), F the coupling of the two; chunk 0 has the prior eliminated into it.  The Schur complement of a chunk's interior
// does not depend on the direction it was eliminated in, so these summaries hold everything passes 1 and 2 compute again:
//   prefix compositions (chunks 0 .. j):  (Dv, tv) = everything on the LEFT of block (j+1) L, own terms included = (Lam, lam);
//   suffix compositions (chunks j .. P-1), their last block eliminated too:  (GU, gU) = everything on the RIGHT of block j L,
//   own terms excluded = (Psi, psi), the state the emit pass restarts from.
// post_combine composes two runs in either convention: its first argument is the run whose remaining block is the other's
// separator - here the run on the LEFT.  One wavefront per series, two launches (prefix, then suffix, which finishes the
// marginals); the backward's chunks are groups of k forward chunks.
template <typename T, int D, bool SUFFIX>
__global__ void __launch_bounds__(64) k0_scan_kernel(RedSys<T> in, long B, int G, long k, long Pg, GradIo<T> io, int* info) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // G lanes per series (a power of two >= the lanes that hold a run; 64 when a series has more than 32 chunks): with few chunks
    // per series a wavefront scans 64 / G series side by side
    const int lane = threadIdx.x, lp = lane & (G - 1);
    const long s_raw = (long)blockIdx.x * (64 / G) + lane / G;
    const long s = s_raw < B ? s_raw : B - 1;
    const long P = in.n;
    const long q = (P + 63) / 64;
    const long p0 = lp * q;                   // positions in scan order: position p is chunk p (prefix) or chunk P-1-p (suffix)
    long p1 = p0 + q;
    if (p1 > P) p1 = P;
    const bool has = p0 < P && s_raw < B;
    bool bad = false;
    const PostScanLds<T, D> lds{reinterpret_cast<T*>(smem), lane};
    auto chunk_of = [&](long p) { return SUFFIX ? P - 1 - p : p; };
    // acc <- composition of `acc` (earlier positions) and `nx` (the next position)
    auto fold = [&](PostSummary<T, D>& acc, PostSummary<T, D>& nx) {
        if constexpr (SUFFIX) post_combine<T, D>(nx, acc, bad);            // nx is on the left: the result stays in acc
        else { post_combine<T, D>(acc, nx, bad); acc = nx; }               // acc is on the left: the result lands in nx
    };
    auto emit = [&](long p, const PostSummary<T, D>& run) {
        const long j = chunk_of(p);
        if constexpr (!SUFFIX) {
            if ((j + 1) % k == 0 && (j + 1) / k < Pg) {
                const long c = (j + 1) / k;
                store_sym<T, D>(io.start_S + (s * Pg + c) * D * D, run.Dv);
                store_vec<T, D>(io.start_m + (s * Pg + c) * D, run.tv);
            }
        } else {
            if (j % k == 0 && j >= 1) {
                const long c = j / k;
                PostSummary<T, D> z;
                MF_UNROLL for (int r = 0; r < D; ++r) {
                    z.tv[r] = T(0); z.gU[r] = T(0);
                    MF_UNROLL for (int cc = 0; cc < D; ++cc) { z.Dv[r][cc] = T(0); z.GU[r][cc] = T(0); z.F[r][cc] = T(0); }
                }
                post_combine<T, D>(run, z, bad);                            // the run's last block (block T-1) eliminated
                store_sym<T, D>(io.bPsi + (s * Pg + c - 1) * D * D, z.GU);
                store_vec<T, D>(io.bpsi + (s * Pg + c - 1) * D, z.gU);
                if (io.start_m != nullptr) {          // (NULL: the boundary states alone - the posterior chain from the filter's summaries)
                    T Lam[D][D], lam[D], Psi[D][D], m[D], S[D][D];
                    MF_UNROLL for (int r = 0; r < D; ++r) MF_UNROLL for (int cc = 0; cc < D; ++cc) { Lam[r][cc] = T(0); S[r][cc] = T(0); Psi[r][cc] = z.GU[r][cc]; }
                    load_lower<T, D>(io.start_S + (s * Pg + c) * D * D, Lam);
                    load_vec<T, D>(io.start_m + (s * Pg + c) * D, lam);
                    grad_marginal<T, D>(Lam, lam, Psi, z.gU, m, S, bad);
                    store_vec<T, D>(io.start_m + (s * Pg + c) * D, m);
                    store_sym<T, D>(io.start_S + (s * Pg + c) * D * D, S);
                }
            }
        }
    };
    PostSummary<T, D> acc;
    if (has) {
        post_summary_load<T, D>(in, s * P + chunk_of(p0), acc);
        for (long p = p0 + 1; p < p1; ++p) {
            PostSummary<T, D> nx;
            post_summary_load<T, D>(in, s * P + chunk_of(p), nx);
            fold(acc, nx);
        }
    }
    const int nl = (int)((P + q - 1) / q);
    for (int off = 1; off < nl; off <<= 1) {
        if (has) lds.put(acc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (has && lp >= off) {
            PostSummary<T, D> prev;
            lds.get(lane - off, prev);
            fold(prev, acc);                                              // prev: the earlier positions
            if constexpr (SUFFIX) acc = prev;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    if (q == 1) {
        if (has) emit(p0, acc);
    } else {
        if (has) lds.put(acc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (has) {
            PostSummary<T, D> run;
            if (lp > 0) lds.get(lane - 1, run);
            for (long p = p0; p < p1; ++p) {
                PostSummary<T, D> nx;
                post_summary_load<T, D>(in, s * P + chunk_of(p), nx);
                if (lp > 0 || p > p0) fold(run, nx);
                else run = nx;
                emit(p, run);
            }
        }
    }
    if (bad && info) raise_info(info);
}

// ---- pass 5 -------------------------------------------------------------------------------------------------------------------
template <typename T, int D, int M, bool RSTEP> struct GradLds {
    static constexpr int BG = backward_row_group(M);  // b and H rows are fetched in pairs (two wavefronts per CU: the LDS is there)
    using Cfg = KfLdsCfg<T, D, M, RSTEP, BG>;
    using PL = PostLds<T, D, M, RSTEP, BG>;
    static constexpr int S = (int)sizeof(T);
    static constexpr int H0 = PL::H0, B0 = PL::B0, B1 = PL::B1, Bv = PL::Bv, UNIT = PL::UNIT, U0 = PL::U0, U1 = PL::U1, Uv = PL::Uv;
    static constexpr int BH = M * D * S;                                           // a row of d/dH
    static constexpr int unit_h() { for (int u = 16; u > 4; u /= 2) if (BH % u == 0) return u; return 4; }
    static constexpr int UNITH = unit_h(), UH = BH / UNITH;
    // the chain's record of the step (PostLds: REC bytes = [chol(Q') lower, row-major | b' | pad]) and, per lane, the source offsets
    // of its DMA instructions - in LDS, not in registers (the step has none to spare): 16 slots of 4 B
    using StP = Stream<PL::REC, KeepAll>;
    static constexpr int NG = PL::NG;
    static_assert(StP::NI <= 16, "record: at most 16 DMA instructions per step");
    static constexpr int OFF_P = ((Cfg::LDS_TOTAL + 15) / 16) * 16;
    static constexpr int OFF_voP = OFF_P + StP::LDS_BYTES;
    static constexpr int OFF_relP = OFF_voP + 64 * 64;
    static constexpr int OFF_stageM = OFF_relP + 256;
    // (ONE staging buffer: the pieces are staged and stored in bursts, one after the other)
    static constexpr int OFF_stagev = OFF_stageM;
    static constexpr int STAGE = 64 * (B0 > BH ? (B0 > Bv ? B0 : Bv) : (BH > Bv ? BH : Bv));
    static constexpr int OFF_len = OFF_stageM + ((STAGE + 15) / 16) * 16;
    static constexpr int TOTAL = OFF_len + 256;
    static constexpr int N_DMA = Cfg::StA::NI + Cfg::StC::NI + StP::NI + Cfg::Stb::NI + Cfg::StH::NI + Cfg::Sty::NI + (RSTEP ? Cfg::StR::NI : 0);
    static constexpr bool SUPPORTED = Cfg::SUPPORTED && N_DMA < 64 && TOTAL <= 80 * 1024;
};

// The next step's LDS-DMA, issued at the sites of grad_step: a stream's image is single-buffered, so its batch goes out after the
// last read of the current rows (C, H, y: in registers from the top of the step; A, G, b, b': read from the image where used).
template <typename T, int D, int M, bool RSTEP> struct GradPump {
    using GL = GradLds<T, D, M, RSTEP>;
    using Cfg = typename GL::Cfg;
    const DmaStream<typename Cfg::StA>& dA; const DmaStream<typename Cfg::StC>& dC;
    const DmaStream<typename Cfg::Stb>& db; const DmaStream<typename Cfg::StH>& dH;
    const DmaStream<typename Cfg::Sty>& dy; const DmaStream<typename Cfg::StR>& dR;
    mf_v4i sA, sC, sb, sH, sy, sR, sP;
    unsigned lds0;
    bool more, yfetch, bfetch;
    const char* votab;                                  // this lane's 16 record offsets (LDS)
    MF_DEV void all() const {
        dC.template issue<0, 64>(sC, lds0 + Cfg::OFF_C);
        issue_record<0>();
        db.template issue<0, 64>(sb, lds0 + Cfg::OFF_b);
        dH.template issue<0, 64>(sH, lds0 + Cfg::OFF_H);
        dy.template issue<0, 64>(sy, lds0 + Cfg::OFF_y);
        if (RSTEP) dR.template issue<0, 64>(sR, lds0 + Cfg::OFF_R);
        dA.template issue<0, 64>(sA, lds0 + Cfg::OFF_A);
    }
    // the record's DMA instructions, four per group; their source offsets come from the lane's LDS table
    template <int I0> MF_DEV void issue_record() const {
        constexpr int NI = GL::StP::NI;
        if constexpr (I0 < NI) {
            constexpr int N = (NI - I0) < 4 ? (NI - I0) : 4;
            const mf_v4i o = *reinterpret_cast<const mf_v4i*>(votab + I0 * 4);
            unsigned voff[N];
            MF_UNROLL for (int k = 0; k < N; ++k) voff[k] = (unsigned)o[k];
            dma_b128_group<(GL::StP::UG >= 8), N>(sP, lds0 + GL::OFF_P + I0 * 1024, voff);
            issue_record<I0 + N>();
        }
    }
    template <int K> MF_DEV void site() const {
        asm volatile("" ::: "memory");                  // the image changes behind the compiler's back: no LDS value survives a site
        if (!more) return;
        constexpr int Q = (Cfg::StA::NI + 3) / 4;
        if constexpr (K == 0) dC.template issue<0, 64>(sC, lds0 + Cfg::OFF_C);
        if constexpr (K == 1) {
            if (bfetch) dH.template issue<0, 64>(sH, lds0 + Cfg::OFF_H);
            if (yfetch) dy.template issue<0, 64>(sy, lds0 + Cfg::OFF_y);
            if (RSTEP) dR.template issue<0, 64>(sR, lds0 + Cfg::OFF_R);
        }
        if constexpr (K == 3) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the last reads of A, b, b' have their data
            if (bfetch) db.template issue<0, 64>(sb, lds0 + Cfg::OFF_b);
            dA.template issue<0, Q>(sA, lds0 + Cfg::OFF_A);
        }
        if constexpr (K == 4) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // ... and the last read of G
            issue_record<0>();
            dA.template issue<Q, 2 * Q>(sA, lds0 + Cfg::OFF_A);
        }
        if constexpr (K == 5) dA.template issue<2 * Q, 3 * Q>(sA, lds0 + Cfg::OFF_A);
        if constexpr (K == 6) dA.template issue<3 * Q, 4 * Q>(sA, lds0 + Cfg::OFF_A);
    }
};

// The gradient rows of one step: matrix rows in two halves and d-vectors through the staging buffer (StagedPiece, mf_post_lds.hpp),
// stored in a burst as soon as they are staged; d/dy and Omega (M and M^2 values) straight from the lane.
// (GL: any layout with the piece geometry H0 ... UNITH, B0 and the offsets OFF_stageM, OFF_stagev, OFF_len - GradLds here, GprBwdLds
// in mf_gpr_grad.hpp)
template <typename T, int D, int M, typename GL> struct GradSinkT {
    static constexpr int H0 = GL::H0, U0 = GL::U0, U1 = GL::U1, Uv = GL::Uv, UNIT = GL::UNIT, UH = GL::UH, UNITH = GL::UNITH;
    using W = typename OutWord<UNIT>::type;
    using WH = typename OutWord<UNITH>::type;
    using P0 = StagedPiece<T, U0, UNIT>;
    using P1 = StagedPiece<T, U1, UNIT>;
    using Pv = StagedPiece<T, Uv, UNIT>;
    using PH = StagedPiece<T, UH, UNITH>;
    char* smem; int lane;
    static constexpr bool SAME_HALVES = (U1 == U0);  // even d: the second half of a matrix row has the offsets of the first
    DmaStream<OutPiece<U0, UNIT>> d0;
    DmaStream<OutPiece<(U1 > 0 && !SAME_HALVES ? U1 : 1), UNIT>> d1;
    DmaStream<OutPiece<Uv, UNIT>> dv;
    DmaStream<OutPiece<UH, UNITH>> dh;
    unsigned long long qA, qC, qb, qH, fA, fC, fb, fH;     // the wave's rows at this position; the tensors' ends
    T* gy; T* gOm;                                        // this lane's time point
    long e, minlen;
    W ma, mb;
    WH ha, hb;

    MF_DEV void init(char* smem_, int lane_, int rel_mat, int rel_vec, int rel_h) {
        smem = smem_; lane = lane_;
        d0.init(smem, lane, rel_mat, 0);
        if constexpr (U1 > 0 && !SAME_HALVES) d1.init(smem, lane, rel_mat, 0);
        dv.init(smem, lane, rel_vec, 0);
        dh.init(smem, lane, rel_h, 0);
    }
    template <typename P, int NU, typename WW, typename DS>
    MF_DEV void burst(int off_stage, const DS& ds, unsigned long long q, unsigned long long f, const T* row, WW& wa, WW& wb) {
        const mf_v4i srd = make_srd(q, f);
        P::stage(smem, off_stage, lane, row, wa);
        static_for<0, NU>([&](auto ic) {
            P::template unit<decltype(ic)::value>(smem, off_stage, GL::OFF_len, lane, ds.vo, srd, e < minlen, e, wa, wb);
        });
    }
    template <int HALF, int R> MF_DEV void put_mat(unsigned long long q, unsigned long long f, const T (&rows)[R][D]) {
        T row[R * D];
        MF_UNROLL for (int i = 0; i < R; ++i) MF_UNROLL for (int j = 0; j < D; ++j) row[i * D + j] = rows[i][j];
        if constexpr (HALF == 0) burst<P0, U0>(GL::OFF_stageM, d0, q, f, row, ma, mb);
        else if constexpr (SAME_HALVES) burst<P1, U1>(GL::OFF_stageM, d0, q + GL::B0, f, row, ma, mb);
        else burst<P1, U1>(GL::OFF_stageM, d1, q + GL::B0, f, row, ma, mb);
    }
    template <int HALF, int R> MF_DEV void put_gA(const T (&rows)[R][D], bool) { put_mat<HALF, R>(qA, fA, rows); }
    template <int HALF, int R> MF_DEV void put_gC(const T (&rows)[R][D], bool) { put_mat<HALF, R>(qC, fC, rows); }
    // (a gradient the caller did not ask for - NULL - is not stored: wave-uniform branches)
    bool want_b, want_H, want_y, want_Om;
    MF_DEV void put_gb(const T (&v)[D], bool) {
        if (want_b) burst<Pv, Uv>(GL::OFF_stagev, dv, qb, fb, v, ma, mb);
    }
    MF_DEV void put_obs(const T (&gH)[M * D], const T (&gyv)[M], const T (&gOmv)[M * M], bool active) {
        if (want_H) burst<PH, UH>(GL::OFF_stagev, dh, qH, fH, gH, ha, hb);
        if (active && want_y) { MF_UNROLL for (int i = 0; i < M; ++i) gy[i] = gyv[i]; }
        if (active && want_Om) { MF_UNROLL for (int i = 0; i < M * M; ++i) gOm[i] = gOmv[i]; }
    }
};

// One wavefront per workgroup = 64 (series, chunk) lanes; KfArgs::P = chunks per series, L = transitions per chunk.  Position j of
// a chunk = transition tau0 + j; a chunk shorter than the wave's longest idles LAST (its rows are then dropped by the range check).
template <typename T, int D, int M, bool RSTEP>
__global__ void __launch_bounds__(64) grad_lds_kernel(KfArgs<T> a, long L, GradIo<T> io) {
    using GL = GradLds<T, D, M, RSTEP>;
    using Cfg = typename GL::Cfg;
    using Sink = GradSinkT<T, D, M, GL>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const long total = a.B * a.P;
    const long id_raw = (long)blockIdx.x * 64 + lane;
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;
    const long s = id / a.P, c = id % a.P;
    const long nt = a.Tn - 1;
    const long tau0 = c * L;
    long len = nt - tau0;
    if (len > L) len = L;
    if (len < 0 || !valid) len = 0;
    constexpr int S = sizeof(T);
    long nsteps = len, minlen = len;
    MF_UNROLL for (int off = 32; off > 0; off >>= 1) {
        const long o = __shfl_xor((long long)nsteps, off);
        nsteps = o > nsteps ? o : nsteps;
        const long u = __shfl_xor((long long)minlen, off);
        minlen = u < minlen ? u : minlen;
    }
    nsteps = __builtin_amdgcn_readfirstlane((int)nsteps);
    minlen = __builtin_amdgcn_readfirstlane((int)minlen);

    // ---- row offsets (LDS tables) and wave-uniform stream pointers, as in kf_chunk_lds_kernel -------------------------------
    const unsigned long long offA = (unsigned long long)(s * nt + tau0) * (D * D * S);
    const unsigned long long offb = (unsigned long long)(s * nt + tau0) * (D * S);
    // (the observation a step handles is that of the block its transition LEAVES: time point tau0 + j)
    const unsigned long long offH = (unsigned long long)(s * a.Tn + tau0) * (M * D * S);
    const unsigned long long offy = (unsigned long long)(s * a.Tn + tau0) * (M * S);
    const unsigned long long offR = (unsigned long long)(s * a.Tn + tau0) * (M * M * S);
    const unsigned long long offA0 = uniform64(offA), offb0 = uniform64(offb);
    const unsigned long long offH0 = uniform64(offH), offy0 = uniform64(offy), offR0 = uniform64(offR);
    constexpr int REC = GL::PL::REC;
    const unsigned long long offP = (unsigned long long)(s * nt + tau0) * REC, offP0 = uniform64(offP);
    const bool rowok = valid && len > 0;
    {
        unsigned* tab = reinterpret_cast<unsigned*>(smem);
        tab[Cfg::OFF_relA / 4 + lane] = rowok ? (unsigned)(offA - offA0) : MF_DMA_INVALID;
        tab[Cfg::OFF_relb / 4 + lane] = rowok ? (unsigned)(offb - offb0) : MF_DMA_INVALID;
        tab[Cfg::OFF_relH / 4 + lane] = rowok ? (unsigned)(offH - offH0) : MF_DMA_INVALID;
        tab[Cfg::OFF_rely / 4 + lane] = rowok ? (unsigned)(offy - offy0) : MF_DMA_INVALID;
        tab[Cfg::OFF_relR / 4 + lane] = rowok ? (unsigned)(offR - offR0) : MF_DMA_INVALID;
        reinterpret_cast<int*>(smem)[GL::OFF_len / 4 + lane] = rowok ? (int)len : 0;
        tab[GL::OFF_relP / 4 + lane] = rowok ? (unsigned)(offP - offP0) : MF_DMA_INVALID;
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
    const unsigned long long nA = (unsigned long long)a.B * nt * (D * D * S), nb = (unsigned long long)a.B * nt * (D * S);
    const unsigned long long nH = (unsigned long long)a.B * a.Tn * (M * D * S), ny = (unsigned long long)a.B * a.Tn * (M * S);
    const unsigned long long nR = (unsigned long long)a.B * a.Tn * (M * M * S);
    unsigned long long pA = (unsigned long long)a.A + offA0, pC = (unsigned long long)a.cholQ + offA0;
    unsigned long long pP = (unsigned long long)io.rec_post + offP0;
    unsigned long long pb = (unsigned long long)a.b + offb0, pH = (unsigned long long)a.H + offH0;
    unsigned long long py = (unsigned long long)a.y + offy0;
    unsigned long long pR = (unsigned long long)a.Rinv + (RSTEP ? offR0 : 0ull);
    const unsigned long long eA = (unsigned long long)a.A + nA, eC = (unsigned long long)a.cholQ + nA;
    const unsigned long long eP = (unsigned long long)io.rec_post + (unsigned long long)a.B * nt * REC;
    const unsigned long long eb = (unsigned long long)a.b + nb, eH = (unsigned long long)a.H + nH, ey = (unsigned long long)a.y + ny;
    const unsigned long long eR = (unsigned long long)a.Rinv + (RSTEP ? nR : 0ull);
    const unsigned lds0 = (unsigned)(size_t)smem;

    // ---- the chunk's first block: its smoothed marginal; block 0 also owns the prior -----------------------------------------
    const T wgt = a.weights ? a.weights[s] : T(1);
    bool bad = false;
    T mk[D], Sk[D][D], Rsh[M * M];
    MF_UNROLL for (int i = 0; i < M * M; ++i) Rsh[i] = RSTEP ? T(0) : a.Rinv[i];
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Sk[i][j] = T(0);
    if (io.mu0_post != nullptr && c == 0) {          // block 0: the posterior chain starts from its marginal
        T G0[D][D];
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) G0[i][j] = T(0);
        load_lower<T, D>(io.cp0_post + s * D * D, G0);
        load_vec<T, D>(io.mu0_post + s * D, mk);
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                T acc = T(0);
                MF_UNROLL for (int l = 0; l <= j; ++l) acc += G0[i][l] * G0[j][l];
                Sk[i][j] = acc;
            }
    } else {
        load_vec<T, D>(io.start_m + id * D, mk);
        load_lower<T, D>(io.start_S + id * D * D, Sk);
    }
    if (valid && c == 0) {
        T C0[D][D], mu0[D], gmu0[D], gC0[D][D];
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) C0[i][j] = T(0);
        load_lower<T, D>(a.cholP0 + s * D * D, C0);
        load_vec<T, D>(a.mu0 + s * D, mu0);
        grad_prior<T, D>(C0, mu0, mk, Sk, wgt, gmu0, gC0, bad);
        store_vec<T, D>(io.gmu0 + s * D, gmu0);
        store_mat<T, D, D>(io.gC0 + s * D * D, gC0);
    }
    // the LDS tables must be visible before the first DMA address is formed, and the plain loads / stores above must be done before
    // DMAs are counted; touching the carried values keeps hipcc's wait-count pass from draining the DMA queue inside the loop
    // (mf_post_lds.hpp has the story)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    MF_UNROLL for (int i = 0; i < D; ++i) {
        asm volatile("" : "+v"(mk[i]));
        MF_UNROLL for (int j = 0; j <= i; ++j) asm volatile("" : "+v"(Sk[i][j]));
    }
    {
        T w_ = wgt;
        asm volatile("" : "+v"(w_));
    }
    dA.init(smem, lane, Cfg::OFF_relA, 0);
    dC.init(smem, lane, Cfg::OFF_relA, Cfg::OFF_gtabC);
    db.init(smem, lane, Cfg::OFF_relb, 0);
    dH.init(smem, lane, Cfg::OFF_relH, 0);
    dy.init(smem, lane, Cfg::OFF_rely, 0);
    if (RSTEP) dR.init(smem, lane, Cfg::OFF_relR, 0);
    const char* votab = smem + GL::OFF_voP + lane * 64;
    {
        DmaStream<typename GL::StP> dP;
        dP.init(smem, lane, GL::OFF_relP, 0);
        unsigned* vt = reinterpret_cast<unsigned*>(smem + GL::OFF_voP + lane * 64);
        MF_UNROLL for (int i = 0; i < 16; ++i) vt[i] = i < GL::StP::NI ? dP.vo[i] : MF_DMA_INVALID;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    Sink sink;
    sink.init(smem, lane, Cfg::OFF_relA, Cfg::OFF_relb, Cfg::OFF_relH);
    sink.fA = (unsigned long long)io.gA + nA; sink.fC = (unsigned long long)io.gC + nA;
    sink.fb = (unsigned long long)io.gb + nb; sink.fH = (unsigned long long)io.gH + nH;
    sink.minlen = minlen;
    sink.want_b = io.gb != nullptr; sink.want_H = io.gH != nullptr; sink.want_y = io.gy != nullptr; sink.want_Om = io.gOm != nullptr;
    unsigned long long qA = (unsigned long long)io.gA + offA0, qC = (unsigned long long)io.gC + offA0;
    unsigned long long qb = (unsigned long long)io.gb + offb0, qH = (unsigned long long)io.gH + offH0;
    T* gy_lane = io.gy + (s * a.Tn + tau0) * M;
    T* gOm_lane = io.gOm + (s * a.Tn + tau0) * M * M;

    const RowReader<T, typename Cfg::StA> rA(smem, Cfg::OFF_A, lane);
    const RowReader<T, typename Cfg::StC> rC(smem, Cfg::OFF_C, lane);
    const RowReader<T, typename GL::StP> rP(smem, GL::OFF_P, lane);
    const RowReader<T, typename Cfg::Stb> rb(smem, Cfg::OFF_b, lane);
    const RowReader<T, typename Cfg::StH> rH(smem, Cfg::OFF_H, lane);
    const RowReader<T, typename Cfg::Sty> ry(smem, Cfg::OFF_y, lane);
    const RowReader<T, typename Cfg::StR> rR(smem, Cfg::OFF_R, lane);
    using Pump = GradPump<T, D, M, RSTEP>;
    if (nsteps > 0) {
        const Pump p0{dA, dC, db, dH, dy, dR, make_srd(pA, eA), make_srd(pC, eC), make_srd(pb, eb), make_srd(pH, eH),
                      make_srd(py, ey), make_srd(pR, eR), make_srd(pP, eP), lds0, true, true, true, votab};
        p0.all();
    }
    for (long j = 0; j < nsteps; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const bool more = (j + 1 < nsteps);
        const bool yfetch = ((j + 1) % Cfg::YG) == 0;
        constexpr int BG = GL::BG;
        const bool bfetch = ((j + 1) % BG) == 0;
        const int gb = (int)(j % BG);
        pA += D * D * S; pC += D * D * S; pP += REC;
        if (bfetch) { pb += BG * D * S; pH += BG * M * D * S; }
        if (RSTEP) pR += M * M * S;
        if (yfetch) py += Cfg::YG * M * S;
        T C[D][D], hk[M * D], yk[M];
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int jj = 0; jj <= i; ++jj) C[i][jj] = rC.at(i * D + jj);
        MF_UNROLL for (int i = 0; i < M * D; ++i) hk[i] = *reinterpret_cast<const T*>(rH.row + (gb * M * D + i) * (int)sizeof(T));
        MF_UNROLL for (int i = 0; i < M; ++i)
            yk[i] = *reinterpret_cast<const T*>(ry.row + ((int)(j % Cfg::YG) * M + i) * (int)sizeof(T));
        if (RSTEP) { MF_UNROLL for (int i = 0; i < M * M; ++i) Rsh[i] = rR.at(i); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const Pump pump{dA, dC, db, dH, dy, dR, make_srd(pA, eA), make_srd(pC, eC), make_srd(pb, eb), make_srd(pH, eH),
                        make_srd(py, ey), make_srd(pR, eR), make_srd(pP, eP), lds0, more, yfetch, bfetch, votab};
        const bool active = j < len;
        sink.qA = qA; sink.qC = qC; sink.qb = qb; sink.qH = qH; sink.e = j;
        sink.gy = gy_lane; sink.gOm = gOm_lane;
        auto Aat = [&](int i, int jj) { return rA.at(i * D + jj); };
        auto Gat = [&](int i, int jj) { return rP.at(i * (i + 1) / 2 + jj); };
        auto bqat = [&](int i) { return *reinterpret_cast<const T*>(rb.row + (gb * D + i) * (int)sizeof(T)); };
        auto bpat = [&](int i) { return rP.at(GL::NG + i); };
        grad_step<T, D, M>(mk, Sk, bad, C, hk, yk, Rsh, wgt, Aat, Gat, bqat, bpat, pump, sink, active);
        qA += D * D * S; qC += D * D * S; qb += D * S; qH += M * D * S;
        gy_lane += M; gOm_lane += M * M;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (valid && len > 0 && tau0 + len == nt) {       // the last block of the series: its observation
        T gH[M * D], gyv[M], gOmv[M * M];
        const long k = s * a.Tn + nt;
        if (RSTEP) { MF_UNROLL for (int i = 0; i < M * M; ++i) Rsh[i] = a.Rinv[k * M * M + i]; }
        grad_obs<T, D, M>(a.H + k * M * D, a.y + k * M, Rsh, mk, Sk, wgt, gH, gyv, gOmv);
        if (io.gH) { MF_UNROLL for (int i = 0; i < M * D; ++i) io.gH[k * M * D + i] = gH[i]; }
        if (io.gy) { MF_UNROLL for (int i = 0; i < M; ++i) io.gy[k * M + i] = gyv[i]; }
        if (io.gOm) { MF_UNROLL for (int i = 0; i < M * M; ++i) io.gOm[k * M * M + i] = gOmv[i]; }
    }
    if (valid && bad && a.info) raise_info(a.info);
}

}  // namespace mf
