// The backward of GaussianProcessRegression.log_likelihood with the kernel -> state-space-model step FUSED into its passes
// (the training step of models/gaussian_process_regression.py:150-160 under a GradientTape; kernels/matern.py, sde_kernel.py:421-446
// for the closed forms): the emit pass and the gradient pass of the streamed backward (mf_post_lds.hpp MODE 2, mf_grad_lds.hpp)
// with A_k = exp(F dt_k) and chol Q_k generated in registers from (dt_k, hyper-parameters) as the fused forward does
// (mf_gpr_fused.hpp) - a step reads 16 bytes of the model instead of (2 d^2 + 3 d + 1) s, and the model tensors are never
// materialised.  Sum of one or two Matern components, one output, zero state offsets, H = [1 0 0 | 1 0 0].
//
//   forward (mf_gpr_matern_loglik)    leaves one summary per (series, chunk) in its workspace, in the form of mf_kf_loglik's
//   k0_scan_kernel x 2                 boundary states and start moments from those summaries (mf_grad_lds.hpp)
//   gpr_emit_kernel                    the posterior chain's chol(Q'), b' as packed records, backward in time
//   gpr_grad_kernel                    forward in time: smoothed marginals in registers; writes the diagonal blocks of g_A and of
//                                      g_cholQ as packed records (what the generator's backward, mf_sde_matern_transitions_grad_packed,
//                                      contracts with its forward-mode duals), g_cholP0 and Omega (the noise's gradient)
// No input image in LDS: both kernels run one wavefront per SIMD on the forward's own partition.
#pragma once
#include "mf_gpr_fused.hpp"
#include "mf_grad_lds.hpp"

namespace mf {

// LDS of the two kernels: [record image | A_k per lane (its NA non-zero entries: the diagonal blocks) | staging | tables]
template <typename T, int D, int NA> struct GprBwdLds {
    using PL = PostLds<T, D, 1, false, 1>;            // record and piece geometry only (its offsets are not used)
    static constexpr int S = (int)sizeof(T), M = 1;
    static constexpr int NG = PL::NG, NR = PL::NR, REC = PL::REC, RU = PL::RU, RUa = PL::RUa, RUb = PL::RUb, REa = PL::REa, REb = PL::REb;
    static constexpr int H0 = PL::H0, B0 = PL::B0, B1 = PL::B1, Bv = PL::Bv, UNIT = PL::UNIT, U0 = PL::U0, U1 = PL::U1, Uv = PL::Uv;
    static constexpr int BH = D * S, UNITH = UNIT, UH = Uv;         // (d/dH is not produced; the piece exists for the sink's types)
    using StP = Stream<REC, KeepAll>;
    static constexpr int OFF_P = 0;
    static constexpr int OFF_Amat = OFF_P + StP::LDS_BYTES;        // this step's A_k, NA entries per lane
    static constexpr int OFF_stageM = OFF_Amat + ((64 * NA * S + 15) / 16) * 16;
    static constexpr int OFF_stagev = OFF_stageM;
    // (one staging buffer for every piece: halves of the chain's record, the two parts of the gradient record)
    static constexpr int STAGE0 = 64 * (((NA * S + 15) / 16) * 16), STAGE1 = 64 * RUa * 16;
    static constexpr int OFF_len = OFF_stageM + (((STAGE0 > STAGE1 ? STAGE0 : STAGE1) + 15) / 16) * 16;
    static constexpr int OFF_relP = OFF_len + 256;
    static constexpr int OFF_relA = OFF_relP + 256;
    static constexpr int OFF_relv = OFF_relA + 256;
    static constexpr int OFF_hyp = OFF_relv + 256;                 // lam0, var0, lam1, var1, R^-1 per lane
    static constexpr int TOTAL = OFF_hyp + 5 * 64 * S;
    // the gradients as the generator's backward reads them (mf_sde_matern_transitions_grad_packed): one record per transition,
    // [the diagonal blocks of g_A, row-major | the lower triangles of the diagonal blocks of g_cholQ], each part padded to 16 bytes
    // - 240 B at d = 6 (3 + 3) fp64 instead of two d x d rows of 288 B, of which the generator ignores 58 %
    template <int K0, int K1> struct Grec {
        static constexpr int NGA = K0 * K0 + K1 * K1, NGC = K0 * (K0 + 1) / 2 + K1 * (K1 + 1) / 2;
        static constexpr int RGA = ((NGA * S + 15) / 16) * 16, RGC = ((NGC * S + 15) / 16) * 16, RECG = RGA + RGC;
        static constexpr int NUA = RGA / 16, NUC = RGC / 16, EA = RGA / S, EC = RGC / S;
        static_assert(64 * RGA <= (STAGE0 > STAGE1 ? STAGE0 : STAGE1) && 64 * RGC <= (STAGE0 > STAGE1 ? STAGE0 : STAGE1), "staging buffer");
    };
};

// The sink of the gradient pass: the packed record in two pieces (the blocks of g_A when its second half exists, the factor's
// when its second half exists), Omega straight from the lane; nothing else is wanted.
template <typename T, int D, int K0, int K1, typename GB> struct GprGradSink {
    using GR = typename GB::template Grec<K0, K1>;
    static constexpr int H0 = (D + 1) / 2, NUA = GR::NUA, NUC = GR::NUC;
    using W = typename OutWord<16>::type;
    using PA = StagedPiece<T, NUA, 16>;
    using PC = StagedPiece<T, NUC, 16>;
    char* smem; int lane;
    DmaStream<OutPiece<NUA, 16>> da;
    DmaStream<OutPiece<NUC, 16>> dc;
    unsigned long long qG, fG;
    T* gOm;
    long e, minlen;
    bool want_Om;
    W ma, mb;
    T recA[GR::EA], recC[GR::EC];
    MF_DEV void init(char* smem_, int lane_, int rel_g) {
        smem = smem_; lane = lane_;
        da.init(smem, lane, rel_g, 0);
        dc.init(smem, lane, rel_g, 0);
        MF_UNROLL for (int i = 0; i < GR::EA; ++i) recA[i] = T(0);
        MF_UNROLL for (int i = 0; i < GR::EC; ++i) recC[i] = T(0);
    }
    static constexpr bool same_block(int i, int j) { return (i < K0) == (j < K0); }
    static constexpr int idx_a(int i, int j) { return i < K0 ? i * K0 + j : K0 * K0 + (i - K0) * K1 + (j - K0); }
    static constexpr int idx_c(int i, int j) {
        return i < K0 ? i * (i + 1) / 2 + j : K0 * (K0 + 1) / 2 + (i - K0) * (i - K0 + 1) / 2 + (j - K0);
    }
    template <typename P, int NU, typename DS> MF_DEV void burst(const DS& ds, unsigned long long q, const T* row) {
        const mf_v4i srd = make_srd(q, fG);
        P::stage(smem, GB::OFF_stageM, lane, row, ma);
        static_for<0, NU>([&](auto ic) {
            P::template unit<decltype(ic)::value>(smem, GB::OFF_stageM, GB::OFF_len, lane, ds.vo, srd, e < minlen, e, ma, mb);
        });
    }
    template <int HALF, int R> MF_DEV void put_gA(const T (&rows)[R][D], bool) {
        MF_UNROLL for (int r = 0; r < R; ++r)
            MF_UNROLL for (int j = 0; j < D; ++j)
                if (same_block(HALF * H0 + r, j)) recA[idx_a(HALF * H0 + r, j)] = rows[r][j];
        if constexpr (HALF == 1 || D - H0 == 0) burst<PA, NUA>(da, qG, recA);
    }
    template <int HALF, int R> MF_DEV void put_gC(const T (&rows)[R][D], bool) {
        MF_UNROLL for (int r = 0; r < R; ++r)
            MF_UNROLL for (int j = 0; j < D; ++j)
                if (j <= HALF * H0 + r && same_block(HALF * H0 + r, j)) recC[idx_c(HALF * H0 + r, j)] = rows[r][j];
        if constexpr (HALF == 1 || D - H0 == 0) burst<PC, NUC>(dc, qG + GR::RGA, recC);
    }
    MF_DEV void put_gb(const T (&)[D], bool) {}
    MF_DEV void put_obs(const T (&)[D], const T (&)[1], const T (&gOmv)[1], bool active) {
        if (active && want_Om) gOm[0] = gOmv[0];
    }
};

template <typename T> struct GprBwdIo {
    void* rec; T* bPsi; T* bpsi; T* mu0_post; T* cp0_post;        // emit: records out; boundary states in; block 0's marginal out
    T* a_post; T* b_post; T* cq_post;                             // FULL: the posterior chain itself instead of the records
};

// Layout of the FULL emit (the posterior chain as a state space model, mf_gpr_matern_posterior_chain): the staging buffers and
// tables of the streamed emit pass (PostLds) - its input image region is unused and holds the hyper-parameters
template <typename T, int D> struct GprPostLds {
    using PL = PostLds<T, D, 1, false, 1>;
    static constexpr int OFF_hyp = 0, OFF_len = PL::OFF_len, OFF_relA = PL::Cfg::OFF_relA, OFF_relb = PL::Cfg::OFF_relb;
    static constexpr int OFF_relP = OFF_relA, REC = PL::REC;          // (names the record variant of the kernel refers to; unused)
    static constexpr int TOTAL = PL::TOTAL;
    static_assert(5 * 64 * (int)sizeof(T) <= PL::Cfg::OFF_relA, "hyper-parameters inside the unused image region");
};

template <typename T, int O0, int O1> MF_DEV GprGen<T, O0, O1> gpr_load_gen(const char* smem, int off_hyp, int lane, T jitter) {
    const T* h = reinterpret_cast<const T*>(smem + off_hyp);
    GprGen<T, O0, O1> g;
    g.lam[0] = h[0 * 64 + lane]; g.var[0] = h[1 * 64 + lane];
    g.lam[1] = h[2 * 64 + lane]; g.var[1] = h[3 * 64 + lane];
    g.jitter = jitter;
    return g;
}
template <typename T, int O1> MF_DEV void gpr_store_hyp(char* smem, int off_hyp, int lane, const GprArgs<T>& a, long s) {
    T* h = reinterpret_cast<T*>(smem + off_hyp);
    h[0 * 64 + lane] = a.lam[s * a.hstride];
    h[1 * 64 + lane] = a.var[s * a.hstride];
    h[2 * 64 + lane] = O1 ? a.lam[s * a.hstride + 1] : T(0);
    h[3 * 64 + lane] = O1 ? a.var[s * a.hstride + 1] : T(0);
    h[4 * 64 + lane] = a.rinv[0];
}

// ---- emit: position e of a chunk = transition tau0 + e; the wave walks e = nsteps-1 ... 0 (a shorter chunk idles FIRST) -----------
// FULL: all five tensors of the posterior chain through the sink of the streamed emit pass (three wavefronts per CU); else the
// packed records for the gradient pass.
template <typename T, int O0, int O1, bool FULL = false>
__global__ void __launch_bounds__(64) gpr_emit_kernel(GprArgs<T> a, GprBwdIo<T> io) {
    using Gen = GprGen<T, O0, O1>;
    constexpr int D = Gen::D, S = sizeof(T);
    using GBR = GprBwdLds<T, D, Gen::K0 * Gen::K0 + Gen::K1 * Gen::K1>;
    using GPL = GprPostLds<T, D>;
    using GB = std::conditional_t<FULL, GPL, GBR>;
    using Sink = std::conditional_t<FULL, PostSink<T, D, 1, false, true, 1>, PackedSinkT<T, D, GBR>>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const long total = a.B * a.P;
    const long id_raw = (long)blockIdx.x * 64 + lane;
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;
    const long s = id / a.P, c = id % a.P;
    const long nt = a.Tn - 1, tau0 = c * a.L;
    long len = nt - tau0;
    if (len > a.L) len = a.L;
    if (len < 0 || !valid) len = 0;
    long nsteps = len, minlen = len;
    MF_UNROLL for (int off = 32; off > 0; off >>= 1) {
        const long o = __shfl_xor((long long)nsteps, off);
        nsteps = o > nsteps ? o : nsteps;
        const long u = __shfl_xor((long long)minlen, off);
        minlen = u < minlen ? u : minlen;
    }
    nsteps = __builtin_amdgcn_readfirstlane((int)nsteps);
    minlen = __builtin_amdgcn_readfirstlane((int)minlen);
    const unsigned long long offP = (unsigned long long)(s * nt + tau0) * GB::REC, offP0 = uniform64(offP);
    const unsigned long long offA = (unsigned long long)(s * nt + tau0) * (D * D * S), offA0 = uniform64(offA);
    const unsigned long long offb = (unsigned long long)(s * nt + tau0) * (D * S), offb0 = uniform64(offb);
    const bool rowok = valid && len > 0;
    {
        unsigned* tab = reinterpret_cast<unsigned*>(smem);
        if constexpr (FULL) {
            tab[GPL::OFF_relA / 4 + lane] = rowok ? (unsigned)(offA - offA0) : MF_DMA_INVALID;
            tab[GPL::OFF_relb / 4 + lane] = rowok ? (unsigned)(offb - offb0) : MF_DMA_INVALID;
        } else {
            tab[GB::OFF_relP / 4 + lane] = rowok ? (unsigned)(offP - offP0) : MF_DMA_INVALID;
        }
        reinterpret_cast<int*>(smem)[GB::OFF_len / 4 + lane] = rowok ? (int)len : 0;
    }
    gpr_store_hyp<T, O1>(smem, GB::OFF_hyp, lane, a, s);
    T hk[D], zero[D];
    MF_UNROLL for (int i = 0; i < D; ++i) { hk[i] = (i == 0 || (O1 && i == Gen::K0)) ? T(1) : T(0); zero[i] = T(0); }
    const T* ts = a.t + s * a.Tn;
    const T* ys = a.y + s * a.Tn;
    T Phi[D][D], tv[D];
    bool bad = false;
    MF_UNROLL for (int i = 0; i < D; ++i) { tv[i] = T(0); MF_UNROLL for (int j = 0; j < D; ++j) Phi[i][j] = T(0); }
    if (valid && c + 1 < a.P) {
        load_lower<T, D>(io.bPsi + id * D * D, Phi);
        load_vec<T, D>(io.bpsi + id * D, tv);
    }
    // time points and observations one position ahead of their use; a chunk that is not active yet holds its LAST position's
    const long e_top = nsteps > 0 ? nsteps - 1 : 0;
    const long e0 = len > 0 ? (e_top < len ? e_top : len - 1) : 0;
    T t_hi = len > 0 ? ts[tau0 + e0 + 1] : T(0), t_lo = len > 0 ? ts[tau0 + e0] : T(0), y_cur = len > 0 ? ys[tau0 + e0 + 1] : T(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    MF_UNROLL for (int i = 0; i < D; ++i) {
        asm volatile("" : "+v"(tv[i]));
        MF_UNROLL for (int j = 0; j <= i; ++j) asm volatile("" : "+v"(Phi[i][j]));
    }
    Sink sink;
    unsigned long long qR = 0, qA = 0, qC = 0, qb = 0;
    if constexpr (FULL) {
        sink.init(smem, lane, GPL::OFF_relA, GPL::OFF_relb);
        const unsigned long long nA = (unsigned long long)a.B * nt * (D * D * S), nb = (unsigned long long)a.B * nt * (D * S);
        sink.fA = (unsigned long long)io.a_post + nA; sink.fC = (unsigned long long)io.cq_post + nA;
        sink.fb = (unsigned long long)io.b_post + nb;
        qA = (unsigned long long)io.a_post + offA0 + (unsigned long long)e_top * (D * D * S);
        qC = (unsigned long long)io.cq_post + offA0 + (unsigned long long)e_top * (D * D * S);
        qb = (unsigned long long)io.b_post + offb0 + (unsigned long long)e_top * (D * S);
    } else {
        sink.init(smem, lane, 0, 0);
        sink.fR = (unsigned long long)io.rec + (unsigned long long)a.B * nt * GB::REC;
        qR = (unsigned long long)io.rec + offP0 + (unsigned long long)e_top * GB::REC;
    }
    sink.minlen = minlen;
    sink.e = 0;
    const NoPump pump;
    for (long j = 0; j < nsteps; ++j) {
        const long e = nsteps - 1 - j;
        const bool active = e < len;
        const T dt = t_hi - t_lo;
        T yk[1] = {y_cur}, Rk[1] = {reinterpret_cast<const T*>(smem + GB::OFF_hyp)[4 * 64 + lane]};
        // the next position this lane will be ACTIVE at: e-1 once it is active, still len-1 before
        const long en = e - 1 < len ? e - 1 : len - 1;
        if (en >= 0 && en != (e < len ? e : len - 1)) { t_hi = t_lo; t_lo = ts[tau0 + en]; y_cur = ys[tau0 + en + 1]; }
        T C[D][D], Bm[D][D];
        gpr_load_gen<T, O0, O1>(smem, GB::OFF_hyp, lane, a.jitter).make(dt, false, Bm, C);
        sink.e = e;
        if constexpr (FULL) {
            sink.qA = qA; sink.qC = qC; sink.qb = qb;
            qA -= D * D * S; qC -= D * D * S; qb -= D * S;
        } else {
            sink.qR = qR;
            qR -= GB::REC;
        }
        post_emit_step<T, D, 1, FULL>(Phi, tv, bad, C, zero, hk, yk, Rk, Bm, pump, sink, active);
    }
    if constexpr (FULL) sink.flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (valid && c == 0) {          // block 0: the stationary prior closes the chain
        T C0[D][D], dummy[D][D], mean[D], Gi[D][D];
        gpr_load_gen<T, O0, O1>(smem, GB::OFF_hyp, lane, a.jitter).make(T(0), true, dummy, C0);
        T y0[1] = {ys[0]}, Rk[1] = {a.rinv[0]};
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Gi[i][j] = T(0);
        post_emit_prior<T, D, 1>(Phi, tv, bad, C0, zero, hk, y0, Rk, mean, Gi);
        store_vec<T, D>(io.mu0_post + s * D, mean);
        store_lower<T, D>(io.cp0_post + s * D * D, Gi);
    }
    if (valid && bad && a.info) raise_info(a.info);
}

// ---- gradient pass: position j of a chunk = transition tau0 + j, forward in time (a shorter chunk idles LAST) ----------------------
template <typename T, typename GB> struct GprGradPump {
    const DmaStream<typename GB::StP>& dP;
    mf_v4i sP;
    unsigned lds0;
    bool more;
    MF_DEV void issue_record() const { dP.template issue<0, 64>(sP, lds0 + GB::OFF_P); }
    template <int K> MF_DEV void site() const {
        asm volatile("" ::: "memory");
        if constexpr (K == 4) {
            if (!more) return;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the last read of the record's factor has its data
            issue_record();
        }
    }
};

template <typename T, int O0, int O1>
__global__ void __launch_bounds__(64) gpr_grad_kernel(GprArgs<T> a, GradIo<T> io, const T* __restrict__ weights) {
    using Gen = GprGen<T, O0, O1>;
    constexpr int D = Gen::D, S = sizeof(T);
    constexpr int K0 = Gen::K0, K1 = Gen::K1, NA = K0 * K0 + K1 * K1;
    using GB = GprBwdLds<T, D, NA>;
    using GR = typename GB::template Grec<K0, K1>;
    using Sink = GprGradSink<T, D, K0, K1, GB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const long total = a.B * a.P;
    const long id_raw = (long)blockIdx.x * 64 + lane;
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;
    const long s = id / a.P, c = id % a.P;
    const long nt = a.Tn - 1, tau0 = c * a.L;
    long len = nt - tau0;
    if (len > a.L) len = a.L;
    if (len < 0 || !valid) len = 0;
    long nsteps = len, minlen = len;
    MF_UNROLL for (int off = 32; off > 0; off >>= 1) {
        const long o = __shfl_xor((long long)nsteps, off);
        nsteps = o > nsteps ? o : nsteps;
        const long u = __shfl_xor((long long)minlen, off);
        minlen = u < minlen ? u : minlen;
    }
    nsteps = __builtin_amdgcn_readfirstlane((int)nsteps);
    minlen = __builtin_amdgcn_readfirstlane((int)minlen);
    const unsigned long long offG = (unsigned long long)(s * nt + tau0) * GR::RECG, offG0 = uniform64(offG);
    const unsigned long long offP = (unsigned long long)(s * nt + tau0) * GB::REC, offP0 = uniform64(offP);
    const bool rowok = valid && len > 0;
    {
        unsigned* tab = reinterpret_cast<unsigned*>(smem);
        tab[GB::OFF_relA / 4 + lane] = rowok ? (unsigned)(offG - offG0) : MF_DMA_INVALID;        // rows of the gradient records
        tab[GB::OFF_relP / 4 + lane] = rowok ? (unsigned)(offP - offP0) : MF_DMA_INVALID;
        reinterpret_cast<int*>(smem)[GB::OFF_len / 4 + lane] = rowok ? (int)len : 0;
    }
    gpr_store_hyp<T, O1>(smem, GB::OFF_hyp, lane, a, s);
    T hk[D];
    MF_UNROLL for (int i = 0; i < D; ++i) hk[i] = (i == 0 || (O1 && i == Gen::K0)) ? T(1) : T(0);
    const T* ts = a.t + s * a.Tn;
    const T* ys = a.y + s * a.Tn;
    const T wgt = weights ? weights[s] : T(1);
    bool bad = false;
    T mk[D], Sk[D][D];
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Sk[i][j] = T(0);
    if (c == 0) {                    // block 0: the posterior chain starts from its marginal
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
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (valid && c == 0) {          // the prior's gradient: mu0 = 0, cholP0 = chol(Pinf + jitter)
        T C0[D][D], dummy[D][D], mu0[D], gmu0[D], gC0[D][D];
        gpr_load_gen<T, O0, O1>(smem, GB::OFF_hyp, lane, a.jitter).make(T(0), true, dummy, C0);
        MF_UNROLL for (int i = 0; i < D; ++i) mu0[i] = T(0);
        grad_prior<T, D>(C0, mu0, mk, Sk, wgt, gmu0, gC0, bad);
        store_mat<T, D, D>(io.gC0 + s * D * D, gC0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    MF_UNROLL for (int i = 0; i < D; ++i) {
        asm volatile("" : "+v"(mk[i]));
        MF_UNROLL for (int j = 0; j <= i; ++j) asm volatile("" : "+v"(Sk[i][j]));
    }
    DmaStream<typename GB::StP> dP;
    dP.init(smem, lane, GB::OFF_relP, 0);
    Sink sink;
    sink.init(smem, lane, GB::OFF_relA);
    sink.fG = (unsigned long long)io.gA + (unsigned long long)a.B * nt * GR::RECG;      // io.gA: the packed gradient records
    sink.minlen = minlen;
    sink.want_Om = io.gOm != nullptr;
    unsigned long long qG = (unsigned long long)io.gA + offG0;
    T* gOm_lane = io.gOm + (s * a.Tn + tau0);
    const unsigned lds0 = (unsigned)(size_t)smem;
    unsigned long long pP = (unsigned long long)io.rec_post + offP0;
    const unsigned long long eP = (unsigned long long)io.rec_post + (unsigned long long)a.B * nt * GB::REC;
    const RowReader<T, typename GB::StP> rP(smem, GB::OFF_P, lane);
    T* Amat = reinterpret_cast<T*>(smem + GB::OFF_Amat + lane * (NA * S));
    using Pump = GprGradPump<T, GB>;
    if (nsteps > 0) {
        const Pump p0{dP, make_srd(pP, eP), lds0, true};
        p0.issue_record();
    }
    // time points and observations one position ahead of their use (an idle lane keeps reading its last position)
    T t_lo = len > 0 ? ts[tau0] : T(0), t_hi = len > 0 ? ts[tau0 + 1] : T(0), y_cur = len > 0 ? ys[tau0] : T(0);
    for (long j = 0; j < nsteps; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const bool more = (j + 1 < nsteps);
        const bool active = j < len;
        pP += GB::REC;
        const T dt = t_hi - t_lo;
        T yk[1] = {y_cur}, Rk[1] = {reinterpret_cast<const T*>(smem + GB::OFF_hyp)[4 * 64 + lane]};
        if (j + 1 < len) { t_lo = t_hi; t_hi = ts[tau0 + j + 2]; y_cur = ys[tau0 + j + 1]; }
        T C[D][D];
        {
            T Am[D][D];
            gpr_load_gen<T, O0, O1>(smem, GB::OFF_hyp, lane, a.jitter).make(dt, false, Am, C);
            MF_UNROLL for (int i = 0; i < K0; ++i) MF_UNROLL for (int jj = 0; jj < K0; ++jj) Amat[i * K0 + jj] = Am[i][jj];
            MF_UNROLL for (int i = 0; i < K1; ++i) MF_UNROLL for (int jj = 0; jj < K1; ++jj) Amat[K0 * K0 + i * K1 + jj] = Am[K0 + i][K0 + jj];
        }
        const Pump pump{dP, make_srd(pP, eP), lds0, more};
        sink.qG = qG; sink.e = j;
        sink.gOm = gOm_lane;
        // (A_k is block diagonal: its zeros are known at compile time)
        auto Aat = [&](int i, int jj) {
            if (i < K0 && jj < K0) return Amat[i * K0 + jj];
            if (i >= K0 && jj >= K0) return Amat[K0 * K0 + (i - K0) * K1 + (jj - K0)];
            return T(0);
        };
        auto Gat = [&](int i, int jj) { return rP.at(i * (i + 1) / 2 + jj); };
        auto bqat = [&](int) { return T(0); };
        auto bpat = [&](int i) { return rP.at(GB::NG + i); };
        grad_step<T, D, 1>(mk, Sk, bad, C, hk, yk, Rk, wgt, Aat, Gat, bqat, bpat, pump, sink, active);
        qG += GR::RECG;
        gOm_lane += 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (valid && len > 0 && tau0 + len == nt && io.gOm) {       // the last block of the series: its observation
        T gH[D], gyv[1], gOmv[1], y1[1] = {ys[nt]}, Rk[1] = {a.rinv[0]};
        grad_obs<T, D, 1>(hk, y1, Rk, mk, Sk, wgt, gH, gyv, gOmv);
        io.gOm[s * a.Tn + nt] = gOmv[0];
    }
    if (valid && bad && a.info) raise_info(a.info);
}

}  // namespace mf
