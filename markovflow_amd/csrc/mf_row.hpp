// Row kernels: ONE 16-lane DPP row = one (series, time-chunk) sub-problem, for state dimensions whose elimination state does
// not fit the registers of a single lane (7 <= d <= 15).
//
// Lane r < D of a row holds row r (or column r) of every D x D matrix of the step in D registers; lane D holds the vector
// quantities (right-hand side, offsets, observations); lanes D+1..15 idle.  The only cross-lane primitive is the DPP
// `row_newbcast:K` operand of a fused multiply-add - gfx950's single DPP control for 64-bit operations, issued at the rate of the
// plain v_fmac_f64 (scripts/micro/dpp_f64_rate.hip) - so a D x D x D product is D*D wave instructions for FOUR chunks, no LDS,
// no moves, no barriers, and the whole state of a chunk is ~70 registers per lane: three wavefronts per SIMD instead of the
// half wavefront per SIMD of the spike-in-LDS kernels (mf_kf_x.hpp), which this replaces for the log-likelihood.
//
// With "row layout" R(M) (lane i holds M[i][:]) and "column layout" C(M) = R(M^T), one instruction
//     acc_j += bcast_K(src_q) * own_q
// gives, from row layouts of P and Q:   (P Q)[i][j]   = sum_k own P_k * bcast_k(Q_j)
//                                       (P Q^T)[i][j] = sum_k own P_k * bcast_j(Q_k)
// and an in-lane forward substitution with the factor's elements broadcast (L[i][k] = bcast_i(L_k)) solves L x = (own column).
// The step of the partitioned elimination (same math and chunk convention as kf_chunk_x_kernel, same RedSys output) is arranged
// so that every operand is available in the layout its consumer needs without a transposition:
//     C(Ba) = C^-1 [A | mvec]   columns of B = C^-1 A in lanes < D, w = C^-1 mvec in lane D      (substitution)
//     C(Ci) = C^-1 [I | mvec]                                                                    (substitution)
//     R(S)  = -Ci^T B  in lanes < D,   t - B^T w in lane D                                       (P Q^T form)
//     R(Phi) += B^T B                                                                           (P Q^T form) -> Cholesky in place
//     C(V)  = L^-1 X,  R(W) = S L^-T in lanes < D and z = L^-1 t in lane D                        (substitutions)
//     GU -= V^T V, gU -= V^T z, X' = -W V, Phi' = Ci^T Ci - W W^T + H^T R^-1 H                    (P Q^T / P Q forms)
// Lane D takes part in the SAME instructions with the vector quantities in the registers where the other lanes keep a matrix
// row: its row of Phi' is the next right-hand side (Ci^T w - W z + H^T R^-1 y), for free.
// The op sequence was first validated register by register on the CPU (scripts/row_sim.py).
//
// Hazards: a DPP operand must not have been written by a VALU instruction in the two preceding issue slots, and hipcc does not
// see inside asm statements.  Every DPP source that was just written goes through fence(): it ties the registers (so the
// compiler cannot sink their definitions below it) and issues `s_nop 1`.  scripts/check_dpp_hazards.py scans the final ISA.
#pragma once
#include <utility>

#include "mf_kernels.hpp"

namespace mf {
namespace row {

template <int N, typename F, int... I> MF_DEV void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, i>) for i = 0..N-1, unrolled at compile time (the DPP lane select is an immediate)
template <int N, typename F> MF_DEV void sfor(F&& f) {
    if constexpr (N > 0) sfor_impl<N>(f, std::make_integer_sequence<int, N>{});
}
template <int A, int B, typename F> MF_DEV void sfor2(F&& f) {
    sfor<B - A>([&](auto i) { f(std::integral_constant<int, A + decltype(i)::value>{}); });
}

template <typename T> struct Dpp;
template <> struct Dpp<double> {
    // acc += bcast_K(src) * own
    template <int K> static MF_DEV void fmac(double& acc, const double& src, const double& own) {
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(K));
    }
    // acc -= bcast_K(src) * own
    template <int K> static MF_DEV void fnmac(double& acc, const double& src, const double& own) {
        asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(K));
    }
    template <int K> static MF_DEV double bcast(const double& src) {
        double r;
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(K));
        return r;
    }
};
template <> struct Dpp<float> {
    template <int K> static MF_DEV void fmac(float& acc, const float& src, const float& own) {
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(K));
    }
    template <int K> static MF_DEV void fnmac(float& acc, const float& src, const float& own) {
        asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(K));
    }
    template <int K> static MF_DEV float bcast(const float& src) {
        float r;
        asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(K));
        return r;
    }
};

// 1 / sqrt(x): hardware seed (relative error <= 2^-22) and ONE third-order correction r (1 + e/2 + 3e^2/8), e = 1 - x r^2:
// the truncation error 5 e^3 / 16 is below 2^-67, six instructions instead of the nine of two Newton steps (t_rsqrt) on the
// critical path of every pivot - here a pivot's chain serves four chunks, not sixty-four.
MF_DEV double row_rsqrt(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * r), r, 1.0);
    const double q = e * __builtin_fma(e, 0.375, 0.5);
    return __builtin_fma(r, q, r);
}
MF_DEV float row_rsqrt(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    const float e = __builtin_fmaf(-(x * r), r, 1.0f);
    return __builtin_fmaf(r, 0.5f * e, r);
}

// The registers of v are DPP sources from here on: their definitions stay above, two wait states follow.
template <typename T> MF_DEV void tie(T& x) { asm volatile("" : "+v"(x)); }      // (asm statements cannot sit in a lambda: host pass)
template <typename T, int N> MF_DEV void fence(T (&v)[N]) {
    sfor<N>([&](auto i) { tie(v[decltype(i)::value]); });
    asm volatile("s_nop 1");
}
template <typename T> MF_DEV void fence1(T& v) { asm volatile("s_nop 1" : "+v"(v)); }

// ---- buffer loads: one descriptor per tensor and wave, 32-bit byte offsets per lane, out-of-range lanes read zero ----------
constexpr unsigned ROW_INVALID = 0x80000000u;     // added to a lane's offset: lands beyond any descriptor's range
constexpr unsigned long long ROW_MAXREC = 0x80000000ull;
// largest wave-relative byte offset a launch may produce (checked by the launcher: row_offsets_fit)
constexpr unsigned long long ROW_MAXOFF = 0x70000000ull;

typedef unsigned int row_u2 __attribute__((ext_vector_type(2)));

template <typename T> MF_DEV __amdgpu_buffer_rsrc_t make_rsrc(const T* base, const T* end) {
    unsigned long long bytes = base && end > base ? (unsigned long long)((const char*)end - (const char*)base) : 0ull;
    if (bytes > ROW_MAXREC) bytes = ROW_MAXREC;
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(unsigned)bytes, 0x00020000);
}
template <typename T> MF_DEV T bload(__amdgpu_buffer_rsrc_t rs, unsigned off);
template <> MF_DEV double bload<double>(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0));
}
template <> MF_DEV float bload<float>(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
}

// ---- the elimination state of one chunk, spread over the lanes of a row -------------------------------------------------------
template <typename T, int D, int M> struct RowChunk {
    using P = Dpp<T>;
    T Phi[D];      // lanes < D: row r of the current block's pivot (full symmetric row); lane D: its right-hand side
    T Xa[D];       // lanes < D: column r of X, the coupling of the current block to the chunk's left separator
    T GU[D];       // lanes < D: row r of the separator's accumulated pivot contribution
    T gU;          // lane r < D: element r of the separator's accumulated right-hand side contribution
    T quad, ww, yry;   // meaningful in lane D: sum |z|^2, sum |C^-1 mvec|^2, sum y^T R^-1 y
    T Id[D];       // Id[i] = (r == i)
    T ED;          // (r == D)
    LogAcc<T> laC; // lanes < D: prod of own diagonal elements of the transition factors
    LogAcc<T> laL; // replicated: prod of the pivots = prod diag(L)^2
    bool bad;

    MF_DEV void init(int r) {
        sfor<D>([&](auto i) {
            constexpr int ii = decltype(i)::value;
            Phi[ii] = T(0); Xa[ii] = T(0); GU[ii] = T(0);
            Id[ii] = r == ii ? T(1) : T(0);
        });
        ED = r == D ? T(1) : T(0);
        gU = quad = ww = yry = T(0);
        laC.init();
        laL.init();
        bad = false;
    }

    // Loop order of every product below: the INNER loop runs over independent accumulators (outer-product form), so no
    // instruction depends on its predecessor - and hipcc, which pads a wait state between two asm statements that touch
    // the same register (it cannot see what they are), has nothing to pad.
    //
    // Ba = C^-1 [A | mvec], CiT = C^-1 [I | mvec] as columns (right-looking substitution, both systems interleaved).
    // In: Crow (lane r: row r of C), cdiag (own diagonal element), Aa (lane r < D: column r of A, lanes >= D: zero),
    // bd (lane i < D: mvec[i]).  Aa is consumed.
    MF_DEV void whiten(T (&Crow)[D], T cdiag, T (&Aa)[D], T bd, T (&Ba)[D], T (&CiT)[D]) {
        fence(Crow);
        bad |= !(cdiag != T(0));
        T dinv = t_rcp<T>(cdiag);
        laC.mul(cdiag);
        laC.renorm();
        fence1(dinv);
        fence1(bd);
        sfor<D>([&](auto i) { P::template fmac<decltype(i)::value>(Aa[decltype(i)::value], bd, ED); });   // mvec into lane D
        T accC[D];
        sfor<D>([&](auto i) { accC[decltype(i)::value] = __builtin_fma(Aa[decltype(i)::value], ED, Id[decltype(i)::value]); });
        sfor<D>([&](auto k) {
            constexpr int kk = decltype(k)::value;
            const T dk = P::template bcast<kk>(dinv);          // 1 / C[k][k]
            CiT[kk] = accC[kk] * dk;
            Ba[kk] = Aa[kk] * dk;
            sfor2<kk + 1, D>([&](auto i) {
                constexpr int ii = decltype(i)::value;
                P::template fnmac<ii>(accC[ii], Crow[kk], CiT[kk]);
                P::template fnmac<ii>(Aa[ii], Crow[kk], Ba[kk]);
            });
        });
        sfor<D>([&](auto k) { ww = __builtin_fma(Ba[decltype(k)::value], Ba[decltype(k)::value], ww); });
    }

    // Phi <- Ci^T Ci [- W W^T] + H^T R^-1 H (lanes < D), and in lane D: Ci^T w [- W z] + H^T R^-1 y.
    // Ha: lane r < D: column r of H_k (M values), lanes >= D zero; yd: lane o < M: y_k[o]; Ri: observation precision (row-major).
    template <bool HASW, typename RI> MF_DEV void new_pivot(T (&CiT)[D], T (&W)[D], T (&Ha)[M], T yd, const RI& Ri) {
        fence1(yd);
        sfor<M>([&](auto o) { P::template fmac<decltype(o)::value>(Ha[decltype(o)::value], yd, ED); });       // y into lane D
        T Pn[D];
        sfor<D>([&](auto j) { Pn[decltype(j)::value] = T(0); });
        fence(CiT);
        sfor<D>([&](auto k) {                        // Ci[k][j] = 0 for k < j: exact zeros, skipped
            constexpr int kk = decltype(k)::value;
            sfor<kk + 1>([&](auto j) { P::template fmac<decltype(j)::value>(Pn[decltype(j)::value], CiT[kk], CiT[kk]); });
        });
        if constexpr (HASW) {
            fence(W);
            sfor<D>([&](auto k) {
                constexpr int kk = decltype(k)::value;
                sfor<D>([&](auto j) { P::template fnmac<decltype(j)::value>(Pn[decltype(j)::value], W[kk], W[kk]); });
            });
        }
        T u[M];
        sfor<M>([&](auto o) {
            constexpr int oo = decltype(o)::value;
            u[oo] = Ri(oo, 0) * Ha[0];
            sfor2<1, M>([&](auto p) { u[oo] = __builtin_fma(Ri(oo, decltype(p)::value), Ha[decltype(p)::value], u[oo]); });
        });
        sfor<M>([&](auto o) { yry = __builtin_fma(Ha[decltype(o)::value], u[decltype(o)::value], yry); });
        fence(Ha);
        sfor<M>([&](auto o) {
            constexpr int oo = decltype(o)::value;
            sfor<D>([&](auto j) { P::template fmac<decltype(j)::value>(Pn[decltype(j)::value], Ha[oo], u[oo]); });
        });
        sfor<D>([&](auto j) { Phi[decltype(j)::value] = Pn[decltype(j)::value]; });
    }

    // first block of the chunk: block 0 of the series (Aa = 0) or the block after the chunk's left separator
    template <typename RI>
    MF_DEV void start(T (&Crow)[D], T cdiag, T (&Aa)[D], T bd, T (&Ha)[M], T yd, const RI& Ri) {
        T Ba[D], CiT[D];
        whiten(Crow, cdiag, Aa, bd, Ba, CiT);
        fence(Ba);
        fence(CiT);
        gU = T(0);
        sfor<D>([&](auto i) { GU[decltype(i)::value] = T(0); Xa[decltype(i)::value] = T(0); });
        sfor<D>([&](auto k) {                       // GU = B^T B, gU = -B^T w, X = -Ci^T B (columns)
            constexpr int kk = decltype(k)::value;
            sfor<D>([&](auto j) {
                constexpr int jj = decltype(j)::value;
                P::template fmac<jj>(GU[jj], Ba[kk], Ba[kk]);
                P::template fnmac<jj>(Xa[jj], CiT[kk], Ba[kk]);
            });
            P::template fnmac<D>(gU, Ba[kk], Ba[kk]);
        });
        T Wdummy[D];
        new_pivot<false>(CiT, Wdummy, Ha, yd, Ri);
    }

    // one interior step: completes and eliminates the previous block, forms the pivot of this one
    template <typename RI>
    MF_DEV void step(T (&Crow)[D], T cdiag, T (&Aa)[D], T bd, T (&Ha)[M], T yd, const RI& Ri) {
        T Ba[D], CiT[D];
        whiten(Crow, cdiag, Aa, bd, Ba, CiT);
        fence(Ba);
        // S rows (lanes < D) with the right-hand side row t - B^T w in lane D; pivot of the previous block += B^T B
        T S[D];
        sfor<D>([&](auto j) { S[decltype(j)::value] = Phi[decltype(j)::value] * ED; });
        sfor<D>([&](auto k) {
            constexpr int kk = decltype(k)::value;
            sfor<D>([&](auto j) {
                constexpr int jj = decltype(j)::value;
                P::template fnmac<jj>(S[jj], Ba[kk], CiT[kk]);
                P::template fmac<jj>(Phi[jj], Ba[kk], Ba[kk]);
            });
        });
        // Right-looking Cholesky in place; the substitutions V = L^-1 X (columns, in place) and W = S L^-T (rows; lane D:
        // z = L^-1 t) ride along column by column.  The pivot is broadcast and every lane takes the reciprocal square root.
        T W[D];
        sfor<D>([&](auto j) {
            constexpr int jj = decltype(j)::value;
            fence1(Phi[jj]);
            const T s = P::template bcast<jj>(Phi[jj]);
            bad |= !(s > T(0));
            const T inv = row_rsqrt(s);
            laL.mul(s);                                   // the pivots themselves: log|L| = log(prod) / 2
            Phi[jj] *= inv;
            Xa[jj] *= inv;
            W[jj] = S[jj] * inv;
            fence1(Phi[jj]);
            sfor2<jj + 1, D>([&](auto k) {
                constexpr int kk = decltype(k)::value;
                P::template fnmac<kk>(Phi[kk], Phi[jj], Phi[jj]);
                P::template fnmac<kk>(Xa[kk], Phi[jj], Xa[jj]);
                P::template fnmac<kk>(S[kk], Phi[jj], W[jj]);
            });
        });
        laL.renorm();
        sfor<D>([&](auto k) { quad = __builtin_fma(W[decltype(k)::value], W[decltype(k)::value], quad); });
        // separator: GU -= V^T V, gU -= V^T z;  X' = -W V
        fence(Xa);
        fence(W);
        T Xn[D];
        sfor<D>([&](auto i) { Xn[decltype(i)::value] = T(0); });
        sfor<D>([&](auto k) {
            constexpr int kk = decltype(k)::value;
            sfor<D>([&](auto j) {
                constexpr int jj = decltype(j)::value;
                P::template fnmac<jj>(GU[jj], Xa[kk], Xa[kk]);
                P::template fnmac<jj>(Xn[jj], W[kk], Xa[kk]);
            });
            P::template fnmac<D>(gU, W[kk], Xa[kk]);
        });
        sfor<D>([&](auto i) { Xa[decltype(i)::value] = Xn[decltype(i)::value]; });
        new_pivot<true>(CiT, W, Ha, yd, Ri);
    }
};


// observation precision: shared by all steps (read once, made wave-uniform: the compiler keeps it in SGPRs) or per step
// (loaded into registers with the step's other loads)
MF_DEV double to_uniform(double x) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
MF_DEV float to_uniform(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}
template <typename T, int M> struct RiRegs {
    T v[M * M];
    MF_DEV T operator()(int o, int q) const { return v[o * M + q]; }
};

// Wavefronts per SIMD the row kernels are compiled for: a row-distributed matrix costs d registers (2 d in fp64) per lane, the
// elimination state ~ 9 matrices: three waves up to d = 9 (168 registers), two up to d = 12 (256), one beyond (512 with AGPRs).
constexpr int row_waves_per_simd(int d) { return d <= 9 ? 3 : d <= 12 ? 2 : 1; }

// True when every wave-relative byte offset of a launch stays below ROW_MAXOFF (a wave's four chunks span at most
// ceil(4 / P) + 1 series).
inline bool row_offsets_fit(long Tn, long P, int D, int M, int elem) {
    const unsigned long long span = (unsigned long long)((4 + P - 1) / P + 1);
    const unsigned long long per_series = (unsigned long long)Tn * (unsigned long long)(D * D > M * D ? D * D : M * D) * elem;
    return span * per_series < ROW_MAXOFF;
}

// Level 0 of the log-likelihood, chunk convention and output of kf_chunk_x_kernel: chunk c owns blocks [c T / P, (c+1) T / P)
// and leaves its last block as a reduced block.  64 threads = 4 rows = 4 chunks.
template <typename T, int D, int M, bool RSTEP>
__global__ void __launch_bounds__(64, row_waves_per_simd(D)) kf_row_kernel(KfArgs<T> a, RedSys<T> out) {
    static_assert(D + 1 <= 16 && M <= D, "one row of 16 lanes per chunk");
    constexpr unsigned S = sizeof(T);
    const int lane = threadIdx.x;
    const int r = lane & 15;
    const int rcl = r < D ? r : D - 1;
    const long total = a.B * a.P;
    const long id_raw = (long)blockIdx.x * 4 + (lane >> 4);
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;
    const long s = id / a.P, c = id % a.P;
    const long k0 = (c * a.Tn) / a.P, k1 = ((c + 1) * a.Tn) / a.P;
    const long id0 = (long)blockIdx.x * 4 < total ? (long)blockIdx.x * 4 : total - 1;
    const long s0 = id0 / a.P;                                        // first series of this wave: wave-uniform
    const long nt = a.Tn - 1;
    const unsigned ds = (unsigned)(s - s0);

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.A + s0 * nt * D * D, a.A + a.B * nt * D * D);
    const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(a.cholQ + s0 * nt * D * D, a.cholQ + a.B * nt * D * D);
    const __amdgpu_buffer_rsrc_t rsb = make_rsrc(a.b + s0 * nt * D, a.b + a.B * nt * D);
    const __amdgpu_buffer_rsrc_t rsH = make_rsrc(a.H + s0 * a.Tn * M * D, a.H + a.B * a.Tn * M * D);
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(a.y + s0 * a.Tn * M, a.y + a.B * a.Tn * M);
    const __amdgpu_buffer_rsrc_t rsP = make_rsrc(a.cholP0 + s0 * D * D, a.cholP0 + a.B * D * D);
    const __amdgpu_buffer_rsrc_t rsm = make_rsrc(a.mu0 + s0 * D, a.mu0 + a.B * D);
    const __amdgpu_buffer_rsrc_t rsR =
        RSTEP ? make_rsrc(a.Rinv + s0 * a.Tn * M * M, a.Rinv + a.B * a.Tn * M * M) : make_rsrc(a.Rinv, a.Rinv + M * M);

    // per-lane parts of the offsets
    const unsigned lrow = rcl * D * S;                                // own row of a D x D block
    const unsigned ldiag = rcl * (D + 1) * S;                         // own diagonal element
    const unsigned lcol = r < D ? r * S : ROW_INVALID;                // own column (lanes >= D read zero)
    const unsigned lvec = r < D ? r * S : ROW_INVALID;                // own element of a D-vector
    const unsigned lobs = r < M ? r * S : ROW_INVALID;                // own element of an M-vector

    RowChunk<T, D, M> E;
    E.init(r);

    auto load_obs = [&](long k, T (&Ha)[M], T& yd) {
        const unsigned ob = (ds * (unsigned)a.Tn + (unsigned)k) * (M * D * S);
        sfor<M>([&](auto o) { Ha[decltype(o)::value] = bload<T>(rsH, ob + decltype(o)::value * D * S + lcol); });
        yd = bload<T>(rsy, (ds * (unsigned)a.Tn + (unsigned)k) * (M * S) + lobs);
    };
    auto load_rinv = [&](long k, RiRegs<T, M>& R) {
        const unsigned rb = (ds * (unsigned)a.Tn + (unsigned)k) * (M * M * S);
        sfor<M * M>([&](auto e) { R.v[decltype(e)::value] = bload<T>(rsR, rb + decltype(e)::value * S); });
    };
    RiRegs<T, M> rshared;
    if constexpr (!RSTEP) sfor<M * M>([&](auto e) { rshared.v[decltype(e)::value] = to_uniform(a.Rinv[decltype(e)::value]); });

    // ---- first block of the chunk: its transition (or the prior of block 0) with pointer selects, no branches ----
    {
        T Crow[D], Aa[D], Ha[M], cdiag, bd, yd;
        const bool first = k0 == 0;
        const unsigned tb = (ds * (unsigned)nt + (unsigned)(first ? 0 : k0 - 1)) * (D * D * S);
        const unsigned tvb = (ds * (unsigned)nt + (unsigned)(first ? 0 : k0 - 1)) * (D * S);
        const unsigned qsel = first ? ROW_INVALID : 0u, psel = first ? 0u : ROW_INVALID;
        sfor<D>([&](auto k) {
            constexpr int kk = decltype(k)::value;
            Crow[kk] = bload<T>(rsQ, tb + lrow + kk * S + qsel) + bload<T>(rsP, ds * (D * D * S) + lrow + kk * S + psel);
            Aa[kk] = bload<T>(rsA, tb + kk * D * S + (lcol | qsel));
        });
        cdiag = bload<T>(rsQ, tb + ldiag + qsel) + bload<T>(rsP, ds * (D * D * S) + ldiag + psel);
        bd = bload<T>(rsb, tvb + (lvec | qsel)) + bload<T>(rsm, ds * (D * S) + (lvec | psel));
        load_obs(k0, Ha, yd);
        if constexpr (RSTEP) {
            RiRegs<T, M> R;
            load_rinv(k0, R);
            E.start(Crow, cdiag, Aa, bd, Ha, yd, R);
        } else {
            E.start(Crow, cdiag, Aa, bd, Ha, yd, rshared);
        }
    }
    // ---- interior blocks ----
    unsigned tb = (ds * (unsigned)nt + (unsigned)k0) * (D * D * S);   // transition k0 -> k0 + 1
    unsigned tvb = (ds * (unsigned)nt + (unsigned)k0) * (D * S);
    for (long k = k0 + 1; k < k1; ++k) {
        asm volatile("s_nop 4");                                      // EXEC may have changed at the loop head: DPP needs 5 wait states
        T Crow[D], Aa[D], Ha[M], cdiag, bd, yd;
        sfor<D>([&](auto q) {
            constexpr int kk = decltype(q)::value;
            Crow[kk] = bload<T>(rsQ, tb + lrow + kk * S);
            Aa[kk] = bload<T>(rsA, tb + kk * D * S + lcol);
        });
        cdiag = bload<T>(rsQ, tb + ldiag);
        bd = bload<T>(rsb, tvb + lvec);
        load_obs(k, Ha, yd);
        if constexpr (RSTEP) {
            RiRegs<T, M> R;
            load_rinv(k, R);
            E.step(Crow, cdiag, Aa, bd, Ha, yd, R);
        } else {
            E.step(Crow, cdiag, Aa, bd, Ha, yd, rshared);
        }
        tb += D * D * S;
        tvb += D * S;
    }
    // ---- the chunk's reduced block ----
    using P = Dpp<T>;
    T logc = E.laC.value(), qd = E.quad, w2 = E.ww, yr = E.yry;
    fence1(logc); fence1(qd); fence1(w2); fence1(yr);
    T lc = T(0);
    sfor<D>([&](auto i) { lc += P::template bcast<decltype(i)::value>(logc); });
    const T scalar = T(-0.5) * (P::template bcast<D>(yr) + P::template bcast<D>(w2)) + T(0.5) * P::template bcast<D>(qd) - lc -
                     T(0.5) * E.laL.value();
    if (valid) {
        if (r < D) {
            T* dv = out.Dv + id * D * D + r * D;
            T* gu = out.GU + id * D * D + r * D;
            T* f = out.F + id * D * D + r;
            sfor<D>([&](auto j) {
                constexpr int jj = decltype(j)::value;
                dv[jj] = E.Phi[jj];
                gu[jj] = E.GU[jj];
                f[jj * D] = E.Xa[jj];
            });
            out.gU[id * D + r] = E.gU;
            if (r == 0) out.sc[id] = scalar;
        } else if (r == D) {
            T* tv = out.tv + id * D;
            sfor<D>([&](auto j) { tv[decltype(j)::value] = E.Phi[decltype(j)::value]; });
        }
        if (E.bad && a.info) raise_info(a.info);
    }
}

// ---- reduction levels of the log-likelihood in row form -----------------------------------------------------------------------
// The reduced block-tridiagonal system a level leaves (RedSys: pivot parts Dv, contributions GU / gU to the block on the left,
// couplings F, right-hand sides tv, scalars) is reduced again by the same partitioned elimination; a row per (series, chunk).
// Same conventions as red_chunk_kernel / red_final_kernel (mf_kernels.hpp): block j's pivot is Dv_j + GU_{j+1}, its right-hand
// side tv_j + gU_{j+1}, F_j couples block j (rows) with block j - 1 (columns).
template <typename T, int D> struct RowElim {
    using P = Dpp<T>;
    T Phi[D];      // lanes < D: row r of the current pivot; lane D: its right-hand side
    T Xa[D];       // lanes < D: column r of the coupling to the chunk's left separator
    T GU[D];       // lanes < D: row r of the separator's accumulated pivot contribution
    T gU, quad;
    LogAcc<T> laL;
    bool bad;

    MF_DEV void init() {
        sfor<D>([&](auto i) { Phi[decltype(i)::value] = T(0); Xa[decltype(i)::value] = T(0); GU[decltype(i)::value] = T(0); });
        gU = quad = T(0);
        laL.init();
        bad = false;
    }
    // Eliminates the current block and moves to the next one.  S: rows of the coupling F (lanes < D; lane D: anything),
    // Dn: rows of the next pivot (lane D: the next right-hand side).  Lane D's S row becomes the current right-hand side, so
    // z = L^-1 t comes out of the same substitution as W = F L^-T.
    template <bool SPIKE> MF_DEV void advance(T (&S)[D], const T (&Dn)[D], bool is_vec) {
        sfor<D>([&](auto j) { S[decltype(j)::value] = is_vec ? Phi[decltype(j)::value] : S[decltype(j)::value]; });
        T W[D];
        sfor<D>([&](auto j) {
            constexpr int jj = decltype(j)::value;
            fence1(Phi[jj]);
            const T s = P::template bcast<jj>(Phi[jj]);
            bad |= !(s > T(0));
            const T inv = row_rsqrt(s);
            laL.mul(s);                                   // the pivots themselves: log|L| = log(prod) / 2
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
        laL.renorm();
        sfor<D>([&](auto k) { quad = __builtin_fma(W[decltype(k)::value], W[decltype(k)::value], quad); });
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
                P::template fnmac<D>(gU, W[kk], Xa[kk]);
            });
            sfor<D>([&](auto i) { Xa[decltype(i)::value] = Xn[decltype(i)::value]; });
        }
        sfor<D>([&](auto j) { Phi[decltype(j)::value] = Dn[decltype(j)::value]; });
        sfor<D>([&](auto k) {
            constexpr int kk = decltype(k)::value;
            sfor<D>([&](auto j) { P::template fnmac<decltype(j)::value>(Phi[decltype(j)::value], W[kk], W[kk]); });
        });
    }
    // The last block of a series: Cholesky with the right-hand side as one more row (lane D), so that its row of the
    // factor is z; quad and the log-determinant complete.
    MF_DEV void finish() {
        sfor<D>([&](auto j) {
            constexpr int jj = decltype(j)::value;
            fence1(Phi[jj]);
            const T s = P::template bcast<jj>(Phi[jj]);
            bad |= !(s > T(0));
            const T inv = row_rsqrt(s);
            laL.mul(s);                                   // the pivots themselves: log|L| = log(prod) / 2
            Phi[jj] *= inv;
            fence1(Phi[jj]);
            sfor2<jj + 1, D>([&](auto k) { P::template fnmac<decltype(k)::value>(Phi[decltype(k)::value], Phi[jj], Phi[jj]); });
        });
        laL.renorm();
        sfor<D>([&](auto k) { quad = __builtin_fma(Phi[decltype(k)::value], Phi[decltype(k)::value], quad); });   // lane D: |z|^2
    }
};

// Loads of one reduced block for the lanes of a row: pivot rows (lane D: the right-hand side) with the contribution of the
// next block's interior folded in.  Plain global loads through per-lane pointers: the four tensors are different allocations
// of one workspace (or user tensors), the lanes < D read matrix rows and lane D reads vectors.
template <typename T, int D>
MF_DEV void load_red_rows(const RedSys<T>& in, long s, long k, int r, T (&Dn)[D], T& sc) {
    const long idx = s * in.n + k;
    const bool vec = r >= D;
    const int rc = r < D ? r : 0;
    const T* p = vec ? in.tv + idx * D : in.Dv + idx * D * D + rc * D;
    const bool fold = in.GU != nullptr && k + 1 < in.n;
    const long idn = fold ? idx + 1 : idx;
    const T* g = in.GU == nullptr ? p : (vec ? in.gU + idn * D : in.GU + idn * D * D + rc * D);
    const T f = fold ? T(1) : T(0);
    sfor<D>([&](auto j) { Dn[decltype(j)::value] = __builtin_fma(g[decltype(j)::value], f, p[decltype(j)::value]); });
    sc = in.sc ? in.sc[idx] : T(0);
}

// One level: RedSys(n) -> RedSys(P); 64 threads = 4 rows = 4 (series, chunk) pairs; chunk c owns blocks [c n / P, (c+1) n / P).
template <typename T, int D>
__global__ void __launch_bounds__(64, row_waves_per_simd(D)) red_row_kernel(RedSys<T> in, RedSys<T> out, long B, long Pn, int* info) {
    const int lane = threadIdx.x, r = lane & 15;
    const int rc = r < D ? r : 0;
    const long total = B * Pn;
    const long id_raw = (long)blockIdx.x * 4 + (lane >> 4);
    const bool valid = id_raw < total;
    const long id = valid ? id_raw : total - 1;
    const long s = id / Pn, c = id % Pn;
    const long k0 = (c * in.n) / Pn, k1 = ((c + 1) * in.n) / Pn;
    RowElim<T, D> E;
    E.init();
    T acc_sc;
    {
        T Dn[D];
        load_red_rows<T, D>(in, s, k0, r, Dn, acc_sc);
        // coupling of the chunk's first block to the separator on its left, as columns (zero for the first chunk of a series)
        const long kf = k0 > 0 ? k0 : (in.n > 1 ? 1 : 0);
        const T* fcol = in.F + (s * in.f_stride + kf + in.f_off) * D * D + rc;
        const T keep = (k0 > 0 && r < D) ? T(1) : T(0);
        sfor<D>([&](auto i) {
            E.Phi[decltype(i)::value] = Dn[decltype(i)::value];
            E.Xa[decltype(i)::value] = in.n > 1 ? fcol[decltype(i)::value * D] * keep : T(0);
        });
    }
    for (long k = k0 + 1; k < k1; ++k) {
        asm volatile("s_nop 4");
        T Dn[D], S[D], sc;
        load_red_rows<T, D>(in, s, k, r, Dn, sc);
        const T* frow = in.F + (s * in.f_stride + k + in.f_off) * D * D + rc * D;
        sfor<D>([&](auto j) { S[decltype(j)::value] = frow[decltype(j)::value]; });
        acc_sc += sc;
        E.template advance<true>(S, Dn, r == D);
    }
    if (valid) {
        if (r < D) {
            T* dv = out.Dv + id * D * D + r * D;
            T* gu = out.GU + id * D * D + r * D;
            T* f = out.F + id * D * D + r;
            sfor<D>([&](auto j) {
                constexpr int jj = decltype(j)::value;
                dv[jj] = E.Phi[jj];
                gu[jj] = E.GU[jj];
                f[jj * D] = E.Xa[jj];
            });
            out.gU[id * D + r] = E.gU;
        } else if (r == D) {
            T* tv = out.tv + id * D;
            sfor<D>([&](auto j) { tv[decltype(j)::value] = E.Phi[decltype(j)::value]; });
            out.sc[id] = acc_sc + T(0.5) * (E.quad - E.laL.value());        // lane D holds |z|^2
        }
        if (E.bad && info) raise_info(info);
    }
}

// Last level: a row per series walks the remaining blocks; out[s] = add_const + scalars + 0.5 |z|^2 - log|L|.
template <typename T, int D>
__global__ void __launch_bounds__(64, row_waves_per_simd(D)) red_row_final_kernel(RedSys<T> in, long B, T add_const, T* __restrict__ out, int* info) {
    const int lane = threadIdx.x, r = lane & 15;
    const int rc = r < D ? r : 0;
    const long id_raw = (long)blockIdx.x * 4 + (lane >> 4);
    const bool valid = id_raw < B;
    const long s = valid ? id_raw : B - 1;
    RowElim<T, D> E;
    E.init();
    T acc_sc;
    {
        T Dn[D];
        load_red_rows<T, D>(in, s, 0, r, Dn, acc_sc);
        sfor<D>([&](auto i) { E.Phi[decltype(i)::value] = Dn[decltype(i)::value]; });
    }
    for (long k = 1; k < in.n; ++k) {
        T Dn[D], S[D], sc;
        load_red_rows<T, D>(in, s, k, r, Dn, sc);
        const T* frow = in.F + (s * in.f_stride + k + in.f_off) * D * D + rc * D;
        sfor<D>([&](auto j) { S[decltype(j)::value] = frow[decltype(j)::value]; });
        acc_sc += sc;
        E.template advance<false>(S, Dn, r == D);
    }
    E.finish();
    if (valid && r == D) out[s] = add_const + acc_sc + T(0.5) * (E.quad - E.laL.value());
    if (valid && E.bad && info) raise_info(info);
}

}   // namespace row
}   // namespace mf
