// Body of mf_big.hpp, included once per scalar type (MF_BIG_T = float / double, MF_BIG_NS = big / bigd).
namespace mf {
namespace MF_BIG_NS {

typedef MF_BIG_T real;

typedef real real4 __attribute__((ext_vector_type(4)));
constexpr int NTHR = 256;
constexpr int MAXM_BIG = 32;   // observation dimension supported by the large-d path

template <int DP> struct Geo {
    static constexpr int LD = DP + 4;
    static constexpr int NT = DP / 16;
    static constexpr int TILE = DP * LD;          // floats
};

using namespace bigcommon;
__device__ __forceinline__ constexpr int acc_row(int q, int e) { return sizeof(real) == 4 ? 4 * q + e : q + 4 * e; }

enum KMode { K_FULL = 0, K_A_LOWER = 1, K_A_UPPER = 2, K_B_LOWER = 3, K_B_UPPER = 4 };
enum OMode { O_FULL = 0, O_LOWER = 1 };

// C (=, +=) alpha * opA(A) * opB(B) on DP x DP LDS images.  TA/TB = 1 use the transpose.  KM names a triangular
// operand (in its op() form) so that all-zero 16 x 16 K-tiles are skipped; kt_end limits the K tiles (for K < DP).
// Every wave owns whole output tiles; no barrier inside.
template <int DP, int TA, int TB, int ACC, int KM, int OM>
__device__ __forceinline__ void gemm(const real* __restrict__ A, const real* __restrict__ B, real* __restrict__ C,
                                     real alpha, int kt_end = Geo<DP>::NT) {
#ifdef MF_BIG_SKIP_GEMM
    return;
#endif
    constexpr int LD = Geo<DP>::LD, NT = Geo<DP>::NT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    if constexpr (NT >= 3) {
        // one COLUMN of output tiles per wave: the B fragment of a K-tile is read once for all of them and the NT
        // accumulators give the matrix pipe independent chains (a single 16x16x4 chain is latency-bound)
        const int tj = wave;
        if (tj >= NT) return;
        real4 acc[NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) acc[ti] = real4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < kt_end; ++kt) {
            if (KM == K_B_LOWER && kt < tj) continue;
            if (KM == K_B_UPPER && kt > tj) continue;
            const int kb = 16 * kt + 4 * q;
            real bv[4], av[NT][4];
            if (TB) {
                const real4 t4 = *reinterpret_cast<const real4*>(B + (16 * tj + r) * LD + kb);
                bv[0] = t4[0]; bv[1] = t4[1]; bv[2] = t4[2]; bv[3] = t4[3];
            } else {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) bv[kk] = B[(kb + kk) * LD + 16 * tj + r];
            }
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                const bool on = !(OM == O_LOWER && tj > ti) && !(KM == K_A_LOWER && kt > ti) && !(KM == K_A_UPPER && kt < ti);
                if (on) {
                    if (TA) {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) av[ti][kk] = A[(kb + kk) * LD + 16 * ti + r];
                    } else {
                        const real4 t4 = *reinterpret_cast<const real4*>(A + (16 * ti + r) * LD + kb);
                        av[ti][0] = t4[0]; av[ti][1] = t4[1]; av[ti][2] = t4[2]; av[ti][3] = t4[3];
                    }
                } else {
                    av[ti][0] = av[ti][1] = av[ti][2] = av[ti][3] = 0.f;
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) {
                    const bool on = !(OM == O_LOWER && tj > ti) && !(KM == K_A_LOWER && kt > ti) && !(KM == K_A_UPPER && kt < ti);
                    if (on) acc[ti] = mfma(av[ti][kk], bv[kk], acc[ti]);
                }
            }
        }
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            if (OM == O_LOWER && tj > ti) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                real* p = C + (16 * ti + acc_row(q, e)) * LD + 16 * tj + r;
                if (ACC) *p += alpha * acc[ti][e]; else *p = alpha * acc[ti][e];
            }
        }
        return;
    }
    for (int idx = wave; idx < NT * NT; idx += NTHR / 64) {
        const int ti = idx / NT, tj = idx % NT;
        if (OM == O_LOWER && tj > ti) continue;
        int k0 = 0, k1 = kt_end;
        if (KM == K_A_LOWER) k1 = min(k1, ti + 1);
        if (KM == K_A_UPPER) k0 = ti;
        if (KM == K_B_LOWER) k0 = tj;
        if (KM == K_B_UPPER) k1 = min(k1, tj + 1);
        real4 acc = {0.f, 0.f, 0.f, 0.f};
        // the sum over k may run in any order as long as A and B agree: lane (r, q) takes k = 16 kt + 4 q + kk in MFMA
        // kk, so an operand that is contiguous in k is ONE 16-byte LDS read per K-tile
        for (int kt = k0; kt < k1; ++kt) {
            const int kb = 16 * kt + 4 * q;
            real av[4], bv[4];
            if (TA) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) av[kk] = A[(kb + kk) * LD + 16 * ti + r];
            } else {
                const real4 t4 = *reinterpret_cast<const real4*>(A + (16 * ti + r) * LD + kb);
                av[0] = t4[0]; av[1] = t4[1]; av[2] = t4[2]; av[3] = t4[3];
            }
            if (TB) {
                const real4 t4 = *reinterpret_cast<const real4*>(B + (16 * tj + r) * LD + kb);
                bv[0] = t4[0]; bv[1] = t4[1]; bv[2] = t4[2]; bv[3] = t4[3];
            } else {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) bv[kk] = B[(kb + kk) * LD + 16 * tj + r];
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = mfma(av[kk], bv[kk], acc);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            real* p = C + (16 * ti + acc_row(q, e)) * LD + 16 * tj + r;
            if (ACC) *p += alpha * acc[e]; else *p = alpha * acc[e];
        }
    }
}

// One 16 x 16 diagonal tile, handled by ONE wavefront (all 64 lanes run it, lanes 16..63 shadow lanes 0..15).
// CHOL: S holds a symmetric positive definite tile (lower triangle read): on exit S = L (lower, zeros above) and
//       Inv = L^-1.   !CHOL: S holds a lower-triangular tile, Inv = S^-1.
// Returns sum_i log(diag_i) of the triangular factor (uniform across lanes); `bad` on a non-positive pivot.
template <int LD, bool CHOL>
__device__ __forceinline__ real diag_tile(real* __restrict__ S, real* __restrict__ Inv, bool& bad) {
#ifdef MF_BIG_SKIP_DIAG
    return real(0);
#endif
    const int lane = threadIdx.x & 63, i = lane & 15;
    real a[16], rd[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = S[i * LD + k];
    real logsum = 0.f;
    // column `i` of the inverse by forward substitution; L[row][k] is lane `row`'s a[k]
    // (column-oriented: once x[k] is final every later row takes its contribution - independent updates instead of one
    // dependent accumulation chain per row)
    real x[16];
#pragma unroll
    for (int row = 0; row < 16; ++row) x[row] = (row == i) ? 1.f : 0.f;
    if constexpr (sizeof(real) == 4) {
        // fp32: every broadcast is the DPP operand (row_newbcast) of the instruction that consumes it (mf_big.hpp) - with
        // v_readlane the broadcast values of a tile lived in > 100 SGPRs, spilt to VGPR lanes and back
        if (CHOL) {
            static_for<0, 16>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const real p = mov_bcast16<j>(a[j]);
                bad |= !(p > 0.f);
                const real ri = t_rsqrt<real>(p);
                logsum += 0.5f * mf_log(p);
                const real lij = a[j] * ri;
                x[j] *= ri;
                if constexpr (j + 1 < 16) {
                    fnma_bcast16<j + 1>(a[j + 1], lij, lij);             // the next pivot first: it is the dependent chain
                    if constexpr (j + 2 < 16) fnma_bcast16_from<j + 2>(a, lij, lij);
                    fnma_bcast16_from<j + 1>(x, lij, x[j]);
                }
                a[j] = lij;
            });
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (lane < 16) S[i * LD + k] = (k <= i) ? a[k] : 0.f;
        } else {
            static_for<0, 16>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const real p = mov_bcast16<j>(a[j]);
                bad |= !(p != 0.f);
                rd[j] = t_rcp<real>(p);
                logsum += mf_log((p < 0 ? -p : p));
            });
            static_for<0, 16>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                x[k] *= rd[k];
                if constexpr (k + 1 < 16) fnma_bcast16_from<k + 1>(x, a[k], x[k]);
            });
        }
    } else
    if (CHOL) {
        // The factorisation and the substitution are ONE loop: step j of the inverse needs column j of L only, which is final
        // as soon as step j of the Cholesky has scaled it.  The two recurrences are independent chains of ~190 cycles per step
        // each; fused, the second one fills the issue slots the first leaves empty (r02: [5] of profiles/r02_big_v3_phases.txt).
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const real p = bcast(a[j], j);
            bad |= !(p > 0.f);
            const real ri = t_rsqrt<real>(p);
            rd[j] = ri;
            logsum += 0.5f * mf_log(p);
            const real lij = a[j] * ri;
            a[j] = lij;
            x[j] *= ri;
#pragma unroll
            for (int k = j + 1; k < 16; ++k) {
                const real lkj = bcast(lij, k);          // L[k][j]: one broadcast feeds the trailing update AND the substitution
                a[k] -= lij * lkj;
                x[k] -= lkj * x[j];
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (lane < 16) S[i * LD + k] = (k <= i) ? a[k] : 0.f;
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const real p = bcast(a[j], j);
            bad |= !(p != 0.f);
            rd[j] = t_rcp<real>(p);
            logsum += mf_log((p < 0 ? -p : p));
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            x[k] *= rd[k];
#pragma unroll
            for (int row = k + 1; row < 16; ++row) x[row] -= bcast(a[k], row) * x[k];
        }
    }
    if (lane < 16) {
#pragma unroll
        for (int row = 0; row < 16; ++row) Inv[row * LD + i] = x[row];
    }
    return logsum;
}

// ---- 16 x 16 tile tasks of the blocked factorisation / inversion: each is worked by ONE wavefront ------------------------------
// S_ti,jb <- S_ti,jb Inv_jb,jb^T  (panel: L_ij, one 16-deep product, in place)
template <int LD> __device__ __forceinline__ void fi_panel_tile(real* __restrict__ S, const real* __restrict__ Inv, int ti, int jb) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    real4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int k = 16 * jb + 4 * kk + q;
        acc = mfma(S[(16 * ti + r) * LD + k], Inv[(16 * jb + r) * LD + k], acc);          // Inv_jj[r][k] = (Inv_jj^T)[k][r]
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) S[(16 * ti + acc_row(q, e)) * LD + 16 * jb + r] = acc[e];
}
// S_ti,tk -= L_ti,jb L_tk,jb^T  (trailing update)
template <int LD> __device__ __forceinline__ void fi_trailing_tile(real* __restrict__ S, int ti, int tk, int jb) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    real4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int k = 16 * jb + 4 * kk + q;
        acc = mfma(S[(16 * ti + r) * LD + k], S[(16 * tk + r) * LD + k], acc);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) S[(16 * ti + acc_row(q, e)) * LD + 16 * tk + r] -= acc[e];
}
// Inv_ti,tj <- T_ti,tj = sum_{k=tj}^{ti-1} L_ti,k Inv_k,tj  (first half of an off-diagonal block of the inverse)
// (kend: the sum stops before block kend - the recursive order of the triangular inverse)
template <int LD> __device__ __forceinline__ void fi_inv_t_tile(const real* __restrict__ S, real* __restrict__ Inv, int ti, int tj,
                                                                int kend = 1 << 30) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    real4 acc = {0.f, 0.f, 0.f, 0.f};
    if (kend > ti) kend = ti;
    for (int kt = tj; kt < kend; ++kt) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = 16 * kt + 4 * kk + q;
            acc = mfma(S[(16 * ti + r) * LD + k], Inv[k * LD + 16 * tj + r], acc);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) Inv[(16 * ti + acc_row(q, e)) * LD + 16 * tj + r] = acc[e];
}
// Inv_ti,tj <- -Inv_ti,ti T_ti,tj  (second half, in place)
template <int LD> __device__ __forceinline__ void fi_inv_finish_tile(real* __restrict__ Inv, int ti, int tj) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    real4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int k = 4 * kk + q;
        acc = mfma(Inv[(16 * ti + r) * LD + 16 * ti + k], Inv[(16 * ti + k) * LD + 16 * tj + r], acc);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) Inv[(16 * ti + acc_row(q, e)) * LD + 16 * tj + r] = -acc[e];
}

// Column tj of the lower-left 2 x 2 block of the inverse from its first halves T_2j, T_3j (in Inv_2j, Inv_3j):
// Inv_3j = -(Inv_32 T_2j + Inv_33 T_3j), Inv_2j = -Inv_22 T_2j; both products are formed before either tile is overwritten.
template <int LD, int NT> __device__ __forceinline__ void fi_inv_finish_rows23(real* __restrict__ Inv, int tj) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    real4 a2 = {0.f, 0.f, 0.f, 0.f}, a3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int k = 4 * kk + q;
        const real t2 = Inv[(32 + k) * LD + 16 * tj + r];
        a2 = mfma(Inv[(32 + r) * LD + 32 + k], t2, a2);
        if (NT >= 4) {
            a3 = mfma(Inv[(48 + r) * LD + 32 + k], t2, a3);
            a3 = mfma(Inv[(48 + r) * LD + 48 + k], Inv[(48 + k) * LD + 16 * tj + r], a3);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        Inv[(32 + acc_row(q, e)) * LD + 16 * tj + r] = -a2[e];
        if (NT >= 4) Inv[(48 + acc_row(q, e)) * LD + 16 * tj + r] = -a3[e];
    }
}

// (CHOL) S <- chol(S) lower with zeros above inside the diagonal tiles, Inv <- L^-1 (full lower, upper tiles zeroed);
// (!CHOL) Inv <- S^-1 for lower-triangular S.  Returns log|det L| in wave 0 (valid on every lane of wave 0).
// All threads must call; ends with a barrier.
//
// CHOL schedule (look-ahead): the four diagonal tiles are a dependent chain on wave 0 (~2 k cycles each); everything else
// hides beside it.  Phases, one barrier each:
//   P0      wave 0: diagonal tile 0                         | waves 1-3: zero the upper tiles of Inv
//   A(jb)   panel tiles (ti, jb), ti > jb  +  second halves of the inverse's block row jb (-Inv_jj T_j,tj)
//   B(jb)   wave 0: trailing update of tile (jb+1, jb+1), then diagonal tile jb+1
//           waves 1-3: the other trailing tiles  +  first halves T_(jb+1),tj of the inverse's next block row
// = 2 NT phases instead of the 3 NT + 2 (NT - 1) of a phase per kind of work (r02: [5] of profiles/r02_big_v5_phases.txt).
template <int DP, bool CHOL>
__device__ __forceinline__ real factor_invert(real* __restrict__ S, real* __restrict__ Inv, bool& bad,
                                               real* __restrict__ red) {
    constexpr int LD = Geo<DP>::LD, NT = Geo<DP>::NT, NW = NTHR / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    real logdet = 0.f;
    if (CHOL) {
        if (wave == 0) {
            logdet += diag_tile<LD, true>(S, Inv, bad);
        } else {
            for (int e = threadIdx.x - 64; e < DP * DP; e += NTHR - 64) {
                const int row = e / DP, col = e % DP;
                if ((col >> 4) > (row >> 4)) Inv[row * LD + col] = 0.f;
            }
        }
        __syncthreads();
        for (int jb = 0; jb < NT; ++jb) {
            // A(jb): at most NT - 1 <= 3 tasks; wave 0 takes the panel tile it needs next
            {
                const int panels = NT - 1 - jb;
                for (int t = wave; t < NT - 1; t += NW) {
                    if (t < panels) fi_panel_tile<LD>(S, Inv, jb + 1 + t, jb);
                    else fi_inv_finish_tile<LD>(Inv, jb, t - panels);
                }
            }
            __syncthreads();
            if (jb + 1 == NT) break;
            // B(jb)
            if (wave == 0) {
                fi_trailing_tile<LD>(S, jb + 1, jb + 1, jb);
                logdet += diag_tile<LD, true>(S + 16 * (jb + 1) * LD + 16 * (jb + 1), Inv + 16 * (jb + 1) * LD + 16 * (jb + 1), bad);
            } else {
                int cnt = 0;
                for (int ti = jb + 1; ti < NT; ++ti)
                    for (int tk = jb + 1; tk <= ti; ++tk) {
                        if (ti == jb + 1) continue;                      // (jb+1, jb+1) is wave 0's
                        if ((cnt++ % (NW - 1)) + 1 == wave) fi_trailing_tile<LD>(S, ti, tk, jb);
                    }
                for (int tj = 0; tj <= jb; ++tj)
                    if ((cnt++ % (NW - 1)) + 1 == wave) fi_inv_t_tile<LD>(S, Inv, jb + 1, tj);
            }
            __syncthreads();
        }
        return logdet;
    }
    // zero the strictly-upper tiles of Inv
    for (int e = threadIdx.x; e < DP * DP; e += NTHR) {
        const int row = e / DP, col = e % DP;
        if ((col >> 4) > (row >> 4)) Inv[row * LD + col] = 0.f;
    }
    {
        // the diagonal tiles of a triangular matrix invert independently: one per wave; the per-wave
        // log-determinants and pivot flags meet through eight words of `red`
        real mine = 0.f;
        bool mybad = false;
        for (int jb = wave; jb < NT; jb += NW)
            mine += diag_tile<LD, false>(S + 16 * jb * LD + 16 * jb, Inv + 16 * jb * LD + 16 * jb, mybad);
        if (lane == 0) { red[wave] = mine; red[4 + wave] = mybad ? 1.f : 0.f; }
        __syncthreads();
        logdet = red[0] + red[1] + red[2] + red[3];
        bad |= (red[4] + red[5] + red[6] + red[7]) != 0.f;
    }
    // off-diagonal blocks of the inverse, Inv_ij = -Inv_ii sum_{k=j}^{i-1} L_ik Inv_kj, in the recursive 2 x 2 order: first the
    // blocks (1,0) and (3,2) inside the two halves, then the block rows 2, 3 against the finished upper half - four phases for
    // NT = 4 instead of the six of a row-by-row sweep, and all four waves busy in the last two
    if (NT >= 2) {
        if (wave == 0) fi_inv_t_tile<LD>(S, Inv, 1, 0);
        if (wave == 1 && NT >= 4) fi_inv_t_tile<LD>(S, Inv, 3, 2);
        __syncthreads();
        if (wave == 0) fi_inv_finish_tile<LD>(Inv, 1, 0);
        if (wave == 1 && NT >= 4) fi_inv_finish_tile<LD>(Inv, 3, 2);
        __syncthreads();
    }
    if (NT >= 3) {
        // T_ij = sum_{k=j}^{1} L_ik Inv_kj for i in {2, 3}, j in {0, 1}: the sum stops at the upper half (k <= 1)
        for (int t = wave; t < 2 * (NT - 2); t += NW) fi_inv_t_tile<LD>(S, Inv, 2 + (t >> 1), t & 1, 2);
        __syncthreads();
        // [Inv_2j; Inv_3j] = -[Inv_22 0; Inv_32 Inv_33] [T_2j; T_3j]: one wave per column j, row 3 first (it reads T_2j)
        if (wave < 2) fi_inv_finish_rows23<LD, NT>(Inv, wave);
        __syncthreads();
    }
    return logdet;
}


// ---- vectors ---------------------------------------------------------------------------------------------------
// out[r] = beta out[r] + alpha sum_k op(M)[r][k] v[k];  out must not alias v.  `scratch`: 4*64 floats (TRANS only).
// Ends with a barrier.
template <int DP, int TRANS>
__device__ __forceinline__ void matvec(const real* __restrict__ M, const real* __restrict__ v, real* __restrict__ out,
                                       real alpha, real beta, real* __restrict__ scratch) {
#ifdef MF_BIG_SKIP_MATVEC
    return;
#endif
    constexpr int LD = Geo<DP>::LD;
    if (!TRANS) {
        const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
        real p = 0.f;
        if (r < DP)
            for (int k = q; k < DP; k += 4) p += M[r * LD + k] * v[k];
        p += __shfl_xor(p, 1);
        p += __shfl_xor(p, 2);
        if (q == 0 && r < DP) out[r] = (beta == 0.f ? 0.f : beta * out[r]) + alpha * p;
        __syncthreads();
    } else {
        const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
        real p = 0.f;
        if (c < DP)
            for (int k = q; k < DP; k += 4) p += M[k * LD + c] * v[k];
        scratch[q * 64 + c] = p;
        __syncthreads();
        if (threadIdx.x < DP) {
            const int t = threadIdx.x;
            const real sum = scratch[t] + scratch[64 + t] + scratch[128 + t] + scratch[192 + t];
            out[t] = (beta == 0.f ? 0.f : beta * out[t]) + alpha * sum;
        }
        __syncthreads();
    }
}

// sum_i v[i]^2 (i < DP), the same value on every thread
template <int DP> __device__ __forceinline__ real sumsq(const real* __restrict__ v) {
    const int lane = threadIdx.x & 63;
    real x = lane < DP ? v[lane] : 0.f;
    x *= x;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// ---- global <-> LDS ----------------------------------------------------------------------------------------------
// tile <- g [d x d] (+ g2), zero padded; lower: the upper triangle is zeroed; idpad: identity on the padded diagonal
template <int DP>
__device__ __forceinline__ void load_tile(real* __restrict__ tile, const real* __restrict__ g, const real* __restrict__ g2,
                                          int d, bool lower, bool idpad) {
    constexpr int LD = Geo<DP>::LD;
    for (int e = threadIdx.x; e < DP * DP; e += NTHR) {
        const int row = e / DP, col = e % DP;
        real v = 0.f;
        if (row < d && col < d && (!lower || col <= row)) {
            v = g[row * d + col];
            if (g2) v += g2[row * d + col];
        } else if (idpad && row == col && row >= d) {
            v = 1.f;
        }
        tile[row * LD + col] = v;
    }
}
// The next transition's d x d block held in registers (NE floats per thread) between its global load, issued one
// step ahead, and its landing in an LDS tile: HBM latency is covered by a whole step of arithmetic.
template <int DP> struct TilePrefetch {
    static constexpr int NE = DP * DP / NTHR;
    real v[NE];
    __device__ __forceinline__ void load(const real* __restrict__ g, int d, bool lower, bool idpad) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + i * NTHR, row = e / DP, col = e % DP;
            real x = 0.f;
            if (row < d && col < d && (!lower || col <= row)) x = g[row * d + col];
            else if (idpad && row == col && row >= d) x = 1.f;
            v[i] = x;
        }
    }
    __device__ __forceinline__ void store(real* __restrict__ tile) const {
        constexpr int LD = Geo<DP>::LD;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + i * NTHR;
            tile[(e / DP) * LD + (e % DP)] = v[i];
        }
    }
};

template <int DP> __device__ __forceinline__ void zero_tile(real* __restrict__ tile) {
    constexpr int LD = Geo<DP>::LD;
    for (int e = threadIdx.x; e < DP * LD; e += NTHR) tile[e] = 0.f;
}
template <int DP> __device__ __forceinline__ void store_tile(real* __restrict__ g, const real* __restrict__ tile, int d) {
    constexpr int LD = Geo<DP>::LD;
    for (int e = threadIdx.x; e < d * d; e += NTHR) g[e] = tile[(e / d) * LD + (e % d)];
}
template <int DP>
__device__ __forceinline__ void load_vec_lds(real* __restrict__ v, const real* __restrict__ g, const real* __restrict__ g2, int d) {
    if (threadIdx.x < DP) {
        real x = 0.f;
        if (threadIdx.x < d) { x = g[threadIdx.x]; if (g2) x += g2[threadIdx.x]; }
        v[threadIdx.x] = x;
    }
}

// LDS carve of one workgroup
template <int DP> struct Smem {
    static constexpr int TILE = Geo<DP>::TILE, LD = Geo<DP>::LD;
    static constexpr int OBS_ROWS = MAXM_BIG;
    static constexpr int N_TILES = 7;
    static constexpr int FLOATS = N_TILES * TILE + 2 * OBS_ROWS * LD + MAXM_BIG * MAXM_BIG + 10 * 64 + 256 + 2 * MAXM_BIG;
    static constexpr int BYTES = FLOATS * (int)sizeof(real);
    real* base;
    __device__ real* tile(int i) const { return base + i * TILE; }
    __device__ real* Hs() const { return base + N_TILES * TILE; }
    __device__ real* Gs() const { return Hs() + OBS_ROWS * LD; }
    __device__ real* Rs() const { return Gs() + OBS_ROWS * LD; }
    __device__ real* vec(int i) const { return Rs() + MAXM_BIG * MAXM_BIG + i * 64; }
    __device__ real* scratch() const { return vec(10); }
    __device__ real* ys() const { return scratch() + 256; }
    __device__ real* rys() const { return ys() + MAXM_BIG; }
};
enum { T_PHI = 0, T_X = 1, T_GU = 2, T_U1 = 3, T_U2 = 4, T_U3 = 5, T_U4 = 6 };
enum { V_M = 0, V_W = 1, V_BTW = 2, V_T = 3, V_Z = 4, V_GU = 5, V_RN = 6, V_TMP = 7 };

// Observation terms of one block: Phi += H^T R^-1 H, tvec += H^T R^-1 y, returns y^T R^-1 y (uniform).
// Rs already holds R^-1 when it is shared; per-step precisions are loaded here.  Ends with a barrier.
// Hk == NULL: H_k and y_k were already put into Hs / ys by the caller (the level-0 kernel prefetches them one step ahead).
template <int DP>
__device__ __forceinline__ real obs_terms(const Smem<DP>& sm, real* __restrict__ Phi, real* __restrict__ tvec,
                                           const real* __restrict__ Hk, const real* __restrict__ yk,
                                           const real* __restrict__ Rk, int d, int m) {
    constexpr int LD = Geo<DP>::LD;
    const int mp = (m + 15) & ~15;
    real *Hs = sm.Hs(), *Gs = sm.Gs(), *Rs = sm.Rs(), *ys = sm.ys(), *rys = sm.rys();
    if (Hk) {
        for (int e = threadIdx.x; e < mp * DP; e += NTHR) {
            const int o = e / DP, i = e % DP;
            Hs[o * LD + i] = (o < m && i < d) ? Hk[o * d + i] : 0.f;
        }
        if (threadIdx.x < m) ys[threadIdx.x] = yk ? yk[threadIdx.x] : real(0);      // yk == NULL: precision only
    }
    if (Rk) for (int e = threadIdx.x; e < m * m; e += NTHR) Rs[e] = Rk[e];
    if (Hk || Rk) __syncthreads();
    {   // G = R^-1 H on the matrix cores: (mp/16) x NT output tiles of 16 x 16, K = mp
        constexpr int NT = Geo<DP>::NT;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
        for (int idx = wave; idx < (mp / 16) * NT; idx += NTHR / 64) {
            const int ti = idx / NT, tj = idx % NT, o = 16 * ti + r;
            real4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int kt = 0; kt < mp / 16; ++kt) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int p = 16 * kt + 4 * q + kk;
                    // R^-1 is symmetric: read it down a column (consecutive lanes -> consecutive words) instead of along a row
                    // (stride m = 32 words: every lane of the wave on the same LDS bank)
                    const real av = (o < m && p < m) ? Rs[p * m + o] : real(0);
                    acc = mfma(av, Hs[p * LD + 16 * tj + r], acc);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) Gs[(16 * ti + acc_row(q, e)) * LD + 16 * tj + r] = acc[e];
        }
    }
    {   // R^-1 y: wave w takes outputs 8 w ... 8 w + 7 with eight lanes per output (m <= 32; R^-1 symmetric: read down a column)
        const int lane = threadIdx.x & 63, o = 8 * (threadIdx.x >> 6) + (lane & 7), part = lane >> 3;
        real a = 0.f;
        if (o < m)
            for (int p = part; p < m; p += 8) a += Rs[p * m + o] * ys[p];
        a += __shfl_xor(a, 8);
        a += __shfl_xor(a, 16);
        a += __shfl_xor(a, 32);
        if (part == 0 && o < m) rys[o] = a;
    }
    __syncthreads();
    gemm<DP, 1, 0, 1, K_FULL, O_FULL>(Hs, Gs, Phi, 1.f, mp / 16);
    {   // tvec += H^T (R^-1 y): wave w takes outputs 16 w ... 16 w + 15 with four lanes per output
        const int lane = threadIdx.x & 63, i = 16 * (threadIdx.x >> 6) + (lane & 15), part = lane >> 4;
        real a = 0.f;
        if (i < DP)
            for (int o = part; o < m; o += 4) a += Hs[o * LD + i] * rys[o];
        a += __shfl_xor(a, 16);
        a += __shfl_xor(a, 32);
        if (part == 0 && i < DP) tvec[i] += a;
    }
    // y^T R^-1 y: one product per lane and a wave reduction (every wave holds the same value; m <= 32 < 64 lanes)
    const int lane = threadIdx.x & 63;
    real yry = lane < m ? ys[lane] * rys[lane] : real(0);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) yry += __shfl_xor(yry, off);
    __syncthreads();
    return yry;
}

// Eliminate the block whose complete pivot is in Phi: Linv -> U2, z -> V_Z, spike V -> U4 folded into GU / gU.
template <int DP>
__device__ __forceinline__ void eliminate(const Smem<DP>& sm, bool spike, double& logL, double& quad, bool& bad) {
    logL += (double)factor_invert<DP, true>(sm.tile(T_PHI), sm.tile(T_U2), bad, sm.scratch());
    matvec<DP, 0>(sm.tile(T_U2), sm.vec(V_T), sm.vec(V_Z), 1.f, 0.f, sm.scratch());
    quad += (double)sumsq<DP>(sm.vec(V_Z));
    if (spike) {
        gemm<DP, 0, 0, 0, K_A_LOWER, O_FULL>(sm.tile(T_U2), sm.tile(T_X), sm.tile(T_U4), 1.f);      // V = Linv X
        __syncthreads();
        gemm<DP, 1, 0, 1, K_FULL, O_FULL>(sm.tile(T_U4), sm.tile(T_U4), sm.tile(T_GU), -1.f);       // GU -= V^T V
        matvec<DP, 1>(sm.tile(T_U4), sm.vec(V_Z), sm.vec(V_GU), -1.f, 1.f, sm.scratch());           // gU -= V^T z
    }
}
// After eliminate(): W is in U1, the next block's own pivot part in Phi and rhs part in V_RN.
template <int DP> __device__ __forceinline__ void advance(const Smem<DP>& sm, bool spike) {
    gemm<DP, 0, 1, 1, K_FULL, O_FULL>(sm.tile(T_U1), sm.tile(T_U1), sm.tile(T_PHI), -1.f);          // Phi -= W W^T
    if (threadIdx.x < DP) sm.vec(V_T)[threadIdx.x] = sm.vec(V_RN)[threadIdx.x];
    __syncthreads();
    matvec<DP, 0>(sm.tile(T_U1), sm.vec(V_Z), sm.vec(V_T), -1.f, 1.f, sm.scratch());                // t = rn - W z
    if (spike) gemm<DP, 0, 0, 0, K_FULL, O_FULL>(sm.tile(T_U1), sm.tile(T_U4), sm.tile(T_X), -1.f); // X = -W V
    __syncthreads();
}

template <int DP>
__device__ __forceinline__ void store_chunk_big(const Smem<DP>& sm, const RedSys<real>& out, long idx, int d, real scalar) {
    store_tile<DP>(out.Dv + idx * d * d, sm.tile(T_PHI), d);
    store_tile<DP>(out.GU + idx * d * d, sm.tile(T_GU), d);
    store_tile<DP>(out.F + idx * d * d, sm.tile(T_X), d);
    if (threadIdx.x < d) {
        out.tv[idx * d + threadIdx.x] = sm.vec(V_T)[threadIdx.x];
        out.gU[idx * d + threadIdx.x] = sm.vec(V_GU)[threadIdx.x];
    }
    if (threadIdx.x == 0) out.sc[idx] = scalar;
}

struct BigArgs {
    long B, Tn;
    int d, m;
    const real *mu0, *cholP0, *A, *b, *cholQ, *H, *y, *Rinv;
    int rinv_per_step;
    long P, L;          // chunks per series, transitions per chunk
    int* info;
};

// Phase timing of the level-0 step (diagnostic builds only: -DMF_BIG_STAMP; scripts/bench_big.py prints the table that
// block 0 emits).  s_memtime on thread 0 between the phases of a step, accumulated over the chunk.
#ifdef MF_BIG_STAMP
#define MF_STAMP_DECL unsigned long long st_acc[12] = {0}, st_prev = __builtin_readcyclecounter();
#define MF_STAMP(i) { const unsigned long long now = __builtin_readcyclecounter(); st_acc[i] += now - st_prev; st_prev = now; }
#define MF_STAMP_PRINT(steps)                                                                                          \
    if (blockIdx.x == 1 && threadIdx.x == 0) {                                                                         \
        printf("big step phases (memtime ticks per step, %ld steps):", (long)(steps));                                 \
        for (int i = 0; i < 12; ++i) printf(" [%d] %.0f", i, (double)st_acc[i] / (double)((steps) > 0 ? (steps) : 1)); \
        printf("\n");                                                                                                  \
    }
#else
#define MF_STAMP_DECL
#define MF_STAMP(i)
#define MF_STAMP_PRINT(steps)
#endif

// Level 0: workgroup (s, c) eliminates the transitions [c L, min((c+1) L, T-1)) of series s.
template <int DP> __global__ void __launch_bounds__(NTHR) big_kf_chunk_kernel(BigArgs a, RedSys<real> out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long id = blockIdx.x, s = id / a.P, c = id % a.P;
    const int d = a.d, m = a.m;
    const long nt = a.Tn - 1, tau0 = c * a.L;
    long len = nt - tau0;
    if (len > a.L) len = a.L;
    if (len < 0) len = 0;
    const bool spike = c > 0;
    double logC = 0.0, logL = 0.0, quad = 0.0, acc_ww = 0.0, acc_yry = 0.0;
    bool bad = false;
    for (int i = 0; i < 3; ++i) zero_tile<DP>(sm.tile(i));
    if (threadIdx.x < 64) { sm.vec(V_T)[threadIdx.x] = 0.f; sm.vec(V_GU)[threadIdx.x] = 0.f; }
    if (!a.rinv_per_step) for (int e = threadIdx.x; e < m * m; e += NTHR) sm.Rs()[e] = a.Rinv[e];
    __syncthreads();
    real *Phi = sm.tile(T_PHI), *U1 = sm.tile(T_U1), *U2 = sm.tile(T_U2), *Ci = sm.tile(T_U3);

    // pivot part Q^-1 (+ observation) and rhs part of the block a Cholesky factor C (already inverted into Ci) leads to
    // `staged`: H and y of the block are already in LDS (prefetched one step ahead with the transition)
    auto own_terms = [&](long blk, real* tvec, bool staged) {
        gemm<DP, 1, 0, 0, K_A_UPPER, O_FULL>(Ci, Ci, Phi, 1.f);                                   // Phi = Ci^T Ci
        matvec<DP, 1>(Ci, sm.vec(V_W), tvec, 1.f, 0.f, sm.scratch());                             // Ci^T w
        const real* Rk = a.rinv_per_step ? a.Rinv + (s * a.Tn + blk) * m * m : nullptr;
        acc_yry += (double)obs_terms<DP>(sm, Phi, tvec, staged ? nullptr : a.H + (s * a.Tn + blk) * m * d,
                                         a.y + (s * a.Tn + blk) * m, Rk, d, m);
    };

    if (c == 0) {   // block 0: the prior
        load_tile<DP>(U1, a.cholP0 + s * d * d, nullptr, d, true, true);
        load_vec_lds<DP>(sm.vec(V_M), a.mu0 + s * d, nullptr, d);
        __syncthreads();
        logC += (double)factor_invert<DP, false>(U1, Ci, bad, sm.scratch());
        matvec<DP, 0>(Ci, sm.vec(V_M), sm.vec(V_W), 1.f, 0.f, sm.scratch());
        acc_ww += (double)sumsq<DP>(sm.vec(V_W));
        own_terms(0, sm.vec(V_T), false);
    }
    TilePrefetch<DP> pfC, pfA;
    real pfb = 0.f;
    // H (m x d, zero-padded to mp x DP in LDS) and y of the block the transition leads to: NH values per thread
    constexpr int NH = MAXM_BIG * DP / NTHR;
    real pfH[NH], pfy = 0.f;
    const int mp_all = (m + 15) & ~15;
    auto prefetch = [&](long tau) {
        pfC.load(a.cholQ + (s * nt + tau) * d * d, d, true, true);
        pfA.load(a.A + (s * nt + tau) * d * d, d, false, false);
        pfb = (threadIdx.x < d) ? a.b[(s * nt + tau) * d + threadIdx.x] : 0.f;
    };
    // The observation rows of the block a transition leads to are needed at the END of its step (own_terms).  Their loads are
    // issued in the middle of the step - after the factorisation, whose unrolled diagonal-tile code has no registers to
    // spare - and land in LDS just before own_terms: the latency hides behind the two GEMMs in between.
    auto load_obs = [&](long blk) {
        const real* Hn = a.H + (s * a.Tn + blk) * m * d;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int e = threadIdx.x + i * NTHR, o = e / DP, c2 = e % DP;
            pfH[i] = (o < m && c2 < d) ? Hn[o * d + c2] : 0.f;
        }
        pfy = (threadIdx.x < m) ? a.y[(s * a.Tn + blk) * m + threadIdx.x] : 0.f;
    };
    auto stage_obs = [&]() {
        constexpr int LDh = Geo<DP>::LD;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int e = threadIdx.x + i * NTHR;
            if (e < mp_all * DP) sm.Hs()[(e / DP) * LDh + (e % DP)] = pfH[i];
        }
        if (threadIdx.x < m) sm.ys()[threadIdx.x] = pfy;
        __syncthreads();
    };
    if (len > 0) prefetch(tau0);
    MF_STAMP_DECL
    for (long j = 0; j < len; ++j) {
        const long tau = tau0 + j;
        MF_STAMP(11)
        pfC.store(U1);
        pfA.store(U2);
        if (threadIdx.x < DP) sm.vec(V_M)[threadIdx.x] = pfb;
        __syncthreads();
        if (j + 1 < len) prefetch(tau + 1);      // in flight during the whole step
        MF_STAMP(0)
        logC += (double)factor_invert<DP, false>(U1, Ci, bad, sm.scratch());
        MF_STAMP(1)
        matvec<DP, 0>(Ci, sm.vec(V_M), sm.vec(V_W), 1.f, 0.f, sm.scratch());
        acc_ww += (double)sumsq<DP>(sm.vec(V_W));
        MF_STAMP(2)
        gemm<DP, 0, 0, 0, K_A_LOWER, O_FULL>(Ci, U2, U1, 1.f);                                    // Bm = Ci A
        __syncthreads();
        MF_STAMP(3)
        matvec<DP, 1>(U1, sm.vec(V_W), sm.vec(V_BTW), 1.f, 0.f, sm.scratch());                    // Bm^T w
        MF_STAMP(2)
        if (j == 0 && spike) {
            // the block on the left is the chunk's separator: its coupling seeds the spike
            load_obs(tau + 1);
            gemm<DP, 1, 0, 0, K_FULL, O_FULL>(U1, U1, sm.tile(T_GU), 1.f);                         // GU = Bm^T Bm
            gemm<DP, 1, 0, 0, K_A_UPPER, O_FULL>(Ci, U1, sm.tile(T_X), -1.f);                      // X = -Ci^T Bm
            if (threadIdx.x < DP) sm.vec(V_GU)[threadIdx.x] = -sm.vec(V_BTW)[threadIdx.x];
            stage_obs();
            own_terms(tau + 1, sm.vec(V_T), true);
        } else {
            gemm<DP, 1, 0, 1, K_FULL, O_FULL>(U1, U1, Phi, 1.f);                                  // Phi += Bm^T Bm
            if (threadIdx.x < DP) sm.vec(V_T)[threadIdx.x] -= sm.vec(V_BTW)[threadIdx.x];
            __syncthreads();
            MF_STAMP(4)
#ifdef MF_BIG_STAMP
            logL += (double)factor_invert<DP, true>(sm.tile(T_PHI), sm.tile(T_U2), bad, sm.scratch());
            MF_STAMP(5)
            matvec<DP, 0>(sm.tile(T_U2), sm.vec(V_T), sm.vec(V_Z), 1.f, 0.f, sm.scratch());
            quad += (double)sumsq<DP>(sm.vec(V_Z));
            MF_STAMP(2)
            if (spike) {
                gemm<DP, 0, 0, 0, K_A_LOWER, O_FULL>(sm.tile(T_U2), sm.tile(T_X), sm.tile(T_U4), 1.f);
                __syncthreads();
                gemm<DP, 1, 0, 1, K_FULL, O_FULL>(sm.tile(T_U4), sm.tile(T_U4), sm.tile(T_GU), -1.f);
                MF_STAMP(6)
                matvec<DP, 1>(sm.tile(T_U4), sm.vec(V_Z), sm.vec(V_GU), -1.f, 1.f, sm.scratch());
                MF_STAMP(2)
            }
#else
            eliminate<DP>(sm, spike, logL, quad, bad);
#endif
            load_obs(tau + 1);
            gemm<DP, 0, 1, 0, K_B_UPPER, O_FULL>(U1, U2, sm.tile(T_X), 1.f);                       // Y = Bm Linv^T
            __syncthreads();
            gemm<DP, 1, 0, 0, K_A_UPPER, O_FULL>(Ci, sm.tile(T_X), U1, -1.f);                      // W = -Ci^T Y
            stage_obs();
            MF_STAMP(7)
            own_terms(tau + 1, sm.vec(V_RN), true);
            MF_STAMP(8)
            advance<DP>(sm, spike);
            MF_STAMP(9)
        }
    }
    MF_STAMP_PRINT(len)
    store_chunk_big<DP>(sm, out, id, d, (real)(-0.5 * (acc_yry + acc_ww) + 0.5 * quad - logC - logL));
    if (threadIdx.x == 0 && bad && a.info) raise_info(a.info);
}

// Reduction level: RedSys(n) -> RedSys(P) (FINAL: P = 1, the last block is eliminated too and out_scalar written).
template <int DP, bool FINAL>
__global__ void __launch_bounds__(NTHR) big_red_kernel(RedSys<real> in, RedSys<real> out, long B, long P, int d,
                                                      real add_const, real* __restrict__ out_scalar, int* info) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long id = blockIdx.x, s = id / P, c = id % P;
    const long k0 = (c * in.n) / P, k1 = ((c + 1) * in.n) / P;
    const bool spike = !FINAL && k0 > 0;
    double logL = 0.0, quad = 0.0, acc_sc = 0.0;
    bool bad = false;
    for (int i = 0; i < 3; ++i) zero_tile<DP>(sm.tile(i));
    if (threadIdx.x < 64) { sm.vec(V_T)[threadIdx.x] = 0.f; sm.vec(V_GU)[threadIdx.x] = 0.f; }
    __syncthreads();
    for (long k = k0; k < k1; ++k) {
        const long idx = s * in.n + k;
        const bool has_next = in.GU && (k + 1 < in.n);
        acc_sc += in.sc ? (double)in.sc[idx] : 0.0;
        if (k > k0) {
            eliminate<DP>(sm, spike, logL, quad, bad);
            load_tile<DP>(sm.tile(T_U3), in.F + (s * in.f_stride + k + in.f_off) * d * d, nullptr, d, false, false);
            __syncthreads();
            gemm<DP, 0, 1, 0, K_B_UPPER, O_FULL>(sm.tile(T_U3), sm.tile(T_U2), sm.tile(T_U1), 1.f);   // W = F Linv^T
        } else if (k > 0) {
            load_tile<DP>(sm.tile(T_X), in.F + (s * in.f_stride + k + in.f_off) * d * d, nullptr, d, false, false);
        }
        // the block's own pivot / rhs parts (padded diagonal = 1 keeps the padded states harmless)
        load_tile<DP>(sm.tile(T_PHI), in.Dv + idx * d * d, has_next ? in.GU + (idx + 1) * d * d : nullptr, d, false, true);
        load_vec_lds<DP>(sm.vec(V_RN), in.tv + idx * d, has_next ? in.gU + (idx + 1) * d : nullptr, d);
        __syncthreads();
        if (k > k0) {
            advance<DP>(sm, spike);
        } else {
            if (threadIdx.x < DP) sm.vec(V_T)[threadIdx.x] = sm.vec(V_RN)[threadIdx.x];
            __syncthreads();
        }
    }
    if (FINAL) {
        eliminate<DP>(sm, false, logL, quad, bad);
        if (threadIdx.x == 0) out_scalar[s] = (real)((double)add_const + acc_sc + 0.5 * quad - logL);
    } else {
        store_chunk_big<DP>(sm, out, id, d, (real)(acc_sc + 0.5 * quad - logL));
    }
    if (threadIdx.x == 0 && bad && info) raise_info(info);
}


// ---- host side: workspace carving, partition choice, launches ---------------------------------------------------------------


constexpr long BIG_RED_CHUNK = 8, BIG_RED_FINAL = 8;
// the panel kernels' reduction levels: a block step costs ~10 us whatever the number of workgroups, so the tree is made shallow in
// SERIAL steps - 64 chunk ends -> 16 -> 4 -> 1 is 3 + 3 + 4 block steps in three launches, radix 8 is 7 + 8 in two
constexpr long PANEL_RED_CHUNK = 4, PANEL_RED_FINAL = 4;
inline long cdivl(long a, long b) { return (a + b - 1) / b; }
inline size_t align_up_big(size_t x) { return (x + 255) & ~size_t(255); }
inline long red_elems(int d) { return 3L * d * d + 2L * d + 1; }
inline size_t red_bytes_big(long B, long n, int d) { return align_up_big(size_t(B) * n * red_elems(d) * sizeof(real)); }

inline RedSys<real> carve_big(char*& p, long B, long n, int d) {
    RedSys<real> r;
    real* base = reinterpret_cast<real*>(p);
    const long nb = B * n, dd = long(d) * d;
    r.Dv = base;
    r.GU = r.Dv + nb * dd;
    r.F = r.GU + nb * dd;
    r.tv = r.F + nb * dd;
    r.gU = r.tv + nb * d;
    r.sc = r.gU + nb * d;
    r.n = n;
    r.f_stride = n;
    r.f_off = 0;
    p += red_bytes_big(B, n, d);
    return r;
}

// chunks per series: enough workgroups for two rounds over the 256 CUs, chunks of at least 4 transitions.  wave_target > 0: the
// level-0 kernel is the wave kernel (mf_wave.hpp), one WAVEFRONT per chunk - one round of wavefronts over the SIMDs at its occupancy
inline void big_partition(long B, long Tn, long chunks, long& P, long& L, long wave_target = 0) {
    static const long target_wg = [] { const char* e = mf_knob("MF_BIG_TARGET_WGS"); return e ? std::atol(e) : 512L; }();
    static const long wave_force = [] { const char* e = mf_knob("MF_WAVE_TARGET"); return e ? std::atol(e) : 0L; }();
    const long target = wave_target > 0 ? (wave_force > 0 ? wave_force : wave_target) : target_wg;
    const long nt = Tn - 1;
    if (nt < 1) { P = 1; L = 1; return; }
    long want = chunks > 0 ? chunks : cdivl(target, B);
    const long maxP = nt / 4 > 0 ? nt / 4 : 1;
    if (want > maxP) want = maxP;
    if (want < 1) want = 1;
    L = cdivl(nt, want);
    P = cdivl(nt, L);
}

template <int DP> int launch_big(long B, long Tn, int d, int m, const real* mu0, const real* cholP0, const real* A,
                                 const real* b, const real* cholQ, const real* H, const real* y, const real* Rinv,
                                 int rinv_per_step, real add_const, real* out, void* ws, int* info, long P, long L,
                                 hipEvent_t ev0, hipEvent_t ev1, hipStream_t st, bool wave = false) {
    using SM = Smem<DP>;
    static const bool attr_ok = [] {
        bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&big_kf_chunk_kernel<DP>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SM::BYTES) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&big_red_kernel<DP, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, SM::BYTES) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&big_red_kernel<DP, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, SM::BYTES) == hipSuccess;
        return ok;
    }();
    if (!attr_ok) return -1000;
    char* p = static_cast<char*>(ws);
    RedSys<real> cur = carve_big(p, B, P, d);
    BigArgs a{B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, info};
    if (ev0) (void)hipEventRecord(ev0, st);
    if (wave) {
        const int rc = wave_kf_level0(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, cur, info, st);
        if (rc != 0) return rc;
    } else {
        hipLaunchKernelGGL((big_kf_chunk_kernel<DP>), dim3((unsigned)(B * P)), dim3(NTHR), SM::BYTES, st, a, cur);
    }
    if (ev1) (void)hipEventRecord(ev1, st);
    while (cur.n > BIG_RED_FINAL) {
        const long Pn = cdivl(cur.n, BIG_RED_CHUNK);
        RedSys<real> nxt = carve_big(p, B, Pn, d);
        hipLaunchKernelGGL((big_red_kernel<DP, false>), dim3((unsigned)(B * Pn)), dim3(NTHR), SM::BYTES, st, cur,
                           nxt, B, Pn, d, 0.f, static_cast<real*>(nullptr), info);
        cur = nxt;
    }
    hipLaunchKernelGGL((big_red_kernel<DP, true>), dim3((unsigned)B), dim3(NTHR), SM::BYTES, st, cur, cur, B, 1L,
                       d, add_const, out, info);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}


// 32 < d <= 64: level 0 and the reduction levels on the panel kernels (mf_panel.hpp); same partition, same workspace layout
inline bool panel_path(int d, int m) {
    static const bool off = [] { const char* e = mf_knob("MF_PANEL"); return e && e[0] == '0'; }();
    return !off && panel_covers(d, m);
}
inline int launch_panel(long B, long Tn, int d, int m, const real* mu0, const real* cholP0, const real* A, const real* b,
                        const real* cholQ, const real* H, const real* y, const real* Rinv, int rinv_per_step, real add_const,
                        real* out, void* ws, int* info, long P, long L, hipEvent_t ev0, hipEvent_t ev1, hipStream_t st) {
    char* p = static_cast<char*>(ws);
    RedSys<real> cur = carve_big(p, B, P, d);
    if (ev0) (void)hipEventRecord(ev0, st);
    int rc = panel_kf_level0(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, L, cur, info, st);
    if (rc != 0) return rc;
    if (ev1) (void)hipEventRecord(ev1, st);
    while (cur.n > PANEL_RED_FINAL) {
        const long Pn = cdivl(cur.n, PANEL_RED_CHUNK);
        RedSys<real> nxt = carve_big(p, B, Pn, d);
        rc = panel_red(cur, nxt, B, Pn, d, real(0), static_cast<real*>(nullptr), info, 0, st);
        if (rc != 0) return rc;
        cur = nxt;
    }
    return panel_red(cur, cur, B, 1L, d, add_const, out, info, 1, st);
}

// wavefronts of the wave kernel in one round over the chip (0: the state dimension is not the wave kernel's)
inline long wave_target(int d) {
    static const bool off = [] { const char* e = mf_knob("MF_WAVE"); return e && e[0] == '0'; }();
    if (off || !wave_covers(d, 1)) return 0;
    // (one tile per matrix in fp32: two rounds of wavefronts - the first chunk of a series carries no spike and finishes early, a
    // second round evens that out: 1.10 -> 1.05 ms at d = 16, B = 512, T = 1000; in fp64, since the paired pass, one round is as
    // good or better: 1.555 against 1.58 ms; 2 x 2 tiles: one round, more chunks only add spikes)
    return 256L * 4 * wave_waves_per_simd(d, (int)sizeof(real)) * (d <= 16 && sizeof(real) == 4 ? 2 : 1);
}
inline long panel_target_small() { return sizeof(real) == 4 ? 2048 : 1024; }
inline size_t kf_loglik_ws_for(long B, long Tn, int d, long chunks, long wtarget, bool panel = false) {
    long P, L;
    big_partition(B, Tn, chunks, P, L, wtarget);
    // (the panel kernels' levels shrink by PANEL_RED_CHUNK)
    const long rchunk = panel ? PANEL_RED_CHUNK : BIG_RED_CHUNK, rfinal = panel ? PANEL_RED_FINAL : BIG_RED_FINAL;
    size_t total = red_bytes_big(B, P, d);
    long n = P;
    while (n > rfinal) {
        n = cdivl(n, rchunk);
        total += red_bytes_big(B, n, d);
    }
    return total;
}
// (the query does not know the observation dimension, which decides between the wave kernel and the engine: the larger of the two)
inline size_t kf_loglik_ws(long B, long Tn, int d, long chunks) {
    const size_t w0 = kf_loglik_ws_for(B, Tn, d, chunks, 0);
    const long wt = wave_target(d);
    const size_t w1 = wt > 0 ? kf_loglik_ws_for(B, Tn, d, chunks, wt) : 0;
    size_t w2 = 0;
    if (d > 32) w2 = kf_loglik_ws_for(B, Tn, d, chunks, 0, true);
    else if (d > 16) w2 = kf_loglik_ws_for(B, Tn, d, chunks, panel_target_small(), true);
    return std::max(std::max(w0, w1), w2);
}

inline int kf_loglik(long B, long Tn, int d, int m, const real* mu0, const real* cholP0, const real* A, const real* b,
                     const real* cholQ, const real* H, const real* y, const real* Rinv, int rinv_per_step, real add_const,
                     real* out, void* ws, size_t ws_bytes, int* info, long chunks, hipEvent_t ev0, hipEvent_t ev1,
                     hipStream_t st) {
    if (m < 1 || m > MAXM_BIG) return -4;
    if (ws == nullptr || ws_bytes < kf_loglik_ws(B, Tn, d, chunks)) return -15;
    long P, L;
    const long wt = (m <= 4 && Tn >= 2) ? wave_target(d) : 0;
    const bool wave = wt > 0 && wave_covers(d, m);
    big_partition(B, Tn, chunks, P, L, wave ? wt : 0);
    // (two wavefronts per workgroup at d <= 32: four - fp32: eight - workgroups fit a CU)
    if (panel_path(d, m) && d <= 32) big_partition(B, Tn, chunks, P, L, panel_target_small());
    if (panel_path(d, m))
        return launch_panel(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, add_const, out, ws, info, P, L, ev0, ev1,
                            st);
#define MF_BIG_CASE(DP)                                                                                               \
    return launch_big<DP>(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, add_const, out, ws, info, \
                          P, L, ev0, ev1, st, wave);
    if (d <= 16) { MF_BIG_CASE(16) }
    if (d <= 32) { MF_BIG_CASE(32) }
    if constexpr (sizeof(real) == 4) {
        if (d <= 48) { MF_BIG_CASE(48) }
        if (d <= 64) { MF_BIG_CASE(64) }
    }
#undef MF_BIG_CASE
    return -100;
}

}  // namespace MF_BIG_NS
}  // namespace mf
