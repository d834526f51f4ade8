// Operators of the block-tridiagonal factor for 16 <= d <= 32 with the time axis walked SERIALLY inside a wavefront and the batch
// spread over the chip (mf_wave.hpp's regime: hundreds of series).  The tile engine partitions these operators in time with one
// 256-thread workgroup per chunk - built for config 5's eight series of d = 64; at d = 16, B = 512, T = 1000 its `solve` takes 8.1 ms
// (profiles/r05_bigops_d16.txt) although a block step is two 16 x 16 matrix-vector products.
//
// LowerTriangularBlockTriDiagonal.solve (block_tri_diag.py:339-351): z_k = L_k^-1 (r_k - W_{k-1} z_{k-1}), or with the transpose
// z_k = L_k^-T (r_k - W_k^T z_{k+1}) backwards.  A series occupies NR = 1 (d <= 16) or 2 (d <= 32) of the wavefront's four 16-lane
// rows - a wavefront walks 4 / NR series - with one vector element per lane and the lane's ROW of the step's matrices in
// registers; products and substitutions use the DPP row_newbcast operand of v_fmac (mf_row.hpp's primitive), the two halves of a
// d > 16 vector meet through one ds_bpermute per use.  The transposed solve is the same code on index-reversed rows and columns
// (an upper-triangular system read backwards is a lower-triangular one).  No LDS, no matrix cores: the arithmetic intensity of a
// matrix-vector recursion is 1 flop per 4 bytes.
#pragma once
#include "mf_wave.hpp"

namespace mf {
namespace wv {

template <typename T> struct SolveArgs {
    long Bl, Br, n;          // factors, right-hand sides (series r uses factor r % Bl), blocks
    int d;
    const T *ldiag, *lsub, *rhs;
    T* out;
};

// the lane's row of one step, in the (possibly reversed) coordinates p = 0 .. 16 NR - 1 of its series
template <typename T, int NR> struct SolveRow {
    T m[NR][16];      // M'[p_i][16 c + j], strictly lower part (c == h: j < r; c < h: all of it)
    T c[NR][16];      // coupling C'[p_i][16 c + j]
    T dinv, x;        // 1 / M'[p_i][p_i]; right-hand side element
};

template <typename T, int NR, bool TRANS>
__global__ void __launch_bounds__(64) wave_solve_kernel(SolveArgs<T> a) {
    using D = Dpp<T>;
    constexpr int DP = 16 * NR, NS = 4 / NR;
    const int r = threadIdx.x & 15, q = threadIdx.x >> 4, g = q / NR, h = q % NR;
    const long sr_raw = (long)blockIdx.x * NS + g;
    const bool valid = sr_raw < a.Br;
    const long sr = valid ? sr_raw : a.Br - 1, sl = sr % a.Bl;
    const int d = a.d;
    const long n = a.n, dd = long(d) * d;
    const int pi = 16 * h + r;                              // the lane's position in its series' (reversed) coordinates
    auto idx = [&](int p) { return TRANS ? DP - 1 - p : p; };
    const int i = idx(pi);                                  // logical row / vector element
    const bool row_in = i < d;
    const T* Ld = a.ldiag + sl * n * dd;
    const T* Ls = a.lsub ? a.lsub + sl * (n - 1) * dd : nullptr;
    const T* rh = a.rhs + sr * n * d;
    T* zo = a.out + sr * n * d;

    auto load = [&](long k, SolveRow<T, NR>& s) {
        // block k of the factor; the coupling that brings in the previously solved block (k - 1 forwards, k + 1 backwards)
        const T* Lk = Ld + k * dd;
        const long kc = TRANS ? k : k - 1;
        const bool has_c = Ls != nullptr && (TRANS ? k + 1 < n : k > 0);
        const T* Wk = has_c ? Ls + kc * dd : Ld;
        MF_UNROLL for (int c = 0; c < NR; ++c)
            MF_UNROLL for (int j = 0; j < 16; ++j) {
                const int pj = 16 * c + j, jj = idx(pj);
                const bool in = row_in && jj < d;
                const long off = TRANS ? (long)jj * d + i : (long)i * d + jj;          // M'[pi][pj] = L[i][j] or L[j][i]
                const bool lower = pj < pi;
                const T mv = Lk[(in && lower) ? off : 0];
                s.m[c][j] = (in && lower) ? mv : T(0);
                const T cv = Wk[(in && has_c) ? off : 0];
                s.c[c][j] = (in && has_c) ? cv : T(0);
            }
        const T dg = Lk[row_in ? (long)i * d + i : 0];
        s.dinv = row_in ? t_rcp<T>(dg) : T(1);
        const T xv = rh[k * d + (row_in ? i : 0)];
        s.x = row_in ? xv : T(0);
    };

    T z = T(0);                                             // the previously solved block's element p_i
    SolveRow<T, NR> cur, nxt;
    const long k0 = TRANS ? n - 1 : 0, step = TRANS ? -1 : 1;
    load(k0, cur);
    for (long t = 0; t < n; ++t) {
        const long k = k0 + step * t;
        if (t + 1 < n) load(k + step, nxt);
        __builtin_amdgcn_sched_barrier(0);
        // ---- y = r_k - C' z_prev --------------------------------------------------------------------------------------------
        T zc[NR];                                           // z_prev's half c, as held by the lanes of THIS row (DPP sources)
        if constexpr (NR == 1) zc[0] = z;
        else {
            const T other = __shfl_xor(z, 16, 64);
            zc[0] = h == 0 ? z : other;
            zc[1] = h == 0 ? other : z;
        }
        T y0 = cur.x, y1 = T(0);
        sfor<NR>([&](auto cc) {                             // (compile-time half index: the row arrays must stay in registers)
            constexpr int c = decltype(cc)::value;
            fence1(zc[c]);
            sfor<16>([&](auto j) {
                constexpr int jj = decltype(j)::value;
                if constexpr (jj % 2 == 0) D::template fnmac<jj>(y0, zc[c], cur.c[c][jj]);
                else D::template fnmac<jj>(y1, zc[c], cur.c[c][jj]);
            });
        });
        T y = y0 + y1;
        // ---- forward substitution with M' (strictly lower rows in cur.m, reciprocal diagonal in cur.dinv) ----------------------
        sfor<NR>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            // half c: its 16 unknowns in the rows with h == c; rows of later halves take the finished ones afterwards
            sfor<16>([&](auto j) {
                constexpr int jj = decltype(j)::value;
                T tq = y * cur.dinv;                                  // the candidate z of every lane; lane jj's is final
                fence1(tq);
                const T mj = (h == c) ? cur.m[c][jj] : T(0);
                D::template fnmac<jj>(y, tq, mj);                     // y -= z_jj M'[pi][16 c + jj]   (zero at and above the diagonal)
            });
            if constexpr (NR == 2) {
                if constexpr (c == 0) {
                    // the finished half 0 (rows h == 0) to the rows of half 1, which subtract their off-diagonal block's share
                    const T zfin = y * cur.dinv;
                    T zlow = __shfl_xor(zfin, 16, 64);
                    zlow = h == 1 ? zlow : T(0);
                    fence1(zlow);
                    T acc = T(0);
                    sfor<16>([&](auto j) { D::template fnmac<decltype(j)::value>(acc, zlow, cur.m[0][decltype(j)::value]); });
                    y += (h == 1) ? acc : T(0);
                }
            }
        });
        z = y * cur.dinv;
        if (valid && row_in) zo[k * d + i] = z;
        cur = nxt;
    }
}

}  // namespace wv
}  // namespace mf
