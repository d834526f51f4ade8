// Operators for 16 <= d <= 32 with the time axis walked SERIALLY inside a wavefront and the batch spread over the chip (mf_wave.hpp's
// regime: hundreds of series).  In this file: solve (below; a row per lane, DPP products), then - on mf_wave.hpp's register tiles and
// the matrix cores - cholesky, upper_diagonal_lower + the posterior chain, block_diagonal_of_inverse, marginals / covariance blocks.  The tile engine partitions these operators in time with one
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
    // partitioned in time (wave_solve_up_kernel / wave_solve_boundary_kernel / wave_solve_kernel<.., PART>; Bl == Br): P chunks of Lc
    // blocks in walking order; per (series, chunk) the composed map z_out = M z_in + v, and the z every chunk starts from
    long P, Lc;
    T *wM, *wv, *zin;        // [Br, P, d, d], [Br, P, d], [Br, P, d]
};

// the lane's row of one step, in the (possibly reversed) coordinates p = 0 .. 16 NR - 1 of its series
template <typename T, int NR> struct SolveRow {
    T m[NR][16];      // M'[p_i][16 c + j], strictly lower part (c == h: j < r; c < h: all of it)
    T c[NR][16];      // coupling C'[p_i][16 c + j]
    T dinv, x;        // 1 / M'[p_i][p_i]; right-hand side element
};

// PART: the emit pass of the time-partitioned form - a group of rows is a (series, chunk) pair and starts from the z the boundary pass
// left for its chunk.
template <typename T, int NR, bool TRANS, bool PART = false>
__global__ void __launch_bounds__(64) wave_solve_kernel(SolveArgs<T> a) {
    using D = Dpp<T>;
    constexpr int DP = 16 * NR, NS = 4 / NR;
    const int r = threadIdx.x & 15, q = threadIdx.x >> 4, g = q / NR, h = q % NR;
    const long units = PART ? a.Br * a.P : a.Br;
    const long id_raw = (long)blockIdx.x * NS + g;
    const bool valid = id_raw < units;
    const long id = valid ? id_raw : units - 1;
    const long sr = PART ? id / a.P : id, ch = PART ? id % a.P : 0, sl = sr % a.Bl;
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
    const long t_lo = PART ? ch * a.Lc : 0, t_hi = PART ? ((ch + 1) * a.Lc < n ? (ch + 1) * a.Lc : n) : n;
    if (PART && ch > 0) {
        const T zv = a.zin[(sr * a.P + ch) * d + (row_in ? i : 0)];
        z = row_in ? zv : T(0);
    }
    load(k0 + step * t_lo, cur);
    for (long t = t_lo; t < t_hi; ++t) {
        const long k = k0 + step * t;
        if (t + 1 < t_hi) load(k + step, nxt);
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

// The composed map of a chunk of the substitution, on the register tiles: in walking order z_t = B_t z_{t-1} + a_t with
//   forwards   B = -L_k^-1 W_{k-1},   a = L_k^-1 r_k;        transposed   B = -L_k^-T W_k^T,   a = L_k^-T r_k,
// so M <- B M, v <- B v + a block after block: one triangular inversion (DPP), two products and two matrix-vector products per
// block.  Every product is P^T Q: forwards W M = tn(W^T, M) with W^T read transposed and L^-1 X = tn(L^-T, X) with L^-T through the
// LDS image; transposed W^T M = tn(W, M), L^-T X = tn(L^-1, X) as they are.  The last chunk's map is not needed.
template <typename T, int NT, bool TRANS>
__global__ void __launch_bounds__(64) wave_solve_up_kernel(SolveArgs<T> a) {
    using v4 = typename Tr<T>::v4;
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x / (a.P - 1), c = blockIdx.x % (a.P - 1), n = a.n;
    const long t_lo = c * a.Lc, t_hi = (c + 1) * a.Lc < n ? (c + 1) * a.Lc : n;
    int d = a.d;
    const long dd = long(d) * d;
    const T* Ld = a.ldiag + s * n * dd;
    const T* Ls = a.lsub + s * (n - 1) * dd;
    const T* rh = a.rhs + s * n * d;
    LogAcc<T> la;
    la.init();
    bool bad = false;
    Mat<T, NT> M;
    CV<T, NT> v;
    identity_mat<T, NT>(M, ln);
    MF_UNROLL for (int j = 0; j < NT; ++j) v.v[j] = T(0);
    for (long t = t_lo; t < t_hi; ++t) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        const long k = TRANS ? n - 1 - t : t;
        const bool has_c = t > 0;
        const long kc = TRANS ? k : k - 1;
        Mat<T, NT> L, Li, W, X;
        v4 c10t = {0, 0, 0, 0};
        CV<T, NT> rk, y;
        load_mat<T, NT, S_LOWER>(L, Ld + k * dd, d, true, true, ln);
        if constexpr (NT == 2) load_tile_t<T>(c10t, Ld + k * dd, d, 1, 0, ln);
        if (!has_c) W.zero();
        else if (TRANS) load_mat<T, NT, S_FULL>(W, Ls + kc * dd, d, false, false, ln);
        else load_mat_t<T, NT>(W, Ls + kc * dd, d, ln);                           // W^T
        load_cv<T, NT>(rk, rh + k * d, d, ln);
        phase();
        tri_inv_mat<T, NT>(L, c10t, Li, lds, ln, la, bad);
        if constexpr (!TRANS) {
            L.zero();
            transpose<T, NT, S_LOWER>(L, Li, lds, ln);                            // L^-T (upper) in L
        }
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, W, M);                           // W M | W^T M
        if constexpr (TRANS) tn<T, NT, S_LOWER, S_FULL, S_FULL, OP_NEG>(M, Li, X);    // -L^-T X
        else tn<T, NT, S_UPPER, S_FULL, S_FULL, OP_NEG>(M, L, X);                     // -L^-1 X
        RV<T, NT> vr;
        cv_to_rv<T, NT>(vr, v, ln);
        tn_mv<T, NT, S_FULL>(y, W, vr);                                               // W v | W^T v
        MF_UNROLL for (int j = 0; j < NT; ++j) rk.v[j] -= y.v[j];
        cv_to_rv<T, NT>(vr, rk, ln);
        if constexpr (TRANS) tn_mv<T, NT, S_LOWER>(v, Li, vr);
        else tn_mv<T, NT, S_UPPER>(v, L, vr);
    }
    const long id = s * a.P + c;
    store_mat<T, NT, false>(a.wM + id * dd, M, d, lds, ln);
    store_cv<T, NT>(a.wv + id * d, v, d, ln);
    (void)bad;
}
// ---- StateSpaceModel.marginal_means partitioned in time: the maps of the chunks on the register tiles -----------------------------
// m_t = A_{t-1} m_{t-1} + o_t (m_0 = o_0).  Chunk c carries position c Lc to (c + 1) Lc: M <- A M, v <- A v + o (one product and one
// matrix-vector product per block, A read transposed: A M = tn(A^T, M)); then m_in(c + 1) = M_c m_in(c) + v_c, a wavefront per series.
// The walk per chunk from m_in(c) is bigop_means_kernel (mf_bigops_impl.hpp: a row of A per lane, the mean by v_readlane).
template <typename T> struct MeansArgs {
    long B, n;
    int d;
    const T *A, *offs;       // offs [B, n, d]: o_0 = mu0, o_t = b_{t-1}; or NULL and (mu0 [B, d], b [B, n - 1, d]) below
    long P, Lc;
    T *wM, *wv, *m_in;       // [B, P, d, d], [B, P, d], [B, P, d]
    const T *mu0, *b;
    MF_DEV const T* off(long s, long t) const {
        return offs ? offs + (s * n + t) * d : (t == 0 ? mu0 + s * d : b + (s * (n - 1) + t - 1) * d);
    }
};
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_means_up_kernel(MeansArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x / (a.P - 1), c = blockIdx.x % (a.P - 1), n = a.n;
    const long t_lo = c * a.Lc + 1, t_hi = (c + 1) * a.Lc + 1 < n ? (c + 1) * a.Lc + 1 : n;
    int d = a.d;
    const long dd = long(d) * d;
    const T* Ag = a.A + s * (n - 1) * dd;
    Mat<T, NT> M, AT;
    CV<T, NT> v, ok;
    identity_mat<T, NT>(M, ln);
    MF_UNROLL for (int j = 0; j < NT; ++j) v.v[j] = T(0);
    if (t_hi > t_lo) {
        load_mat_t<T, NT>(AT, Ag + (t_lo - 1) * dd, d, ln);
        load_cv<T, NT>(ok, a.off(s, t_lo), d, ln);
    }
    for (long t = t_lo; t < t_hi; ++t) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        Mat<T, NT> ATn, X;
        CV<T, NT> on, y;
        const long tn_ = t + 1 < t_hi ? t + 1 : t;
        load_mat_t<T, NT>(ATn, Ag + (tn_ - 1) * dd, d, ln);
        load_cv<T, NT>(on, a.off(s, tn_), d, ln);
        phase();
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, AT, M);                           // A M
        M = X;
        RV<T, NT> vr;
        cv_to_rv<T, NT>(vr, v, ln);
        tn_mv<T, NT, S_FULL>(y, AT, vr);                                               // A v
        MF_UNROLL for (int j = 0; j < NT; ++j) v.v[j] = y.v[j] + ok.v[j];
        AT = ATn;
        MF_UNROLL for (int j = 0; j < NT; ++j) ok.v[j] = on.v[j];
    }
    const long id = s * a.P + c;
    store_mat<T, NT, false>(a.wM + id * dd, M, d, lds, ln);
    store_cv<T, NT>(a.wv + id * d, v, d, ln);
}
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_means_boundary_kernel(MeansArgs<T> a) {
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x;
    const int d = a.d;
    const long dd = long(d) * d;
    CV<T, NT> m;
    load_cv<T, NT>(m, a.off(s, 0), d, ln);
    store_cv<T, NT>(a.m_in + (s * a.P) * d, m, d, ln);
    for (long c = 0; c + 1 < a.P; ++c) {
        const long id = s * a.P + c;
        Mat<T, NT> MT;
        CV<T, NT> vc, y;
        RV<T, NT> mr;
        load_mat_t<T, NT>(MT, a.wM + id * dd, d, ln);
        load_cv<T, NT>(vc, a.wv + id * d, d, ln);
        cv_to_rv<T, NT>(mr, m, ln);
        tn_mv<T, NT, S_FULL>(y, MT, mr);                                               // M m
        MF_UNROLL for (int j = 0; j < NT; ++j) m.v[j] = y.v[j] + vc.v[j];
        store_cv<T, NT>(a.m_in + (id + 1) * d, m, d, ln);
    }
}

// the z every chunk starts from: z_in(c + 1) = M_c z_in(c) + v_c, z_in(0) = 0 - a wavefront per series
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_solve_boundary_kernel(SolveArgs<T> a) {
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x;
    const int d = a.d;
    const long dd = long(d) * d;
    CV<T, NT> z;
    MF_UNROLL for (int j = 0; j < NT; ++j) z.v[j] = T(0);
    for (long c = 0; c + 1 < a.P; ++c) {
        const long id = s * a.P + c;
        Mat<T, NT> MT;
        CV<T, NT> vc, y;
        RV<T, NT> zr;
        load_mat_t<T, NT>(MT, a.wM + id * dd, d, ln);
        load_cv<T, NT>(vc, a.wv + id * d, d, ln);
        cv_to_rv<T, NT>(zr, z, ln);
        tn_mv<T, NT, S_FULL>(y, MT, zr);                                               // M z
        MF_UNROLL for (int j = 0; j < NT; ++j) z.v[j] = y.v[j] + vc.v[j];
        store_cv<T, NT>(a.zin + (id + 1) * d, z, d, ln);
    }
}

}  // namespace wv
}  // namespace mf

// =====================================================================================================================================
// The factorisations with the time axis walked serially by ONE WAVEFRONT PER SERIES on mf_wave.hpp's register tiles: every product in
// the P^T Q form on the matrix cores, the 16 x 16 diagonal tiles factored inside the wave, the next block's operands in flight while
// the current one is worked.  For hundreds of series of 16 <= d <= 32 (the tile engine partitions these in time with one 256-thread
// workgroup per chunk: cholesky 5.4 / 26.7 ms, block_diagonal_of_inverse 8.2 / 25.4 ms at B = 512, T = 1000, d = 16 / 32 in fp64 -
// profiles/r05_bigops_d16.txt).
namespace mf {
namespace wv {

// P (symmetric positive definite tile, lower triangle used) -> L = chol(P) and LiT = L^-T as tiles; the image holds L^-1 afterwards
template <typename T>
MF_DEV void chol_fact_tile(const typename Tr<T>::v4& P, typename Tr<T>::v4& L, typename Tr<T>::v4& LiT, T* img, const Lane& ln,
                           bool& bad) {
    using D = Dpp<T>;
    using v4 = typename Tr<T>::v4;
    T a[16], x[16];
    const v4 in[1] = {P};
    rows_in<T, 1>(in, img, a, ln);
    sfor<16>([&](auto i) { x[decltype(i)::value] = (ln.r == decltype(i)::value) ? T(1) : T(0); });
    sfor<16>([&](auto j) {
        constexpr int jj = decltype(j)::value;
        fence1(a[jj]);
        const T s = D::template bcast<jj>(a[jj]);
        bad |= !(s > T(0));
        const T inv = row::row_rsqrt(s);
        a[jj] *= inv;                  // L[r][j]
        x[jj] *= inv;                  // Li[j][r]
        fence1(a[jj]);
        sfor2<jj + 1, 16>([&](auto k) {
            constexpr int kk = decltype(k)::value;
            D::template fnmac<kk>(a[kk], a[jj], a[jj]);
            D::template fnmac<kk>(x[kk], a[jj], x[jj]);
        });
    });
    if (ln.q == 0) {                   // the rows of L (the lane's entries right of the diagonal are rounding residue: zero)
        MF_UNROLL for (int j = 0; j < 16; ++j) img[ln.r * Tr<T>::LD + j] = (j <= ln.r) ? a[j] : T(0);
    }
    lds_fence();
    MF_UNROLL for (int e = 0; e < 4; ++e) L[e] = img[Tr<T>::row(ln.q, e) * Tr<T>::LD + ln.r];
    lds_fence();
    v4 out[1];
    cols_out<T, 1, true>(x, out, img, ln);
    LiT = out[0];
}
// Phi (symmetric, tiles ti <= tj valid; consumed) -> L = chol(Phi) (S_LOWER) and LiT = L^-T (S_UPPER)
template <typename T, int NT>
MF_DEV void chol_fact_mat(Mat<T, NT>& Phi, Mat<T, NT>& L, Mat<T, NT>& LiT, T* lds, const Lane& ln, bool& bad) {
    using v4 = typename Tr<T>::v4;
    chol_fact_tile<T>(Phi.t[0][0], L.t[0][0], LiT.t[0][0], lds, ln, bad);
    if constexpr (NT == 2) {
        v4 li00, lt01 = {0, 0, 0, 0}, acc = {0, 0, 0, 0}, z = {0, 0, 0, 0}, h = {0, 0, 0, 0};
        MF_UNROLL for (int e = 0; e < 4; ++e) li00[e] = lds[Tr<T>::row(ln.q, e) * Tr<T>::LD + ln.r];            // L00^-1 from the image
        lds_fence();
        MF_UNROLL for (int e = 0; e < 4; ++e) lt01 = Tr<T>::mfma(LiT.t[0][0][e], Phi.t[0][1][e], lt01);      // (L10)^T = Li00 Phi01
        MF_UNROLL for (int e = 0; e < 4; ++e) acc = Tr<T>::mfma(lt01[e], lt01[e], acc);                      // L10 L10^T
        Phi.t[1][1] -= acc;
        chol_fact_tile<T>(Phi.t[1][1], L.t[1][1], LiT.t[1][1], lds, ln, bad);
        MF_UNROLL for (int e = 0; e < 4; ++e) z = Tr<T>::mfma(lt01[e], li00[e], z);                          // Z = L10 Li00
        MF_UNROLL for (int e = 0; e < 4; ++e) h = Tr<T>::mfma(z[e], LiT.t[1][1][e], h);                      // Z^T Li11^T
        LiT.t[0][1] = -h;
        LiT.t[1][0] = v4{0, 0, 0, 0};
        transpose_tile<T>(L.t[1][0], lt01, lds, ln);
        L.t[0][1] = v4{0, 0, 0, 0};
    }
}
// Two SPD tiles factored side by side (rows [0, 2) of the wavefront: PA, rows [2, 4): PB): one instruction stream.  Out: the factors
// and the transposed inverses of both; the images hold LA^-1 and LB^-1 (row-major) afterwards.  img: two tile images.
template <typename T>
MF_DEV void chol_fact_tiles2(const typename Tr<T>::v4& PA, const typename Tr<T>::v4& PB, typename Tr<T>::v4& LA, typename Tr<T>::v4& LiTA,
                             typename Tr<T>::v4& LB, typename Tr<T>::v4& LiTB, T* img, const Lane& ln, bool& bad) {
    using D = Dpp<T>;
    using v4 = typename Tr<T>::v4;
    constexpr int TS = 16 * Tr<T>::LD;
    T a[16], x[16];
    const v4 in[2] = {PA, PB};
    rows_in<T, 2>(in, img, a, ln);
    sfor<16>([&](auto i) { x[decltype(i)::value] = (ln.r == decltype(i)::value) ? T(1) : T(0); });
    sfor<16>([&](auto j) {
        constexpr int jj = decltype(j)::value;
        fence1(a[jj]);
        const T s = D::template bcast<jj>(a[jj]);
        bad |= !(s > T(0));
        const T inv = row::row_rsqrt(s);
        a[jj] *= inv;
        x[jj] *= inv;
        fence1(a[jj]);
        sfor2<jj + 1, 16>([&](auto k) {
            constexpr int kk = decltype(k)::value;
            D::template fnmac<kk>(a[kk], a[jj], a[jj]);
            D::template fnmac<kk>(x[kk], a[jj], x[jj]);
        });
    });
    T* mine = img + (ln.q >> 1) * TS;
    if ((ln.q & 1) == 0) {
        MF_UNROLL for (int j = 0; j < 16; ++j) mine[ln.r * Tr<T>::LD + j] = (j <= ln.r) ? a[j] : T(0);
    }
    lds_fence();
    MF_UNROLL for (int e = 0; e < 4; ++e) {
        LA[e] = img[Tr<T>::row(ln.q, e) * Tr<T>::LD + ln.r];
        LB[e] = img[TS + Tr<T>::row(ln.q, e) * Tr<T>::LD + ln.r];
    }
    lds_fence();
    if ((ln.q & 1) == 0) {
        MF_UNROLL for (int row = 0; row < 16; ++row) mine[row * Tr<T>::LD + ln.r] = x[row];
    }
    lds_fence();
    image_to_tile_t<T>(LiTA, img, ln);
    image_to_tile_t<T>(LiTB, img + TS, ln);
    lds_fence();
}
// A (symmetric, tiles ti <= tj valid; consumed) -> LA = chol(A) (S_LOWER), LiTA = LA^-T (S_UPPER);  B likewise -> LB only.
// The diagonal tiles of the two matrices share their passes (two instead of four at 2 x 2 tiles, one instead of two at one tile).
template <typename T, int NT>
MF_DEV void chol_fact_mat2(Mat<T, NT>& A, Mat<T, NT>& B, Mat<T, NT>& LA, Mat<T, NT>& LiTA, Mat<T, NT>& LB, T* lds, const Lane& ln, bool& bad) {
    using v4 = typename Tr<T>::v4;
    v4 litb00;
    chol_fact_tiles2<T>(A.t[0][0], B.t[0][0], LA.t[0][0], LiTA.t[0][0], LB.t[0][0], litb00, lds, ln, bad);
    if constexpr (NT == 2) {
        constexpr int TS = 16 * Tr<T>::LD;
        v4 lia00, lta = {0, 0, 0, 0}, ltb = {0, 0, 0, 0}, acc = {0, 0, 0, 0}, z = {0, 0, 0, 0}, h = {0, 0, 0, 0}, unused;
        MF_UNROLL for (int e = 0; e < 4; ++e) lia00[e] = lds[Tr<T>::row(ln.q, e) * Tr<T>::LD + ln.r];          // LA00^-1 from the image
        lds_fence();
        MF_UNROLL for (int e = 0; e < 4; ++e) lta = Tr<T>::mfma(LiTA.t[0][0][e], A.t[0][1][e], lta);         // (LA10)^T
        MF_UNROLL for (int e = 0; e < 4; ++e) acc = Tr<T>::mfma(lta[e], lta[e], acc);
        A.t[1][1] -= acc;
        acc = v4{0, 0, 0, 0};
        MF_UNROLL for (int e = 0; e < 4; ++e) ltb = Tr<T>::mfma(litb00[e], B.t[0][1][e], ltb);               // (LB10)^T
        MF_UNROLL for (int e = 0; e < 4; ++e) acc = Tr<T>::mfma(ltb[e], ltb[e], acc);
        B.t[1][1] -= acc;
        chol_fact_tiles2<T>(A.t[1][1], B.t[1][1], LA.t[1][1], LiTA.t[1][1], LB.t[1][1], unused, lds, ln, bad);
        MF_UNROLL for (int e = 0; e < 4; ++e) z = Tr<T>::mfma(lta[e], lia00[e], z);                          // Z = LA10 LA00^-1
        MF_UNROLL for (int e = 0; e < 4; ++e) h = Tr<T>::mfma(z[e], LiTA.t[1][1][e], h);                     // Z^T LA11^-T
        LiTA.t[0][1] = -h;
        LiTA.t[1][0] = v4{0, 0, 0, 0};
        lds_fence();                                                                                         // both transposes in one round trip
        tile_to_image<T>(lta, lds, ln);
        tile_to_image<T>(ltb, lds + TS, ln);
        lds_fence();
        image_to_tile_t<T>(LA.t[1][0], lds, ln);
        image_to_tile_t<T>(LB.t[1][0], lds + TS, ln);
        lds_fence();
        LA.t[0][1] = v4{0, 0, 0, 0};
        LB.t[0][1] = v4{0, 0, 0, 0};
    }
}
// a symmetric block from its LOWER triangle (what the reference's banded Cholesky reads): tiles ti <= tj
template <typename T, int NT> MF_DEV void load_sym_lower(Mat<T, NT>& m, const T* __restrict__ g, int d, const Lane& ln) {
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        MF_UNROLL for (int e = 0; e < 4; ++e) {
            const int i = 16 * ti + Tr<T>::row(ln.q, e), j = 16 * ti + ln.r;
            const bool in = i < d && j < d;
            const T v = g[in ? i * d + j : 0];
            m.t[ti][ti][e] = in ? v : ((i == j) ? T(1) : T(0));
        }
    if constexpr (NT == 2) {
        load_tile_t<T>(m.t[0][1], g, d, 1, 0, ln);
        m.t[1][0] = typename Tr<T>::v4{0, 0, 0, 0};
    }
}
template <typename T, int NT> MF_DEV void copy_cv(CV<T, NT>& o, const CV<T, NT>& i) {
    MF_UNROLL for (int j = 0; j < NT; ++j) o.v[j] = i.v[j];
}

template <typename T> struct FactArgs {
    long B, n;
    int d;
    const T *diag, *sub;     // the symmetric matrix (cholesky, udl) or the factor (inverse blocks)
    T *o1, *o2;              // cholesky: L diag, L sub; udl: U^T sub, chol(Delta); inverse blocks: diagonal, sub-diagonal blocks
    const T* eta;            // udl: information vector (posterior chain) or NULL
    T *m_post, *chol_dinv;   // udl with eta: Delta_k^-1 x_k and chol(Delta_k^-1)
    int* info;
    // the posterior chain partitioned in time (wave_udl_up_kernel / wave_udl_boundary_kernel / wave_udl_kernel<.., PART>): P chunks of
    // L blocks per series, counted from the LAST block; per (series, chunk): the reduced system the up-sweep leaves and the boundary
    // values the emit pass starts from
    long P, L;
    T *rDv, *rGU, *rF, *rtv, *rgU;   // [B, P, d, d] x 3, [B, P, d] x 2
    T *bSig, *bx;                     // [B, P, d, d], [B, P, d]: natural-order pivot Delta and x at the chunk's last (lowest) block
};

// SymmetricBlockTriDiagonal.cholesky (block_tri_diag.py:423-436): L_k = chol(D_k - W_{k-1} W_{k-1}^T), W_k = S_k L_k^-T.
// The wave keeps W^T: D_k - (W^T)^T (W^T) and W_k^T = (LiT)^T S_k^T are both P^T Q products.
// PART: the emit pass of the time-partitioned form (wave_chol_up_kernel, wave_udl_boundary_kernel): wavefront (s, c) walks the blocks
// [c L, (c + 1) L) and, for c > 0, starts from the natural-order pivot of the block in front of its chunk.
template <typename T, int NT, bool PART = false>
__global__ void __launch_bounds__(64) wave_cholesky_kernel(FactArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = PART ? blockIdx.x / a.P : blockIdx.x, c = PART ? blockIdx.x % a.P : 0, n = a.n;
    const long k_lo = PART ? c * a.L : 0, k_hi = PART ? ((c + 1) * a.L < n ? (c + 1) * a.L : n) - 1 : n - 1;
    int d = a.d;
    const long dd = long(d) * d;
    const T* Dg = a.diag + s * n * dd;
    const T* Sg = a.sub ? a.sub + s * (n - 1) * dd : nullptr;
    bool bad = false;
    Mat<T, NT> Dk, WT;
    load_sym_lower<T, NT>(Dk, Dg + k_lo * dd, d, ln);
    WT.zero();
    if constexpr (PART) {
        if (c > 0) {   // W^T of the block in front of the chunk, from its natural-order pivot
            Mat<T, NT> Sg0, L0, LiT0, ST0;
            load_sym_lower<T, NT>(Sg0, a.bSig + (s * a.P + c - 1) * dd, d, ln);
            load_mat_t<T, NT>(ST0, Sg + (k_lo - 1) * dd, d, ln);
            chol_fact_mat<T, NT>(Sg0, L0, LiT0, lds, ln, bad);
            tn<T, NT, S_UPPER, S_FULL, S_FULL, OP_SET>(WT, LiT0, ST0);
        }
    }
    for (long k = k_lo; k <= k_hi; ++k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        const bool more = k < k_hi, coupled = k + 1 < n && Sg != nullptr;
        Mat<T, NT> Dn, ST;
        if (coupled) load_mat_t<T, NT>(ST, Sg + k * dd, d, ln);
        if (more) load_sym_lower<T, NT>(Dn, Dg + (k + 1) * dd, d, ln);
        phase();
        if (k > 0 && Sg) tn<T, NT, S_FULL, S_FULL, S_UPPER, OP_SUB>(Dk, WT, WT);
        Mat<T, NT> L, LiT;
        chol_fact_mat<T, NT>(Dk, L, LiT, lds, ln, bad);
        store_mat<T, NT, false>(a.o1 + (s * n + k) * dd, L, d, lds, ln);
        phase();
        if (coupled) {
            tn<T, NT, S_UPPER, S_FULL, S_FULL, OP_SET>(WT, LiT, ST);                  // W_k^T = L_k^-1 S_k^T
            Mat<T, NT> W;
            transpose<T, NT, S_FULL>(W, WT, lds, ln);
            store_mat<T, NT, false>(a.o2 + (s * (n - 1) + k) * dd, W, d, lds, ln);
        }
        if (more) Dk = Dn;
    }
    if (__any(bad) && threadIdx.x == 0 && a.info) raise_info(a.info);
}
// up-sweep of the time-partitioned Cholesky factorisation (natural order: chunk c = blocks [c L, (c + 1) L), the block in front of it
// is its separator; the scheme of wave_udl_up_kernel below, no right-hand side): Dv, GU, F per chunk
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_chol_up_kernel(FactArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x / a.P, c = blockIdx.x % a.P, n = a.n;
    const long k_lo = c * a.L, k_hi = ((c + 1) * a.L < n ? (c + 1) * a.L : n) - 1;
    int d = a.d;
    const long dd = long(d) * d;
    const T* Dg = a.diag + s * n * dd;
    const T* Sg = a.sub + s * (n - 1) * dd;
    const bool spike = c > 0;
    WaveElim<T, NT> E;
    E.init();
    Mat<T, NT> Dk, ST;
    CV<T, NT> zero;
    MF_UNROLL for (int j = 0; j < NT; ++j) zero.v[j] = T(0);
    load_sym_lower<T, NT>(Dk, Dg + k_lo * dd, d, ln);
    if (spike) load_mat<T, NT, S_FULL>(E.X, Sg + (k_lo - 1) * dd, d, false, false, ln);   // block (k_lo, k_lo - 1)
    MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = i; j < NT; ++j) E.Phi.t[i][j] = Dk.t[i][j];
    for (long k = k_lo + 1; k <= k_hi; ++k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        load_sym_lower<T, NT>(Dk, Dg + k * dd, d, ln);
        load_mat_t<T, NT>(ST, Sg + (k - 1) * dd, d, ln);                               // S_{k-1}^T: block (k - 1, k)
        phase();
        WaveFact<T, NT> f;
        if (spike) E.template eliminate<true>(f, lds, ln); else E.template eliminate<false>(f, lds, ln);
        phase();
        Mat<T, NT> WT;
        tn<T, NT, S_UPPER, S_FULL, S_FULL, OP_SET>(WT, f.LiT, ST);                     // W^T = L^-1 S_{k-1}^T
        phase();
        if (spike) E.template advance<true>(f, WT, Dk, zero); else E.template advance<false>(f, WT, Dk, zero);
    }
    const long id = s * a.P + c;
    store_mat<T, NT, true>(a.rDv + id * dd, E.Phi, d, lds, ln);
    store_mat<T, NT, true>(a.rGU + id * dd, E.GU, d, lds, ln);
    store_mat<T, NT, false>(a.rF + id * dd, E.X, d, lds, ln);
    if (__any(E.bad) && threadIdx.x == 0 && a.info) raise_info(a.info);
}

// upper_diagonal_lower (block_tri_diag.py:438-545) + the posterior chain's means and factors (kalman_filter.py:159-174), backwards:
// Delta_k = D_k - S_k^T Delta_{k+1}^-1 S_k, U_k^T = Delta_{k+1}^-1 S_k, x_k = eta_k - U_k x_{k+1}, m_k = Delta_k^-1 x_k.
// ETA: the information vector rides along (posterior chain); without it the kernel is the plain factorisation - two instantiations,
// not one loop body with both (the longer body cost the plain one 12 %: instruction fetch)
// PART: the emit pass of the time-partitioned form - wavefront (s, c) walks the blocks [k_lo, k_hi] of its chunk (chunks count from the
// last block) and, for c > 0, starts from the boundary pivot and x of the chunk above it (wave_udl_boundary_kernel).
template <typename T, int NT, bool ETA, bool PART = false>
__global__ void __launch_bounds__(64) wave_udl_kernel(FactArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[(NT == 1 ? 2 : NT * NT) * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = PART ? blockIdx.x / a.P : blockIdx.x, c = PART ? blockIdx.x % a.P : 0, n = a.n;
    const long k_hi = PART ? n - 1 - c * a.L : n - 1;
    const long k_lo = PART ? (n - (c + 1) * a.L > 0 ? n - (c + 1) * a.L : 0) : 0;
    int d = a.d;
    const long dd = long(d) * d;
    const T* Dg = a.diag + s * n * dd;
    const T* Sg = a.sub ? a.sub + s * (n - 1) * dd : nullptr;
    const T* Eg = ETA ? a.eta + s * n * d : nullptr;
    bool bad = false;
    Mat<T, NT> Dk, Sk, LiT, Li, Qm;
    CV<T, NT> xk, xp;
    load_sym_lower<T, NT>(Dk, Dg + k_hi * dd, d, ln);
    identity_mat<T, NT>(Qm, ln);
    Sk.zero();
    LiT.zero();
    Li.zero();
    MF_UNROLL for (int j = 0; j < NT; ++j) xk.v[j] = xp.v[j] = T(0);
    if constexpr (ETA) load_cv<T, NT>(xk, Eg + k_hi * d, d, ln);
    if constexpr (PART) {
        if (c > 0) {   // the state the serial walk would have reached above this chunk: factor of Delta_{k_hi + 1}, x_{k_hi + 1}
            Mat<T, NT> Sg0, L0;
            load_sym_lower<T, NT>(Sg0, a.bSig + (s * a.P + c - 1) * dd, d, ln);
            chol_fact_mat<T, NT>(Sg0, L0, LiT, lds, ln, bad);
            transpose<T, NT, S_UPPER>(Li, LiT, lds, ln);
            if constexpr (NT == 2) Li.t[0][1] = typename Tr<T>::v4{0, 0, 0, 0};
            load_mat<T, NT, S_FULL>(Sk, Sg + k_hi * dd, d, false, false, ln);
            if constexpr (ETA) load_cv<T, NT>(xp, a.bx + (s * a.P + c - 1) * d, d, ln);
        }
    }
    for (long k = k_hi; k >= k_lo; --k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        const bool more = k > k_lo, coupled = k + 1 < n && Sg != nullptr;
        Mat<T, NT> Dn, Sn;
        CV<T, NT> xn;
        if (more) {
            load_sym_lower<T, NT>(Dn, Dg + (k - 1) * dd, d, ln);
            if (Sg) load_mat<T, NT, S_FULL>(Sn, Sg + (k - 1) * dd, d, false, false, ln);
            if constexpr (ETA) load_cv<T, NT>(xn, Eg + (k - 1) * d, d, ln);
        }
        phase();
        if (coupled) {
            Mat<T, NT> U, Ut;
            tn<T, NT, S_UPPER, S_FULL, S_FULL, OP_SET>(U, LiT, Sk);                    // L^-1 S_k       (L = chol Delta_{k+1})
            tn<T, NT, S_FULL, S_FULL, S_UPPER, OP_SUB>(Dk, U, U);                      // Delta_k
            tn<T, NT, S_LOWER, S_FULL, S_FULL, OP_SET>(Ut, Li, U);                     // U_k^T = Delta_{k+1}^-1 S_k
            store_mat<T, NT, false>(a.o1 + (s * (n - 1) + k) * dd, Ut, d, lds, ln);
            if constexpr (ETA) {
                RV<T, NT> xr;
                CV<T, NT> t;
                cv_to_rv<T, NT>(xr, xp, ln);
                tn_mv<T, NT, S_FULL>(t, Ut, xr);
                MF_UNROLL for (int j = 0; j < NT; ++j) xk.v[j] -= t.v[j];
            }
        }
        phase();
        {
            Mat<T, NT> L;
            if constexpr (ETA) {
                // the factor of Delta_{k+1}^-1 (the previous step's, an identity at the first) rides along in the other rows
                Mat<T, NT> Lq;
                chol_fact_mat2<T, NT>(Dk, Qm, L, LiT, Lq, lds, ln, bad);
                if (k < k_hi) store_mat<T, NT, false>(a.chol_dinv + (s * n + k + 1) * dd, Lq, d, lds, ln);
            } else {
                chol_fact_mat<T, NT>(Dk, L, LiT, lds, ln, bad);
            }
            store_mat<T, NT, false>(a.o2 + (s * n + k) * dd, L, d, lds, ln);
        }
        transpose<T, NT, S_UPPER>(Li, LiT, lds, ln);
        if constexpr (NT == 2) Li.t[0][1] = typename Tr<T>::v4{0, 0, 0, 0};
        phase();
        if constexpr (ETA) {
            RV<T, NT> r;
            CV<T, NT> w, mk;
            cv_to_rv<T, NT>(r, xk, ln);
            tn_mv<T, NT, S_UPPER>(w, LiT, r);                                          // L^-1 x_k
            cv_to_rv<T, NT>(r, w, ln);
            tn_mv<T, NT, S_LOWER>(mk, Li, r);                                          // L^-T (L^-1 x_k)
            store_cv<T, NT>(a.m_post + (s * n + k) * d, mk, d, ln);
            tn<T, NT, S_LOWER, S_LOWER, S_UPPER, OP_SET>(Qm, Li, Li);                  // Delta_k^-1: factored beside Delta_{k-1}
            copy_cv<T, NT>(xp, xk);
        }
        if (more) {
            Dk = Dn;
            if (Sg) Sk = Sn;
            if constexpr (ETA) copy_cv<T, NT>(xk, xn);
        }
    }
    if constexpr (ETA) {   // the last one has nothing left to ride beside
        Mat<T, NT> Lq, LqiT;
        chol_fact_mat<T, NT>(Qm, Lq, LqiT, lds, ln, bad);
        store_mat<T, NT, false>(a.chol_dinv + (s * n + k_lo) * dd, Lq, d, lds, ln);
    }
    if (__any(bad) && threadIdx.x == 0 && a.info) raise_info(a.info);
}

// ---- the posterior chain partitioned in time (hundreds of series leave half the SIMDs without a wavefront and every step's latency
// exposed: cutting each series into P chunks multiplies the wavefronts) -------------------------------------------------------------
// In the order the U D U^T sweep visits the blocks (from the last one down) the matrix is an ordinary block tridiagonal one whose
// coupling of block k with its predecessor k + 1 is S_k^T.  Up-sweep: chunk c eliminates its blocks in that order with the fill-in
// towards the block above it (the separator) carried as a spike - WaveElim, the log-likelihood's elimination state - and leaves, for
// its LOWEST block e: Dv (pivot given the chunk's interior), F (coupling e <-> separator), GU / gU (what the interior adds to the
// separator's pivot / right-hand side), tv (its right-hand side).  Boundary: Delta_e = Dv - F (Delta_sep + GU)^-1 F^T and
// x_e = tv - F (Delta_sep + GU)^-1 (x_sep + gU) chunk after chunk (the quotient property of Schur complements: GU and gU are the parts
// of the separator that the natural order does NOT have yet).  Emit: wave_udl_kernel<.., PART> from those boundary values.
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_udl_up_kernel(FactArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x / a.P, c = blockIdx.x % a.P, n = a.n;
    const long k_hi = n - 1 - c * a.L, k_lo = n - (c + 1) * a.L > 0 ? n - (c + 1) * a.L : 0;
    int d = a.d;
    const long dd = long(d) * d;
    const T* Dg = a.diag + s * n * dd;
    const T* Sg = a.sub + s * (n - 1) * dd;
    const T* Eg = a.eta ? a.eta + s * n * d : nullptr;              // NULL: the plain factorisation, no right-hand side
    const bool spike = c > 0;
    WaveElim<T, NT> E;
    E.init();
    Mat<T, NT> Dk, Sk;
    CV<T, NT> ek;
    load_sym_lower<T, NT>(Dk, Dg + k_hi * dd, d, ln);
    MF_UNROLL for (int j = 0; j < NT; ++j) ek.v[j] = T(0);
    if (Eg) load_cv<T, NT>(ek, Eg + k_hi * d, d, ln);
    if (spike) load_mat_t<T, NT>(E.X, Sg + k_hi * dd, d, ln);                       // block (k_hi, k_hi + 1) = S_{k_hi}^T
    MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = i; j < NT; ++j) E.Phi.t[i][j] = Dk.t[i][j];
    E.t = ek;
    for (long k = k_hi - 1; k >= k_lo; --k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        load_sym_lower<T, NT>(Dk, Dg + k * dd, d, ln);
        load_mat<T, NT, S_FULL>(Sk, Sg + k * dd, d, false, false, ln);              // S_k: block (k + 1, k)
        if (Eg) load_cv<T, NT>(ek, Eg + k * d, d, ln);
        phase();
        WaveFact<T, NT> f;
        if (spike) E.template eliminate<true>(f, lds, ln); else E.template eliminate<false>(f, lds, ln);
        phase();
        Mat<T, NT> WT;
        tn<T, NT, S_UPPER, S_FULL, S_FULL, OP_SET>(WT, f.LiT, Sk);                  // W^T = L^-1 S_k   (W = S_k^T L^-T)
        phase();
        if (spike) E.template advance<true>(f, WT, Dk, ek); else E.template advance<false>(f, WT, Dk, ek);
    }
    const long id = s * a.P + c;
    store_mat<T, NT, true>(a.rDv + id * dd, E.Phi, d, lds, ln);
    store_mat<T, NT, true>(a.rGU + id * dd, E.GU, d, lds, ln);
    store_mat<T, NT, false>(a.rF + id * dd, E.X, d, lds, ln);
    if (Eg) {
        store_cv<T, NT>(a.rtv + id * d, E.t, d, ln);
        store_cv<T, NT>(a.rgU + id * d, E.gU, d, ln);
    }
    if (__any(E.bad) && threadIdx.x == 0 && a.info) raise_info(a.info);
}
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_udl_boundary_kernel(FactArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x;
    int d = a.d;
    const long dd = long(d) * d;
    bool bad = false;
    LogAcc<T> la;
    la.init();
    Mat<T, NT> Sig;
    CV<T, NT> x;
    const bool rhs = a.eta != nullptr;                     // (the Cholesky factorisation has no right-hand side)
    load_sym_lower<T, NT>(Sig, a.rDv + (s * a.P) * dd, d, ln);
    MF_UNROLL for (int j = 0; j < NT; ++j) x.v[j] = T(0);
    if (rhs) load_cv<T, NT>(x, a.rtv + (s * a.P) * d, d, ln);
    store_mat<T, NT, true>(a.bSig + (s * a.P) * dd, Sig, d, lds, ln);
    if (rhs) store_cv<T, NT>(a.bx + (s * a.P) * d, x, d, ln);
    for (long c = 1; c < a.P; ++c) {
        const long id = s * a.P + c;
        Mat<T, NT> G, Dv, FT, LiT, Y;
        CV<T, NT> g, tv, u, t;
        load_sym_lower<T, NT>(G, a.rGU + id * dd, d, ln);
        load_sym_lower<T, NT>(Dv, a.rDv + id * dd, d, ln);
        load_mat_t<T, NT>(FT, a.rF + id * dd, d, ln);
        MF_UNROLL for (int j = 0; j < NT; ++j) g.v[j] = tv.v[j] = T(0);
        if (rhs) {
            load_cv<T, NT>(g, a.rgU + id * d, d, ln);
            load_cv<T, NT>(tv, a.rtv + id * d, d, ln);
        }
        // (the padded diagonal of both summands is one: take it once)
        MF_UNROLL for (int i = 0; i < NT; ++i)
            MF_UNROLL for (int j = i; j < NT; ++j)
                MF_UNROLL for (int e = 0; e < 4; ++e) {
                    const int row = 16 * i + Tr<T>::row(ln.q, e), col = 16 * j + ln.r;
                    Sig.t[i][j][e] += (row >= d && row == col) ? T(0) : G.t[i][j][e];
                }
        chol_inv_mat<T, NT>(Sig, LiT, lds, ln, la, bad);                              // (Delta_sep + GU)^-1 = LiT Li
        tn<T, NT, S_UPPER, S_FULL, S_FULL, OP_SET>(Y, LiT, FT);                       // Li F^T
        MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = i; j < NT; ++j) Sig.t[i][j] = Dv.t[i][j];
        tn<T, NT, S_FULL, S_FULL, S_UPPER, OP_SUB>(Sig, Y, Y);                        // Delta_e = Dv - F (.)^-1 F^T
        MF_UNROLL for (int j = 0; j < NT; ++j) x.v[j] += g.v[j];
        RV<T, NT> r;
        cv_to_rv<T, NT>(r, x, ln);
        tn_mv<T, NT, S_UPPER>(u, LiT, r);                                             // Li (x_sep + gU)
        cv_to_rv<T, NT>(r, u, ln);
        tn_mv<T, NT, S_FULL>(t, Y, r);                                                // F Li^T (.)
        MF_UNROLL for (int j = 0; j < NT; ++j) x.v[j] = tv.v[j] - t.v[j];
        store_mat<T, NT, true>(a.bSig + id * dd, Sig, d, lds, ln);
        if (rhs) store_cv<T, NT>(a.bx + id * d, x, d, ln);
    }
    if (__any(bad) && threadIdx.x == 0 && a.info) raise_info(a.info);
}

// block_diagonal_of_inverse (block_tri_diag.py:318-337), block Takahashi backwards: with G_k = W_k L_k^-1,
// Sigma_kk = L_k^-T L_k^-1 + G_k^T Sigma_{k+1,k+1} G_k and Sigma_{k+1,k} = -Sigma_{k+1,k+1} G_k.
// PART: the emit pass of the time-partitioned form (wave_inv_up_kernel / wave_inv_boundary_kernel): wavefront (s, c) walks the blocks
// [k_lo, k_hi] of its chunk (chunks count from the last block) from the Sigma the boundary pass left for the block above them.
template <typename T, int NT, bool PART = false>
__global__ void __launch_bounds__(64) wave_inverse_blocks_kernel(FactArgs<T> a) {
    using v4 = typename Tr<T>::v4;
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = PART ? blockIdx.x / a.P : blockIdx.x, c = PART ? blockIdx.x % a.P : 0, n = a.n;
    const long k_hi = PART ? n - 1 - c * a.L : n - 1;
    const long k_lo = PART ? (n - (c + 1) * a.L > 0 ? n - (c + 1) * a.L : 0) : 0;
    int d = a.d;
    const long dd = long(d) * d;
    const T* Lg = a.diag + s * n * dd;
    const T* Wg = a.sub ? a.sub + s * (n - 1) * dd : nullptr;
    bool bad = false;
    LogAcc<T> la;
    la.init();
    Mat<T, NT> Lk, WTk, Sig;
    v4 c10t = {0, 0, 0, 0};
    load_mat<T, NT, S_LOWER>(Lk, Lg + k_hi * dd, d, true, true, ln);
    if constexpr (NT == 2) load_tile_t<T>(c10t, Lg + k_hi * dd, d, 1, 0, ln);
    WTk.zero();
    Sig.zero();
    if constexpr (PART) {
        if (c > 0) {
            load_mat<T, NT, S_FULL>(Sig, a.bSig + (s * a.P + c) * dd, d, false, false, ln);
            load_mat_t<T, NT>(WTk, Wg + k_hi * dd, d, ln);
        }
    }
    for (long k = k_hi; k >= k_lo; --k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        const bool more = k > k_lo, coupled = k + 1 < n && Wg != nullptr;
        Mat<T, NT> Ln, WTn;
        v4 c10tn = {0, 0, 0, 0};
        if (more) {
            load_mat<T, NT, S_LOWER>(Ln, Lg + (k - 1) * dd, d, true, true, ln);
            if constexpr (NT == 2) load_tile_t<T>(c10tn, Lg + (k - 1) * dd, d, 1, 0, ln);
            if (Wg) load_mat_t<T, NT>(WTn, Wg + (k - 1) * dd, d, ln);
        }
        phase();
        Mat<T, NT> Li, Out;
        tri_inv_mat<T, NT>(Lk, c10t, Li, lds, ln, la, bad);
        tn<T, NT, S_LOWER, S_LOWER, S_FULL, OP_SET>(Out, Li, Li);                      // L^-T L^-1
        phase();
        if (coupled) {
            Mat<T, NT> G, NSG;
            tn<T, NT, S_FULL, S_LOWER, S_FULL, OP_SET>(G, WTk, Li);                    // G = W_k L_k^-1
            tn<T, NT, S_FULL, S_FULL, S_FULL, OP_NEG>(NSG, Sig, G);                    // -Sigma_{k+1,k+1} G   (Sigma symmetric)
            if (a.o2) store_mat<T, NT, false>(a.o2 + (s * (n - 1) + k) * dd, NSG, d, lds, ln);
            tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SUB>(Out, G, NSG);                    // + G^T Sigma G
        }
        store_mat<T, NT, false>(a.o1 + (s * n + k) * dd, Out, d, lds, ln);
        Sig = Out;
        if (more) {
            Lk = Ln;
            c10t = c10tn;
            if (Wg) WTk = WTn;
        }
    }
    (void)bad;
}
// the composed map of a chunk of the Takahashi recursion: Sigma_{k_lo} = M^T Sigma_in M + N with M <- M G_k, N <- G_k^T N G_k + L_k^-T L_k^-1
// block after block (rM holds M^T, rGU holds N in FactArgs' reduced-system arrays; the first chunk has no Sigma above it: G = 0 there)
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_inv_up_kernel(FactArgs<T> a) {
    using v4 = typename Tr<T>::v4;
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x / a.P, c = blockIdx.x % a.P, n = a.n;
    const long k_hi = n - 1 - c * a.L, k_lo = n - (c + 1) * a.L > 0 ? n - (c + 1) * a.L : 0;
    int d = a.d;
    const long dd = long(d) * d;
    const T* Lg = a.diag + s * n * dd;
    const T* Wg = a.sub + s * (n - 1) * dd;
    bool bad = false;
    LogAcc<T> la;
    la.init();
    Mat<T, NT> MT, N;
    identity_mat<T, NT>(MT, ln);
    N.zero();
    for (long k = k_hi; k >= k_lo; --k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        Mat<T, NT> Lk, WT, Li, G, X;
        v4 c10t = {0, 0, 0, 0};
        load_mat<T, NT, S_LOWER>(Lk, Lg + k * dd, d, true, true, ln);
        if constexpr (NT == 2) load_tile_t<T>(c10t, Lg + k * dd, d, 1, 0, ln);
        WT.zero();
        if (k + 1 < n) load_mat_t<T, NT>(WT, Wg + k * dd, d, ln);
        phase();
        tri_inv_mat<T, NT>(Lk, c10t, Li, lds, ln, la, bad);
        tn<T, NT, S_FULL, S_LOWER, S_FULL, OP_SET>(G, WT, Li);                         // G = W_k L_k^-1
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, G, MT);                           // (M G)^T = G^T M^T
        MT = X;
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, N, G);                            // N G
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(N, G, X);                            // G^T N G
        tn<T, NT, S_LOWER, S_LOWER, S_FULL, OP_ADD>(N, Li, Li);                        // + L^-T L^-1
    }
    const long id = s * a.P + c;
    store_mat<T, NT, false>(a.rDv + id * dd, MT, d, lds, ln);
    store_mat<T, NT, false>(a.rGU + id * dd, N, d, lds, ln);
    (void)bad;
}
// Sigma above every chunk: Sigma_in(c + 1) = M_c^T Sigma_in(c) M_c + N_c  (Sigma_in(0) = 0: nothing above the last block)
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_inv_boundary_kernel(FactArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x;
    int d = a.d;
    const long dd = long(d) * d;
    Mat<T, NT> Sig;
    Sig.zero();
    for (long c = 0; c + 1 < a.P; ++c) {
        const long id = s * a.P + c;
        Mat<T, NT> M, N, X;
        load_mat_t<T, NT>(M, a.rDv + id * dd, d, ln);                                  // stored M^T -> M
        load_mat<T, NT, S_FULL>(N, a.rGU + id * dd, d, false, false, ln);
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, Sig, M);                          // Sigma M
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(Sig, M, X);                          // M^T Sigma M
        MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j) Sig.t[i][j] += N.t[i][j];
        store_mat<T, NT, false>(a.bSig + (id + 1) * dd, Sig, d, lds, ln);
    }
}

// m = g^T with g lower triangular (its strict upper triangle, whatever it holds, reads as zero): the transposed Cholesky factor
template <typename T, int NT> MF_DEV void load_lower_t(Mat<T, NT>& m, const T* __restrict__ g, int d, const Lane& ln) {
    MF_UNROLL for (int ti = 0; ti < NT; ++ti)
        MF_UNROLL for (int tj = 0; tj < NT; ++tj) {
            if (ti > tj) { m.t[ti][tj] = typename Tr<T>::v4{0, 0, 0, 0}; continue; }
            MF_UNROLL for (int e = 0; e < 4; ++e) {
                const int i = 16 * ti + Tr<T>::row(ln.q, e), j = 16 * tj + ln.r;
                const bool in = i < d && j < d && i <= j;
                const T v = g[in ? j * d + i : 0];
                m.t[ti][tj][e] = in ? v : T(0);
            }
        }
}

template <typename T> struct MargArgs {
    long B, n;               // series, time points
    int d;
    const T *mu0, *cholP0, *A, *b, *cholQ;
    T *omean, *ocov, *osub;  // [B, n, d] | NULL, [B, n, d, d], [B, n - 1, d, d] | NULL
    // partitioned in time (wave_marg_up_kernel / wave_marg_boundary_kernel / wave_marginals_kernel<.., PART>): P chunks of L transitions;
    // per (series, chunk) the composed map (M, N, v): S -> M S M^T + N, m -> M m + v, and the state the chunk starts from
    long P, L;
    T *wM, *wN, *wv, *bP, *bm;   // [B, P, d, d] x 2, [B, P, d]; [B, P, d, d], [B, P, d]
};

// marginal means / covariances / subsequent covariances (state_space_model.py:232-262,326-341; gauss_markov.py:107-117) by the forward
// recursion mu' = A mu + b, P' = A P A^T + C C^T, Cov(x', x) = A P - one wavefront per series; with A^T and C^T read transposed
// every product is P^T Q: P A^T = tn(P, A^T) (P symmetric), A (P A^T) = tn(A^T, .), C C^T = tn(C^T, C^T), A P = tn(A^T, P).
// PART: wavefront (s, c) walks the transitions [c L, (c + 1) L) from the state the boundary pass left for its chunk.
template <typename T, int NT, bool PART = false>
__global__ void __launch_bounds__(64) wave_marginals_kernel(MargArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = PART ? blockIdx.x / a.P : blockIdx.x, c = PART ? blockIdx.x % a.P : 0, n = a.n, nt = n - 1;
    const long t_lo = PART ? c * a.L : 0, t_hi = PART ? ((c + 1) * a.L < nt ? (c + 1) * a.L : nt) : nt;
    int d = a.d;
    const long dd = long(d) * d;
    const T* Ag = a.A + s * nt * dd;
    const T* Cg = a.cholQ + s * nt * dd;
    const T* bg = a.b ? a.b + s * nt * d : nullptr;
    T* om = a.omean ? a.omean + s * n * d : nullptr;
    T* oc = a.ocov + s * n * dd;
    T* os = a.osub ? a.osub + s * nt * dd : nullptr;
    Mat<T, NT> P, AT, CT;
    CV<T, NT> m, bk;
    MF_UNROLL for (int j = 0; j < NT; ++j) m.v[j] = bk.v[j] = T(0);
    if (!PART || c == 0) {
        load_lower_t<T, NT>(CT, a.cholP0 + s * dd, d, ln);
        tn<T, NT, S_UPPER, S_UPPER, S_FULL, OP_SET>(P, CT, CT);
        if (om) {
            load_cv<T, NT>(m, a.mu0 + s * d, d, ln);
            store_cv<T, NT>(om, m, d, ln);
        }
        store_mat<T, NT, false>(oc, P, d, lds, ln);
    } else {
        load_mat<T, NT, S_FULL>(P, a.bP + (s * a.P + c) * dd, d, false, false, ln);
        if (om) load_cv<T, NT>(m, a.bm + (s * a.P + c) * d, d, ln);
    }
    if (t_hi > t_lo) {
        load_mat_t<T, NT>(AT, Ag + t_lo * dd, d, ln);
        load_lower_t<T, NT>(CT, Cg + t_lo * dd, d, ln);
        if (om && bg) load_cv<T, NT>(bk, bg + t_lo * d, d, ln);
    }
    for (long k = t_lo; k < t_hi; ++k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        Mat<T, NT> ATn, CTn;
        CV<T, NT> bn;
        MF_UNROLL for (int j = 0; j < NT; ++j) bn.v[j] = T(0);
        const long kn = k + 1 < t_hi ? k + 1 : k;
        load_mat_t<T, NT>(ATn, Ag + kn * dd, d, ln);
        load_lower_t<T, NT>(CTn, Cg + kn * dd, d, ln);
        if (om && bg) load_cv<T, NT>(bn, bg + kn * d, d, ln);
        phase();
        if (os) {
            Mat<T, NT> AP;
            tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(AP, AT, P);                      // A P
            store_mat<T, NT, false>(os + k * dd, AP, d, lds, ln);
        }
        Mat<T, NT> X;
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, P, AT);                           // P A^T
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(P, AT, X);                           // A P A^T
        tn<T, NT, S_UPPER, S_UPPER, S_FULL, OP_ADD>(P, CT, CT);                        // + C C^T
        if (om) {
            RV<T, NT> mr;
            CV<T, NT> am;
            cv_to_rv<T, NT>(mr, m, ln);
            tn_mv<T, NT, S_FULL>(am, AT, mr);                                          // A m
            MF_UNROLL for (int j = 0; j < NT; ++j) m.v[j] = am.v[j] + bk.v[j];
            store_cv<T, NT>(om + (k + 1) * d, m, d, ln);
        }
        store_mat<T, NT, false>(oc + (k + 1) * dd, P, d, lds, ln);
        AT = ATn;
        CT = CTn;
        MF_UNROLL for (int j = 0; j < NT; ++j) bk.v[j] = bn.v[j];
    }
}
// the composed map of a chunk's transitions: M <- A M, N <- A N A^T + C C^T, v <- A v + b (the same three products per step as the walk)
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_marg_up_kernel(MargArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x / a.P, c = blockIdx.x % a.P, nt = a.n - 1;
    const long t_lo = c * a.L, t_hi = (c + 1) * a.L < nt ? (c + 1) * a.L : nt;
    int d = a.d;
    const long dd = long(d) * d;
    const T* Ag = a.A + s * nt * dd;
    const T* Cg = a.cholQ + s * nt * dd;
    const bool means = a.omean && a.b;
    const T* bg = means ? a.b + s * nt * d : nullptr;
    Mat<T, NT> M, N;
    CV<T, NT> v;
    identity_mat<T, NT>(M, ln);
    N.zero();
    MF_UNROLL for (int j = 0; j < NT; ++j) v.v[j] = T(0);
    for (long k = t_lo; k < t_hi; ++k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        Mat<T, NT> AT, CT, X;
        CV<T, NT> bk;
        MF_UNROLL for (int j = 0; j < NT; ++j) bk.v[j] = T(0);
        load_mat_t<T, NT>(AT, Ag + k * dd, d, ln);
        load_lower_t<T, NT>(CT, Cg + k * dd, d, ln);
        if (means) load_cv<T, NT>(bk, bg + k * d, d, ln);
        phase();
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, AT, M);                           // A M
        M = X;
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, N, AT);                           // N A^T
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(N, AT, X);                           // A N A^T
        tn<T, NT, S_UPPER, S_UPPER, S_FULL, OP_ADD>(N, CT, CT);                        // + C C^T
        RV<T, NT> vr;
        CV<T, NT> av;
        cv_to_rv<T, NT>(vr, v, ln);
        tn_mv<T, NT, S_FULL>(av, AT, vr);
        MF_UNROLL for (int j = 0; j < NT; ++j) v.v[j] = av.v[j] + bk.v[j];
    }
    const long id = s * a.P + c;
    // (the padded diagonal of M is one: only the d x d corner is stored, and read back with zero padding - the padding of the
    // covariances it multiplies is zero as well)
    store_mat<T, NT, false>(a.wM + id * dd, M, d, lds, ln);
    store_mat<T, NT, false>(a.wN + id * dd, N, d, lds, ln);
    store_cv<T, NT>(a.wv + id * d, v, d, ln);
}
// the state every chunk starts from: S_{c+1} = M_c S_c M_c^T + N_c, m_{c+1} = M_c m_c + v_c
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_marg_boundary_kernel(MargArgs<T> a) {
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x;
    int d = a.d;
    const long dd = long(d) * d;
    Mat<T, NT> P, CT;
    CV<T, NT> m;
    load_lower_t<T, NT>(CT, a.cholP0 + s * dd, d, ln);
    tn<T, NT, S_UPPER, S_UPPER, S_FULL, OP_SET>(P, CT, CT);
    MF_UNROLL for (int j = 0; j < NT; ++j) m.v[j] = T(0);
    if (a.omean) load_cv<T, NT>(m, a.mu0 + s * d, d, ln);
    for (long c = 0; c + 1 < a.P; ++c) {
        const long id = s * a.P + c;
        Mat<T, NT> MT, N, X;
        CV<T, NT> v, am;
        load_mat_t<T, NT>(MT, a.wM + id * dd, d, ln);
        load_mat<T, NT, S_FULL>(N, a.wN + id * dd, d, false, false, ln);
        load_cv<T, NT>(v, a.wv + id * d, d, ln);
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, P, MT);                           // S M^T
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(P, MT, X);                           // M S M^T
        MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j) P.t[i][j] += N.t[i][j];
        RV<T, NT> mr;
        cv_to_rv<T, NT>(mr, m, ln);
        tn_mv<T, NT, S_FULL>(am, MT, mr);
        MF_UNROLL for (int j = 0; j < NT; ++j) m.v[j] = am.v[j] + v.v[j];
        store_mat<T, NT, false>(a.bP + (id + 1) * dd, P, d, lds, ln);
        store_cv<T, NT>(a.bm + (id + 1) * d, m, d, ln);
    }
}

// ---- StateSpaceModel.kl_divergence at 16 <= d <= 32 in ONE walk (state_space_model.py:528-593) ---------------------------------
// KL(q1 || q2) = 1/2 sum_k [ tr(D_k Sigma_k) + 2 tr(S_k Cov(x_{k+1}, x_k)) + delta_k^T D_k delta_k + 2 delta_k^T S_k^T delta_{k+1}
//                            + 2 log|C2_k| - 2 log|C1_k| - d ],   D_k = Q2_k^-1 + A2_{k+1}^T Q2_{k+1}^-1 A2_{k+1},  S_k = -Q2_{k+1}^-1 A2_{k+1}
// (the block rows of q2's precision against q1's moments; wave_ssm_kl_terms_kernel, mf_wave.hpp, evaluates a block from moments in
// memory).  Here the moments never exist in memory: wavefront (series, chunk) walks q1's moment recursion (wave_marginals_kernel:
// P' = A1 P A1^T + C1 C1^T, Cov(x', x) = A1 P, m' = A1 m + b1) from the state wave_marg_boundary_kernel left for its chunk, q2's
// means alongside (from wave_means_boundary_kernel), and reduces every block against q2's block row on the spot; the inverse of
// q2's next factor is carried to the next block (one triangular inversion per block).  Per block terms go to `terms` [B, n] and
// are summed in a fixed order by row_sums_kernel.  Round 6, first form: moments 1.37 ms (2 GB written) + terms 1.34 ms (2 GB read back)
// at B = 512, T = 1000, d = 16.
template <typename T> struct KlWalkArgs {
    long B, n;
    int d;
    const T *mu0_1, *cp0_1, *A_1, *b_1, *cq_1;
    const T *mu0_2, *cp0_2, *A_2, *b_2, *cq_2;
    long P, L;                       // chunks of L transitions (P = 1: the whole chain)
    const T *bP, *bm, *m2_in;        // [B, P, d, d], [B, P, d], [B, P, d]: q1's covariance / mean and q2's mean at position c L
    T* terms;                        // [B, n]
};
template <typename T, int NT> MF_DEV void sym_complete(Mat<T, NT>& m, T* lds, const Lane& ln) {
    if constexpr (NT == 2) transpose_tile<T>(m.t[1][0], m.t[0][1], lds, ln);
}
template <typename T, int NT> MF_DEV T frob(const Mat<T, NT>& x, const Mat<T, NT>& y, T acc) {
    MF_UNROLL for (int i = 0; i < NT; ++i) MF_UNROLL for (int j = 0; j < NT; ++j)
        MF_UNROLL for (int e = 0; e < 4; ++e) acc = __builtin_fma(x.t[i][j][e], y.t[i][j][e], acc);
    return acc;
}
// sum_j log|C[j][j]| over the d x d lower factor at g (per 16-lane row: the caller takes sum16)
template <typename T, int NT> MF_DEV T log_diag(const T* __restrict__ g, int d, const Lane& ln) {
    T l = T(0);
    MF_UNROLL for (int i = 0; i < NT; ++i) {
        const int j = 16 * i + ln.r;
        const T v = g[j < d ? j * d + j : 0];
        l += (j < d) ? log(v < T(0) ? -v : v) : T(0);
    }
    return l;
}
template <typename T, int NT>
__global__ void __launch_bounds__(64) wave_kl_walk_kernel(KlWalkArgs<T> a) {
    constexpr bool EARLY = !(sizeof(T) == 8 && NT == 2);
    using v4 = typename Tr<T>::v4;
    constexpr int TS = 16 * Tr<T>::LD;
    __shared__ __attribute__((aligned(16))) T lds[NT * NT * TS];
    Lane ln{(int)(threadIdx.x & 15), (int)(threadIdx.x >> 4)};
    const long s = blockIdx.x / a.P, c = blockIdx.x % a.P, n = a.n, nt = n - 1;
    const long t_lo = c * a.L, t_hi = (c + 1) * a.L < nt ? (c + 1) * a.L : nt;
    int d = a.d;
    const long dd = long(d) * d;
    LogAcc<T> la;
    bool bad = false;
    // q1 at position t_lo, q2's mean there
    Mat<T, NT> P;
    CV<T, NT> m1, m2;
    if (c == 0) {
        Mat<T, NT> CT;
        load_lower_t<T, NT>(CT, a.cp0_1 + s * dd, d, ln);
        tn<T, NT, S_UPPER, S_UPPER, S_FULL, OP_SET>(P, CT, CT);
        load_cv<T, NT>(m1, a.mu0_1 + s * d, d, ln);
        load_cv<T, NT>(m2, a.mu0_2 + s * d, d, ln);
    } else {
        load_mat<T, NT, S_FULL>(P, a.bP + (s * a.P + c) * dd, d, false, false, ln);
        load_cv<T, NT>(m1, a.bm + (s * a.P + c) * d, d, ln);
        load_cv<T, NT>(m2, a.m2_in + (s * a.P + c) * d, d, ln);
    }
    // Q2^-1 of block t_lo and log|C2| of it
    Mat<T, NT> Q0;
    T logdet0;
    {
        Mat<T, NT> C0, Ci;
        v4 c10t = {0, 0, 0, 0};
        const T* c0p = t_lo == 0 ? a.cp0_2 + s * dd : a.cq_2 + (s * nt + t_lo - 1) * dd;
        load_mat<T, NT, S_LOWER>(C0, c0p, d, true, true, ln);
        if constexpr (NT == 2) load_tile_t<T>(c10t, c0p, d, 1, 0, ln);
        la.init();
        tri_inv_mat<T, NT>(C0, c10t, Ci, lds, ln, la, bad);
        logdet0 = tri_logdet<T, NT>(la);
        tn<T, NT, S_LOWER, S_LOWER, S_UPPER, OP_SET>(Q0, Ci, Ci);
        sym_complete<T, NT>(Q0, lds, ln);
    }
    // one block's terms from (D, S | none), q1's covariance and cross covariance, the mean differences
    auto emit = [&](long k, Mat<T, NT>& Dn, const Mat<T, NT>* S, const Mat<T, NT>* cross, const CV<T, NT>& dk, const CV<T, NT>* dnext, T l1) {
        T acc = frob<T, NT>(Dn, P, T(0));
        if (S) acc = __builtin_fma(T(2), frob<T, NT>(*S, *cross, T(0)), acc);                 // + 2 sum(S o Cov(x_{k+1}, x_k))
        T total = xor_rows<T>(sum16<T>(acc));
        RV<T, NT> dr;
        CV<T, NT> y;
        cv_to_rv<T, NT>(dr, dk, ln);
        tn_mv<T, NT, S_FULL>(y, Dn, dr);                                                      // D delta_k (D symmetric)
        T mh = dot_cv<T, NT>(y, dk);
        if (S) {
            cv_to_rv<T, NT>(dr, *dnext, ln);
            tn_mv<T, NT, S_FULL>(y, *S, dr);                                                  // S^T delta_{k+1}
            mh = __builtin_fma(T(2), dot_cv<T, NT>(y, dk), mh);
        }
        total += sum16<T>(mh);
        total += T(2) * logdet0 - T(2) * sum16<T>(l1) - T(d);
        if (threadIdx.x == 0) a.terms[s * n + k] = total;
    };
    for (long k = t_lo; k < t_hi; ++k) {
        asm volatile("" : "+v"(ln.r), "+v"(ln.q));
        asm volatile("" : "+s"(d));
        // EARLY (every instantiation but f64 at NT = 2, which has no registers for it): everything the block needs from memory is
        // asked for here, one exposed round trip per block instead of five; the wavefronts sharing the SIMD cover that one
        Mat<T, NT> C1, Am, A2T, A1T, CT1;
        v4 c10t = {0, 0, 0, 0};
        CV<T, NT> b1v, b2v;
        const T* c1 = a.cq_2 + (s * nt + k) * dd;
        const T* c1p = k == 0 ? a.cp0_1 + s * dd : a.cq_1 + (s * nt + k - 1) * dd;
        T l1 = T(0);
        load_mat<T, NT, S_LOWER>(C1, c1, d, true, true, ln);
        if constexpr (NT == 2) load_tile_t<T>(c10t, c1, d, 1, 0, ln);
        load_mat<T, NT, S_FULL>(Am, a.A_2 + (s * nt + k) * dd, d, false, false, ln);
        if constexpr (EARLY) {
            load_mat_t<T, NT>(A2T, a.A_2 + (s * nt + k) * dd, d, ln);
            load_mat_t<T, NT>(A1T, a.A_1 + (s * nt + k) * dd, d, ln);
            load_lower_t<T, NT>(CT1, a.cq_1 + (s * nt + k) * dd, d, ln);
            load_cv<T, NT>(b1v, a.b_1 + (s * nt + k) * d, d, ln);
            load_cv<T, NT>(b2v, a.b_2 + (s * nt + k) * d, d, ln);
            l1 = log_diag<T, NT>(c1p, d, ln);
        }
        // ---- q2's block row k: D = Q_k^-1 + A^T Q_{k+1}^-1 A, S = -Q_{k+1}^-1 A ------------------------------------------------
        Mat<T, NT> Q1, S, Dn;
        T logdet1;
        {
            Mat<T, NT> Ci;
            la.init();
            tri_inv_mat<T, NT>(C1, c10t, Ci, lds, ln, la, bad);
            logdet1 = tri_logdet<T, NT>(la);
            tn<T, NT, S_LOWER, S_LOWER, S_UPPER, OP_SET>(Q1, Ci, Ci);
            sym_complete<T, NT>(Q1, lds, ln);
        }
        tn<T, NT, S_FULL, S_FULL, S_FULL, OP_NEG>(S, Q1, Am);
        Dn = Q0;
        tn<T, NT, S_FULL, S_FULL, S_UPPER, OP_SUB>(Dn, Am, S);
        sym_complete<T, NT>(Dn, lds, ln);
        if constexpr (!EARLY) phase();
        // ---- the means of both chains at k + 1 --------------------------------------------------------------------------------
        CV<T, NT> m1n, m2n, dk, dn;
        {
            RV<T, NT> r;
            CV<T, NT> y;
            if constexpr (!EARLY) {
                load_mat_t<T, NT>(A2T, a.A_2 + (s * nt + k) * dd, d, ln);                 // (the lines Am came from)
                load_cv<T, NT>(b2v, a.b_2 + (s * nt + k) * d, d, ln);
            }
            cv_to_rv<T, NT>(r, m2, ln);
            tn_mv<T, NT, S_FULL>(y, A2T, r);
            MF_UNROLL for (int j = 0; j < NT; ++j) m2n.v[j] = y.v[j] + b2v.v[j];
            if constexpr (!EARLY) {
                load_mat_t<T, NT>(A1T, a.A_1 + (s * nt + k) * dd, d, ln);
                load_cv<T, NT>(b1v, a.b_1 + (s * nt + k) * d, d, ln);
            }
            cv_to_rv<T, NT>(r, m1, ln);
            tn_mv<T, NT, S_FULL>(y, A1T, r);
            MF_UNROLL for (int j = 0; j < NT; ++j) m1n.v[j] = y.v[j] + b1v.v[j];
            MF_UNROLL for (int j = 0; j < NT; ++j) { dk.v[j] = m2.v[j] - m1.v[j]; dn.v[j] = m2n.v[j] - m1n.v[j]; }
        }
        // ---- q1's cross covariance, the block's terms, then q1's next covariance ---------------------------------------------------
        {
            Mat<T, NT> AP;
            tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(AP, A1T, P);                         // Cov(x_{k+1}, x_k) = A1 P
            if constexpr (!EARLY) l1 = log_diag<T, NT>(c1p, d, ln);
            emit(k, Dn, &S, &AP, dk, &dn, l1);
        }
        if constexpr (!EARLY) phase();
        {
            Mat<T, NT> X;
            tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(X, P, A1T);                          // P A1^T
            tn<T, NT, S_FULL, S_FULL, S_FULL, OP_SET>(P, A1T, X);                          // A1 P A1^T
            if constexpr (!EARLY) load_lower_t<T, NT>(CT1, a.cq_1 + (s * nt + k) * dd, d, ln);
            tn<T, NT, S_UPPER, S_UPPER, S_FULL, OP_ADD>(P, CT1, CT1);                      // + C1 C1^T
        }
        Q0 = Q1;
        logdet0 = logdet1;
        MF_UNROLL for (int j = 0; j < NT; ++j) { m1.v[j] = m1n.v[j]; m2.v[j] = m2n.v[j]; }
        if constexpr (!EARLY) phase();
    }
    if (t_hi == nt) {                                                                     // the last block: no transition out of it
        CV<T, NT> dk;
        MF_UNROLL for (int j = 0; j < NT; ++j) dk.v[j] = m2.v[j] - m1.v[j];
        const T* c1p = nt == 0 ? a.cp0_1 + s * dd : a.cq_1 + (s * nt + nt - 1) * dd;
        emit(nt, Q0, nullptr, nullptr, dk, nullptr, log_diag<T, NT>(c1p, d, ln));
    }
    (void)bad;
}

}  // namespace wv
}  // namespace mf
