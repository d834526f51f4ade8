// Operator entry points (cholesky, solve, dense_mult, log-det, block_diagonal_of_inverse, upper_diagonal_lower + posterior
// chain, precision assembly, marginal means, block products) for LARGE state dimension, built on the LDS-tile / MFMA
// engine of mf_big_impl.hpp: one workgroup per series (or per block where the op is parallel in time), natural order.
// Included once per scalar type, like mf_big_impl.hpp.  These are the API-parity forms (what the reference's
// block_tri_diag.py / state_space_model.py methods compute) for d > 9; the fused log-likelihood does not use them.
namespace mf {
namespace MF_BIG_NS {

// tile -> global, lower triangle kept, upper written as zero
template <int DP> __device__ __forceinline__ void store_tile_lower(real* __restrict__ g, const real* __restrict__ tile, int d) {
    constexpr int LD = Geo<DP>::LD;
    for (int e = threadIdx.x; e < d * d; e += NTHR) {
        const int row = e / d, col = e % d;
        g[e] = col <= row ? tile[row * LD + col] : real(0);
    }
}
template <int DP> __device__ __forceinline__ void copy_tile(real* __restrict__ dst, const real* __restrict__ src) {
    for (int e = threadIdx.x; e < Geo<DP>::TILE; e += NTHR) dst[e] = src[e];
}

// SymmetricBlockTriDiagonal.cholesky (block_tri_diag.py:423-436), natural order, workgroup per series
template <int DP>
__global__ void __launch_bounds__(NTHR) bigop_cholesky_kernel(long B, long n, int d, const real* __restrict__ diag,
                                                             const real* __restrict__ sub, real* __restrict__ ldiag,
                                                             real* __restrict__ lsub, int* info) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long s = blockIdx.x;
    real *S = sm.tile(0), *Linv = sm.tile(1), *W = sm.tile(2), *Fin = sm.tile(3);
    bool bad = false;
    const long dd = (long)d * d;
    for (long k = 0; k < n; ++k) {
        load_tile<DP>(S, diag + (s * n + k) * dd, nullptr, d, false, true);
        if (sub && k > 0) {
            load_tile<DP>(Fin, sub + (s * (n - 1) + k - 1) * dd, nullptr, d, false, false);
            __syncthreads();
            gemm<DP, 0, 1, 0, K_B_UPPER, O_FULL>(Fin, Linv, W, 1.f);                 // W = S_{k-1} L_{k-1}^-T
            __syncthreads();
            store_tile<DP>(lsub + (s * (n - 1) + k - 1) * dd, W, d);
            gemm<DP, 0, 1, 1, K_FULL, O_FULL>(W, W, S, -1.f);                        // D_k - W W^T
        }
        __syncthreads();
        (void)factor_invert<DP, true>(S, Linv, bad, sm.scratch());
        store_tile_lower<DP>(ldiag + (s * n + k) * dd, S, d);
        __syncthreads();
    }
    if (threadIdx.x == 0 && bad && info) raise_info(info);
}

// LowerTriangularBlockTriDiagonal.solve (block_tri_diag.py:339-351), workgroup per right-hand-side series
template <int DP>
__global__ void __launch_bounds__(NTHR) bigop_solve_kernel(long Bl, long Br, long n, int d, const real* __restrict__ ldiag,
                                                          const real* __restrict__ lsub, const real* __restrict__ rhs,
                                                          real* __restrict__ out, int transpose) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long r = blockIdx.x, s = r % Bl;
    real *L = sm.tile(0), *Linv = sm.tile(1), *W = sm.tile(2);
    real *z = sm.vec(0), *x = sm.vec(1);
    bool bad = false;
    const long dd = (long)d * d;
    if (threadIdx.x < 64) z[threadIdx.x] = 0;
    for (long pp = 0; pp < n; ++pp) {
        const long k = transpose ? n - 1 - pp : pp;
        load_tile<DP>(L, ldiag + (s * n + k) * dd, nullptr, d, true, true);
        load_vec_lds<DP>(x, rhs + (r * n + k) * d, nullptr, d);
        const bool coupled = lsub && pp > 0;
        if (coupled) load_tile<DP>(W, lsub + (s * (n - 1) + (transpose ? k : k - 1)) * dd, nullptr, d, false, false);
        __syncthreads();
        if (coupled) {
            if (!transpose) matvec<DP, 0>(W, z, x, -1.f, 1.f, sm.scratch());          // x -= W z
            else matvec<DP, 1>(W, z, x, -1.f, 1.f, sm.scratch());                     // x -= W^T z
        }
        (void)factor_invert<DP, false>(L, Linv, bad, sm.scratch());
        if (!transpose) matvec<DP, 0>(Linv, x, z, 1.f, 0.f, sm.scratch());
        else matvec<DP, 1>(Linv, x, z, 1.f, 0.f, sm.scratch());
        if (threadIdx.x < d) out[(r * n + k) * d + threadIdx.x] = z[threadIdx.x];
        __syncthreads();
    }
}

// BlockTriDiagonal.dense_mult (block_tri_diag.py:175-199): one wavefront per (series, block), thread = output row.  Every d x d block
// goes through LDS first - read from memory in element order, i.e. coalesced (a thread that walks its own row of a row-major block
// reads at a stride of d elements: 1.5 TB/s at d = 16, B = 512, T = 1000) - and is then read by rows or by columns from an image of
// row stride d + 1 (conflict-free either way).
__global__ void __launch_bounds__(64) bigop_matvec_kernel(long Bl, long Br, long n, int d, const real* __restrict__ diag,
                                                         const real* __restrict__ sub, const real* __restrict__ x,
                                                         real* __restrict__ out, int mode) {
    constexpr int DMAX = sizeof(real) == 8 ? 32 : 64;          // the largest state dimension of this scalar type
    __shared__ real tile[DMAX * (DMAX + 1)];
    __shared__ real xs[DMAX];
    const long id = blockIdx.x, r = id / n, k = id % n, s = r % Bl;
    const int i = threadIdx.x, ld = d + 1;
    const bool row = i < d;
    const long dd = (long)d * d;
    auto stage = [&](const real* __restrict__ g, const real* __restrict__ v) {
        __syncthreads();
        for (int e = threadIdx.x; e < d * d; e += 64) tile[(e / d) * ld + (e % d)] = g[e];
        if (row) xs[i] = v[i];
        __syncthreads();
    };
    real a = 0;
    stage(diag + (s * n + k) * dd, x + (r * n + k) * d);
    if (row) {
        for (int j = 0; j < d; ++j) {
            real e;
            if (mode == 0) e = (j <= i) ? tile[i * ld + j] : real(0);
            else if (mode == 1) e = (j >= i) ? tile[j * ld + i] : real(0);
            else e = (j <= i) ? tile[i * ld + j] : tile[j * ld + i];
            a += e * xs[j];
        }
    }
    if (sub) {
        if ((mode == 0 || mode == 2) && k > 0) {
            stage(sub + (s * (n - 1) + k - 1) * dd, x + (r * n + k - 1) * d);
            if (row) for (int j = 0; j < d; ++j) a += tile[i * ld + j] * xs[j];
        }
        if ((mode == 1 || mode == 2) && k + 1 < n) {
            stage(sub + (s * (n - 1) + k) * dd, x + (r * n + k + 1) * d);
            if (row) for (int j = 0; j < d; ++j) a += tile[j * ld + i] * xs[j];
        }
    }
    if (row) out[(r * n + k) * d + i] = a;
}

// LowerTriangularBlockTriDiagonal.abs_log_det (block_tri_diag.py:353-366): one wavefront per series
__global__ void __launch_bounds__(64) bigop_logdet_kernel(long B, long n, int d, const real* __restrict__ ldiag,
                                                         real* __restrict__ out) {
    const long s = blockIdx.x;
    real acc = 0;
    for (long e = threadIdx.x; e < n * d; e += 64) {
        const long k = e / d;
        const int i = (int)(e % d);
        const real v = ldiag[(s * n + k) * (long)d * d + (long)i * d + i];
        acc += real(0.5) * mf_log(v * v);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (threadIdx.x == 0) out[s] = acc;
}

// block_diagonal_of_inverse (block_tri_diag.py:318-337): block Takahashi, backward, workgroup per series
template <int DP>
__global__ void __launch_bounds__(NTHR) bigop_diag_of_inverse_kernel(long B, long n, int d, const real* __restrict__ ldiag,
                                                                    const real* __restrict__ lsub, real* __restrict__ odiag,
                                                                    real* __restrict__ osub) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long s = blockIdx.x;
    real *L = sm.tile(0), *Linv = sm.tile(1), *W = sm.tile(2), *G = sm.tile(3), *Sig = sm.tile(4), *SG = sm.tile(5),
         *Out = sm.tile(6);
    bool bad = false;
    const long dd = (long)d * d;
    for (long k = n - 1; k >= 0; --k) {
        load_tile<DP>(L, ldiag + (s * n + k) * dd, nullptr, d, true, true);
        const bool coupled = lsub && k + 1 < n;
        if (coupled) load_tile<DP>(W, lsub + (s * (n - 1) + k) * dd, nullptr, d, false, false);
        __syncthreads();
        (void)factor_invert<DP, false>(L, Linv, bad, sm.scratch());
        gemm<DP, 1, 0, 0, K_A_UPPER, O_FULL>(Linv, Linv, Out, 1.f);                   // L^-T L^-1
        if (coupled) {
            gemm<DP, 0, 0, 0, K_B_LOWER, O_FULL>(W, Linv, G, 1.f);                    // G = W L^-1
            __syncthreads();
            gemm<DP, 0, 0, 0, K_FULL, O_FULL>(Sig, G, SG, 1.f);                       // Sigma_{k+1} G
            __syncthreads();
            if (osub) {
                constexpr int LD = Geo<DP>::LD;
                for (int e = threadIdx.x; e < d * d; e += NTHR) osub[(s * (n - 1) + k) * dd + e] = -SG[(e / d) * LD + (e % d)];
            }
            gemm<DP, 1, 0, 1, K_FULL, O_FULL>(G, SG, Out, 1.f);                       // + G^T Sigma_{k+1} G
        }
        __syncthreads();
        copy_tile<DP>(Sig, Out);
        store_tile<DP>(odiag + (s * n + k) * dd, Out, d);
        __syncthreads();
    }
}

// upper_diagonal_lower (block_tri_diag.py:438-545) + posterior chain (kalman_filter.py:159-174), backward, workgroup per series
template <int DP>
__global__ void __launch_bounds__(NTHR) bigop_udl_kernel(long B, long n, int d, const real* __restrict__ diag,
                                                        const real* __restrict__ sub, real* __restrict__ ut,
                                                        real* __restrict__ chol_d, const real* __restrict__ eta,
                                                        real* __restrict__ m_post, real* __restrict__ chol_dinv, int* info) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long s = blockIdx.x;
    real *Dl = sm.tile(0), *Linv = sm.tile(1), *S = sm.tile(2), *U = sm.tile(3), *Ut = sm.tile(4), *Q = sm.tile(5),
         *Qi = sm.tile(6);
    real *x = sm.vec(0), *xp = sm.vec(1), *tmp = sm.vec(2), *mk = sm.vec(3);
    bool bad = false;
    const long dd = (long)d * d;
    if (threadIdx.x < 64) xp[threadIdx.x] = 0;
    for (long k = n - 1; k >= 0; --k) {
        load_tile<DP>(Dl, diag + (s * n + k) * dd, nullptr, d, false, true);
        if (eta) load_vec_lds<DP>(x, eta + (s * n + k) * d, nullptr, d);
        const bool coupled = k + 1 < n;
        if (coupled) load_tile<DP>(S, sub + (s * (n - 1) + k) * dd, nullptr, d, false, false);
        __syncthreads();
        if (coupled) {
            gemm<DP, 0, 0, 0, K_A_LOWER, O_FULL>(Linv, S, U, 1.f);                    // L^-1 S  (L = chol Delta_{k+1})
            __syncthreads();
            gemm<DP, 1, 0, 1, K_FULL, O_FULL>(U, U, Dl, -1.f);                        // Delta_k = D_k - S^T Delta_{k+1}^-1 S
            gemm<DP, 1, 0, 0, K_A_UPPER, O_FULL>(Linv, U, Ut, 1.f);                   // U_k^T = Delta_{k+1}^-1 S
            __syncthreads();
            store_tile<DP>(ut + (s * (n - 1) + k) * dd, Ut, d);
            if (eta) matvec<DP, 1>(Ut, xp, x, -1.f, 1.f, sm.scratch());               // x_k = eta_k - U_k x_{k+1}
        }
        (void)factor_invert<DP, true>(Dl, Linv, bad, sm.scratch());
        store_tile_lower<DP>(chol_d + (s * n + k) * dd, Dl, d);
        if (eta) {
            if (threadIdx.x < 64) xp[threadIdx.x] = x[threadIdx.x];
            __syncthreads();
            matvec<DP, 0>(Linv, x, tmp, 1.f, 0.f, sm.scratch());
            matvec<DP, 1>(Linv, tmp, mk, 1.f, 0.f, sm.scratch());                     // m_k = Delta_k^-1 x_k
            if (threadIdx.x < d) m_post[(s * n + k) * d + threadIdx.x] = mk[threadIdx.x];
            gemm<DP, 1, 0, 0, K_A_UPPER, O_FULL>(Linv, Linv, Q, 1.f);                  // Delta_k^-1
            __syncthreads();
            (void)factor_invert<DP, true>(Q, Qi, bad, sm.scratch());
            store_tile_lower<DP>(chol_dinv + (s * n + k) * dd, Q, d);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && bad && info) raise_info(info);
}

// StateSpaceModel._build_precision (+ H^T R^-1 H, + information vector): workgroup per (series, block)
template <int DP>
__global__ void __launch_bounds__(NTHR) bigop_ssm_precision_kernel(BigArgs a, real* __restrict__ diag, real* __restrict__ sub,
                                                                  real* __restrict__ eta) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long id = blockIdx.x, s = id / a.Tn, k = id % a.Tn;
    const int d = a.d, m = a.m;
    const long dd = (long)d * d, nt = a.Tn - 1;
    real *C = sm.tile(0), *Ci = sm.tile(1), *Am = sm.tile(2), *Bm = sm.tile(3), *Dn = sm.tile(4), *Sb = sm.tile(5);
    real *mv = sm.vec(0), *w = sm.vec(1), *rn = sm.vec(2), *btw = sm.vec(3);
    bool bad = false;
    load_tile<DP>(C, k == 0 ? a.cholP0 + s * dd : a.cholQ + (s * nt + k - 1) * dd, nullptr, d, true, true);
    if (eta) load_vec_lds<DP>(mv, k == 0 ? a.mu0 + s * d : a.b + (s * nt + k - 1) * d, nullptr, d);
    if (a.H && !a.rinv_per_step) for (int e = threadIdx.x; e < m * m; e += NTHR) sm.Rs()[e] = a.Rinv[e];
    if (threadIdx.x < 64) rn[threadIdx.x] = 0;
    __syncthreads();
    (void)factor_invert<DP, false>(C, Ci, bad, sm.scratch());
    gemm<DP, 1, 0, 0, K_A_UPPER, O_FULL>(Ci, Ci, Dn, 1.f);
    if (eta) {
        matvec<DP, 0>(Ci, mv, w, 1.f, 0.f, sm.scratch());
        matvec<DP, 1>(Ci, w, rn, 1.f, 0.f, sm.scratch());
    }
    __syncthreads();
    if (a.H) {
        const real* Rk = a.rinv_per_step ? a.Rinv + (s * a.Tn + k) * m * m : nullptr;
        const real* yk = a.y ? a.y + (s * a.Tn + k) * m : nullptr;      // NULL: no observation term in eta
        (void)obs_terms<DP>(sm, Dn, rn, a.H + (s * a.Tn + k) * m * d, yk, Rk, d, m);
    }
    if (k + 1 < a.Tn) {
        load_tile<DP>(C, a.cholQ + (s * nt + k) * dd, nullptr, d, true, true);
        load_tile<DP>(Am, a.A + (s * nt + k) * dd, nullptr, d, false, false);
        if (eta) load_vec_lds<DP>(mv, a.b + (s * nt + k) * d, nullptr, d);
        __syncthreads();
        (void)factor_invert<DP, false>(C, Ci, bad, sm.scratch());
        gemm<DP, 0, 0, 0, K_A_LOWER, O_FULL>(Ci, Am, Bm, 1.f);
        __syncthreads();
        gemm<DP, 1, 0, 1, K_FULL, O_FULL>(Bm, Bm, Dn, 1.f);
        gemm<DP, 1, 0, 0, K_A_UPPER, O_FULL>(Ci, Bm, Sb, -1.f);
        if (eta) {
            matvec<DP, 0>(Ci, mv, w, 1.f, 0.f, sm.scratch());
            matvec<DP, 1>(Bm, w, btw, 1.f, 0.f, sm.scratch());
            if (threadIdx.x < DP) rn[threadIdx.x] -= btw[threadIdx.x];
        }
        __syncthreads();
        store_tile<DP>(sub + (s * nt + k) * dd, Sb, d);
    }
    __syncthreads();
    store_tile<DP>(diag + id * dd, Dn, d);
    if (eta && threadIdx.x < d) eta[id * d + threadIdx.x] = rn[threadIdx.x];
}

// StateSpaceModel.marginal_means / sample propagation: one wavefront per series, lane = state component.  The lane holds ITS ROW
// of the next transition in registers, fetched one step ahead; the current mean reaches the other lanes by v_readlane - no LDS, no
// barrier (round 6; the first form wrote the mean to LDS behind two workgroup barriers per step and loaded its row element by
// element after them: 1.2 / 2.1 ms at B = 512, T = 1000, d = 16 / 32 in fp64, a third of StateSpaceModel.kl_divergence there).
// P > 1: workgroup (r, c) walks positions (c Lc, (c + 1) Lc] from the mean m_in[r, c] of position c Lc (wave_means_up_kernel /
// wave_means_boundary_kernel, mf_wave_ops.hpp: the composed maps of the chunks).
__global__ void __launch_bounds__(64) bigop_means_kernel(long Bl, long Br, long Tn, int d, const real* __restrict__ A,
                                                        const real* __restrict__ offs, real* __restrict__ out, long P, long Lc,
                                                        const real* __restrict__ m_in) {
    const long r = P > 1 ? blockIdx.x / P : blockIdx.x, c = P > 1 ? blockIdx.x % P : 0, s = r % Bl;
    const long k_lo = P > 1 ? c * Lc + 1 : 1, k_hi = P > 1 ? ((c + 1) * Lc + 1 < Tn ? (c + 1) * Lc + 1 : Tn) : Tn;
    const int i = threadIdx.x;
    const int ii = i < d ? i : d - 1;             // the idle lanes shadow the last row: every load stays in range
    const long dd = (long)d * d;
    real cur = c == 0 ? offs[r * Tn * d + ii] : m_in[(r * P + c) * d + ii];
    if (c == 0 && i < d) out[r * Tn * d + i] = cur;
    real a[64], an[64];
    auto load_row = [&](long k, real (&dst)[64]) {
        const real* Am = A + (s * (Tn - 1) + k) * dd + (long)ii * d;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            if (16 * cc >= d) break;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int j = 16 * cc + jj;
                dst[j] = j < d ? Am[j] : real(0);
            }
        }
    };
    if (k_hi > k_lo) load_row(k_lo - 1, a);
    for (long k = k_lo; k < k_hi; ++k) {
        // (the offset is requested BEFORE the prefetch of the next row: loads are waited for in order, and a wait for the offset
        // must not include the row that was only just requested)
        const real o = offs[(r * Tn + k) * d + ii];
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < k_hi) load_row(k, an);
        real acc0 = 0, acc1 = 0;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            if (16 * cc >= d) break;
#pragma unroll
            for (int jj = 0; jj < 16; jj += 2) {
                acc0 += a[16 * cc + jj] * bcast(cur, 16 * cc + jj);             // (columns beyond d hold zeros)
                acc1 += a[16 * cc + jj + 1] * bcast(cur, 16 * cc + jj + 1);
            }
        }
        cur = acc0 + acc1 + o;
        if (i < d) out[(r * Tn + k) * d + i] = cur;
#pragma unroll
        for (int j = 0; j < 64; ++j) a[j] = an[j];
    }
}

// out[s, k] = X[s, k] Y[s, k]: one 64-thread workgroup per block, thread = output row
__global__ void __launch_bounds__(64) bigop_block_matmul_kernel(long B, long n, int d, const real* __restrict__ X, long xs,
                                                               const real* __restrict__ Y, long ys, real* __restrict__ out) {
    const long id = blockIdx.x, s = id / n, k = id % n;
    const int i = threadIdx.x;
    if (i >= d) return;
    const long dd = (long)d * d;
    const real* Xr = X + (s * xs + k) * dd + (long)i * d;
    const real* Ym = Y + (s * ys + k) * dd;
    real* o = out + id * dd + (long)i * d;
    for (int j = 0; j < d; ++j) {
        real a = 0;
        for (int l = 0; l < d; ++l) a += Xr[l] * Ym[(long)l * d + j];
        o[j] = a;
    }
}

// ---- StateSpaceModel.marginal_covariances / subsequent_covariances for large d, partitioned in time --------------------------
// The reference takes the block diagonal of the inverse of the assembled precision (state_space_model.py:254-262); the forward
// recursion  S_0 = P0,  S_{k+1} = A_k S_k A_k^T + Q_k  gives the same blocks with three products per step and, being a
// congruence recursion, splits over time like the small-d scan (mf_btd_par.hpp): pass 0 - workgroup (series, chunk) composes the
// chunk's map  S -> M S M^T + N  (M = product of the chunk's A, N = the recursion started from zero); pass 1 - one workgroup
// per series walks the P chunk boundaries; pass 2 - workgroup (series, chunk) restarts from its boundary value and writes
// every block (and A_k S_k = Cov(x_{k+1}, x_k)).  7 products per step on P chunks instead of 3 on one workgroup per series.
struct BigMeanArgs {
    const real *mu0, *b;      // [B, d], [B, n-1, d]; b == NULL: covariances only
    real* out;                // [B, n, d] (emit pass)
    real* wsv;                // [B, P, d]: chunk offsets v_c (pass 0)
    real* start;              // [B, P, d]: means at the chunk boundaries (pass 1 -> 2)
};
template <int DP, bool EMIT>
__global__ void __launch_bounds__(NTHR) bigop_cov_chunk_kernel(long B, long n, int d, long P, long L, const real* __restrict__ cholP0,
                                                              const real* __restrict__ A, const real* __restrict__ cholQ,
                                                              real* __restrict__ wsM, real* __restrict__ wsN,
                                                              const real* __restrict__ start, real* __restrict__ ocov,
                                                              real* __restrict__ osub, BigMeanArgs mean) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long s = blockIdx.x / P, c = blockIdx.x % P;
    real *mv = sm.vec(0), *mt = sm.vec(1);                        // the mean and its successor (marginal means ride along)
    const long nt = n - 1, dd = (long)d * d, k0 = c * L;
    long k1 = k0 + L;
    if (k1 > nt) k1 = nt;
    if (!EMIT && (c + 1 == P || k0 >= k1)) return;               // the last chunk's map is never applied
    real *At = sm.tile(0), *Ct = sm.tile(1), *S = sm.tile(2), *T1 = sm.tile(3), *M = sm.tile(4);
    if (EMIT) {
        if (c == 0) {
            load_tile<DP>(Ct, cholP0 + s * dd, nullptr, d, true, false);
            __syncthreads();
            gemm<DP, 0, 1, 0, K_A_LOWER, O_FULL>(Ct, Ct, S, 1.f);                    // P0 = C0 C0^T
            __syncthreads();
            store_tile<DP>(ocov + (s * n) * dd, S, d);
        } else {
            load_tile<DP>(S, start + (s * P + c) * dd, nullptr, d, false, false);
        }
        if (mean.out) {
            load_vec_lds<DP>(mv, c == 0 ? mean.mu0 + s * d : mean.start + (s * P + c) * d, nullptr, d);
            if (c == 0 && threadIdx.x < d) mean.out[(s * n) * d + threadIdx.x] = mean.mu0[s * d + threadIdx.x];
        }
    } else {
        zero_tile<DP>(S);
        for (int e = threadIdx.x; e < DP * DP; e += NTHR) M[(e / DP) * Geo<DP>::LD + (e % DP)] = (e / DP == e % DP) ? 1.f : 0.f;
        if (threadIdx.x < DP) mv[threadIdx.x] = 0.f;
    }
    __syncthreads();
    for (long k = k0; k < k1; ++k) {
        load_tile<DP>(At, A + (s * nt + k) * dd, nullptr, d, false, false);
        load_tile<DP>(Ct, cholQ + (s * nt + k) * dd, nullptr, d, true, false);
        if (mean.b) load_vec_lds<DP>(mt, mean.b + (s * nt + k) * d, nullptr, d);
        __syncthreads();
        gemm<DP, 0, 0, 0, K_FULL, O_FULL>(At, S, T1, 1.f);                           // T1 = A S
        if (mean.b) {
            matvec<DP, 0>(At, mv, mt, 1.f, 1.f, sm.scratch());                       // b + A m (ends with a barrier)
            real* t = mv; mv = mt; mt = t;
            if (EMIT && threadIdx.x < d) mean.out[(s * n + k + 1) * d + threadIdx.x] = mv[threadIdx.x];
        } else {
            __syncthreads();
        }
        if (EMIT && osub) store_tile<DP>(osub + (s * nt + k) * dd, T1, d);
        gemm<DP, 0, 1, 0, K_FULL, O_FULL>(T1, At, S, 1.f);                           // S = A S A^T
        gemm<DP, 0, 1, 1, K_A_LOWER, O_FULL>(Ct, Ct, S, 1.f);                        //   + C C^T (same output tiles per wave)
        __syncthreads();
        if (EMIT) {
            store_tile<DP>(ocov + (s * n + k + 1) * dd, S, d);
        } else {
            gemm<DP, 0, 0, 0, K_FULL, O_FULL>(At, M, T1, 1.f);                       // M <- A M
            real* t = M; M = T1; T1 = t;
        }
        __syncthreads();
    }
    if (!EMIT) {
        store_tile<DP>(wsM + (s * P + c) * dd, M, d);
        store_tile<DP>(wsN + (s * P + c) * dd, S, d);
        if (mean.b && threadIdx.x < d) mean.wsv[(s * P + c) * d + threadIdx.x] = mv[threadIdx.x];
    }
}
// pass 1: start[c + 1] = M_c start[c] M_c^T + N_c along the chunk boundaries of one series
template <int DP>
__global__ void __launch_bounds__(NTHR) bigop_cov_boundary_kernel(long B, int d, long P, const real* __restrict__ cholP0,
                                                                 const real* __restrict__ wsM, const real* __restrict__ wsN,
                                                                 real* __restrict__ start, BigMeanArgs mean) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    constexpr int LD = Geo<DP>::LD;
    const long s = blockIdx.x, dd = (long)d * d;
    real *Mt = sm.tile(0), *Nt = sm.tile(1), *S = sm.tile(2), *T1 = sm.tile(3);
    real *mv = sm.vec(0), *mt = sm.vec(1);
    load_tile<DP>(Nt, cholP0 + s * dd, nullptr, d, true, false);
    if (mean.b) load_vec_lds<DP>(mv, mean.mu0 + s * d, nullptr, d);
    __syncthreads();
    gemm<DP, 0, 1, 0, K_A_LOWER, O_FULL>(Nt, Nt, S, 1.f);
    __syncthreads();
    for (long c = 0; c + 1 < P; ++c) {
        load_tile<DP>(Mt, wsM + (s * P + c) * dd, nullptr, d, false, false);
        load_tile<DP>(Nt, wsN + (s * P + c) * dd, nullptr, d, false, false);
        if (mean.b) load_vec_lds<DP>(mt, mean.wsv + (s * P + c) * d, nullptr, d);
        __syncthreads();
        gemm<DP, 0, 0, 0, K_FULL, O_FULL>(Mt, S, T1, 1.f);
        if (mean.b) {
            matvec<DP, 0>(Mt, mv, mt, 1.f, 1.f, sm.scratch());                       // v_c + M_c m
            real* t = mv; mv = mt; mt = t;
            if (threadIdx.x < d) mean.start[(s * P + c + 1) * d + threadIdx.x] = mv[threadIdx.x];
        } else {
            __syncthreads();
        }
        gemm<DP, 0, 1, 0, K_FULL, O_FULL>(T1, Mt, S, 1.f);
        __syncthreads();
        for (int e = threadIdx.x; e < DP * DP; e += NTHR) S[(e / DP) * LD + (e % DP)] += Nt[(e / DP) * LD + (e % DP)];
        __syncthreads();
        store_tile<DP>(start + (s * P + c + 1) * dd, S, d);
        __syncthreads();
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
#define MF_BIGOP_DISPATCH(CALL)                    \
    if (d <= 16) { CALL(16) }                      \
    else if (d <= 32) { CALL(32) }                 \
    else if (sizeof(real) == 4 && d <= 48) { CALL(48) } \
    else if (sizeof(real) == 4 && d <= 64) { CALL(64) } \
    else return -100;

template <typename K> inline bool big_attr(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}
inline int big_ok() { return hipGetLastError() == hipSuccess ? 0 : -1000; }

// MF_WAVE=0 (experiment builds only, mf_env.hpp): the operators of 16 <= d <= 32 stay on the tile engine
inline bool wave_off() {
    static const bool off = [] { const char* e = mf_knob("MF_WAVE"); return e && e[0] == '0'; }();
    return off;
}
inline int op_cholesky(long B, long n, int d, const real* diag, const real* sub, real* ldiag, real* lsub, int* info, hipStream_t st) {
#define MF_C(DP)                                                                                                       \
    { static const bool ok = big_attr(&bigop_cholesky_kernel<DP>, Smem<DP>::BYTES); if (!ok) return -1000;               \
      hipLaunchKernelGGL((bigop_cholesky_kernel<DP>), dim3((unsigned)B), dim3(NTHR), Smem<DP>::BYTES, st, B, n, d, diag, sub, ldiag, lsub, info); }
    MF_BIGOP_DISPATCH(MF_C)
#undef MF_C
    return big_ok();
}
inline int op_solve(long Bl, long Br, long n, int d, const real* ldiag, const real* lsub, const real* rhs, real* out, int transpose, hipStream_t st) {
#define MF_C(DP)                                                                                                       \
    { static const bool ok = big_attr(&bigop_solve_kernel<DP>, Smem<DP>::BYTES); if (!ok) return -1000;                  \
      hipLaunchKernelGGL((bigop_solve_kernel<DP>), dim3((unsigned)Br), dim3(NTHR), Smem<DP>::BYTES, st, Bl, Br, n, d, ldiag, lsub, rhs, out, transpose); }
    MF_BIGOP_DISPATCH(MF_C)
#undef MF_C
    return big_ok();
}
inline int op_matvec(long Bl, long Br, long n, int d, const real* diag, const real* sub, const real* x, real* out, int mode, hipStream_t st) {
    hipLaunchKernelGGL(bigop_matvec_kernel, dim3((unsigned)(Br * n)), dim3(64), 0, st, Bl, Br, n, d, diag, sub, x, out, mode);
    return big_ok();
}
inline int op_logdet(long B, long n, int d, const real* ldiag, real* out, hipStream_t st) {
    hipLaunchKernelGGL(bigop_logdet_kernel, dim3((unsigned)B), dim3(64), 0, st, B, n, d, ldiag, out);
    return big_ok();
}
inline int op_diag_of_inverse(long B, long n, int d, const real* ldiag, const real* lsub, real* odiag, real* osub, hipStream_t st) {
#define MF_C(DP)                                                                                                       \
    { static const bool ok = big_attr(&bigop_diag_of_inverse_kernel<DP>, Smem<DP>::BYTES); if (!ok) return -1000;        \
      hipLaunchKernelGGL((bigop_diag_of_inverse_kernel<DP>), dim3((unsigned)B), dim3(NTHR), Smem<DP>::BYTES, st, B, n, d, ldiag, lsub, odiag, osub); }
    MF_BIGOP_DISPATCH(MF_C)
#undef MF_C
    return big_ok();
}
inline int op_udl(long B, long n, int d, const real* diag, const real* sub, real* ut, real* chol_d, const real* eta, real* m_post,
                  real* chol_dinv, int* info, hipStream_t st) {
#define MF_C(DP)                                                                                                       \
    { static const bool ok = big_attr(&bigop_udl_kernel<DP>, Smem<DP>::BYTES); if (!ok) return -1000;                    \
      hipLaunchKernelGGL((bigop_udl_kernel<DP>), dim3((unsigned)B), dim3(NTHR), Smem<DP>::BYTES, st, B, n, d, diag, sub, ut, chol_d, eta, m_post, chol_dinv, info); }
    MF_BIGOP_DISPATCH(MF_C)
#undef MF_C
    return big_ok();
}
inline int op_ssm_precision(long B, long Tn, int d, int m, const real* mu0, const real* cholP0, const real* A, const real* b,
                            const real* cholQ, const real* H, const real* y, const real* Rinv, int rinv_per_step, real* diag,
                            real* sub, real* eta, hipStream_t st) {
    if (H && (m < 1 || m > MAXM_BIG)) return -4;
    {   // 16 <= d <= 32, at most four outputs: a wavefront per block on register tiles (mf_wave.hpp)
        static const bool off = [] { const char* e = mf_knob("MF_WAVE"); return e && e[0] == '0'; }();
        if (!off && B * Tn > 0) {
            const int rc = wave_ssm_precision(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, diag, sub, eta, st);
            if (rc != -101) return rc;
        }
    }
    {   // 32 < d <= 64: the panel kernels' per-block terms without the elimination (mf_panel.hpp, PREC), chunks of 16 blocks
        static const bool poff = std::getenv("MF_PANEL_PREC_OFF") != nullptr;      // (A/B switch)
        if (!poff && B * Tn > 0) {
            const int rc = panel_ssm_precision(B, Tn, d, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, diag, sub, eta, st);
            if (rc != -101) return rc;
        }
    }
    BigArgs a{B, Tn, d, H ? m : 1, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, 1, 1, nullptr};
#define MF_C(DP)                                                                                                       \
    { static const bool ok = big_attr(&bigop_ssm_precision_kernel<DP>, Smem<DP>::BYTES); if (!ok) return -1000;          \
      hipLaunchKernelGGL((bigop_ssm_precision_kernel<DP>), dim3((unsigned)(B * Tn)), dim3(NTHR), Smem<DP>::BYTES, st, a, diag, sub, eta); }
    MF_BIGOP_DISPATCH(MF_C)
#undef MF_C
    return big_ok();
}
inline int op_means(long Bl, long Br, long Tn, int d, const real* A, const real* offs, real* out, hipStream_t st, long P = 1,
                    long Lc = 0, const real* m_in = nullptr) {
    hipLaunchKernelGGL(bigop_means_kernel, dim3((unsigned)(Br * P)), dim3(64), 0, st, Bl, Br, Tn, d, A, offs, out, P, Lc, m_in);
    return big_ok();
}
// chunks per series of the covariance recursion: about two workgroups per CU, chunks of at least eight transitions
inline long cov_chunks(long B, long n) {
    long P = 512 / (B > 0 ? B : 1);
    if (P < 1) P = 1;
    while (P > 1 && (n - 1) / P < 8) --P;
    return P;
}
inline size_t marginal_covs_ws_engine(long B, long n, int d) {
    const long P = cov_chunks(B, n);
    return P > 1 ? (3 * size_t(B) * P * d * d + 2 * size_t(B) * P * d) * sizeof(real) : 0;
}
// (the query does not know which engine takes the call: the larger of the tile engine's and the wave kernels' partition)
inline size_t marginal_covs_ws(long B, long n, int d) {
    const size_t e = marginal_covs_ws_engine(B, n, d), w = wave_marg_ws(B, n, d, (int)sizeof(real));
    return e > w ? e : w;
}
// mu0, b, omean all NULL: covariances only
inline int op_marginal_covs(long B, long n, int d, const real* mu0, const real* cholP0, const real* A, const real* b,
                            const real* cholQ, real* omean, real* ocov, real* osub, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!wave_off()) {   // 16 <= d <= 32, many series: one wavefront per series walks the forward recursion (mf_wave_ops.hpp)
        const int rc = wave_ssm_marginals<real>(B, n, d, omean ? mu0 : nullptr, cholP0, A, omean ? b : nullptr, cholQ, omean, ocov, osub, ws,
                                                ws_bytes, st);
        if (rc != -101) return rc;
    }
    long P = cov_chunks(B, n);
    if (P > 1 && (ws == nullptr || ws_bytes < marginal_covs_ws_engine(B, n, d))) P = 1;
    const long L = (n - 1 + P - 1) / P;
    P = (n - 1 + L - 1) / L;                                   // no empty chunks
    real* wsM = static_cast<real*>(ws);
    real* wsN = P > 1 ? wsM + size_t(B) * P * d * d : nullptr;
    real* start = P > 1 ? wsN + size_t(B) * P * d * d : nullptr;
    real* wsv = P > 1 ? start + size_t(B) * P * d * d : nullptr;
    const BigMeanArgs mean{mu0, omean ? b : nullptr, omean, wsv, P > 1 ? wsv + size_t(B) * P * d : nullptr};
#define MF_C(DP)                                                                                                       \
    { static const bool ok = big_attr(&bigop_cov_chunk_kernel<DP, false>, Smem<DP>::BYTES) &&                            \
                             big_attr(&bigop_cov_chunk_kernel<DP, true>, Smem<DP>::BYTES) &&                             \
                             big_attr(&bigop_cov_boundary_kernel<DP>, Smem<DP>::BYTES);                                  \
      if (!ok) return -1000;                                                                                           \
      if (P > 1) {                                                                                                     \
          hipLaunchKernelGGL((bigop_cov_chunk_kernel<DP, false>), dim3((unsigned)(B * P)), dim3(NTHR), Smem<DP>::BYTES, st, B, n, d, P, L, \
                             cholP0, A, cholQ, wsM, wsN, static_cast<const real*>(nullptr), static_cast<real*>(nullptr),  \
                             static_cast<real*>(nullptr), mean);                                                        \
          hipLaunchKernelGGL((bigop_cov_boundary_kernel<DP>), dim3((unsigned)B), dim3(NTHR), Smem<DP>::BYTES, st, B, d, P, cholP0, \
                             static_cast<const real*>(wsM), static_cast<const real*>(wsN), start, mean);                \
      }                                                                                                                \
      hipLaunchKernelGGL((bigop_cov_chunk_kernel<DP, true>), dim3((unsigned)(B * P)), dim3(NTHR), Smem<DP>::BYTES, st, B, n, d, P, L, \
                         cholP0, A, cholQ, static_cast<real*>(nullptr), static_cast<real*>(nullptr),                    \
                         static_cast<const real*>(start), ocov, osub, mean); }
    MF_BIGOP_DISPATCH(MF_C)
#undef MF_C
    return big_ok();
}
inline int op_block_matmul(long B, long n, int d, const real* X, long xs, const real* Y, long ys, real* out, hipStream_t st) {
    hipLaunchKernelGGL(bigop_block_matmul_kernel, dim3((unsigned)(B * n)), dim3(64), 0, st, B, n, d, X, xs, Y, ys, out);
    return big_ok();
}

}  // namespace MF_BIG_NS
}  // namespace mf
