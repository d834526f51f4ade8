// Time-partitioned forms of the LARGE-d operators of mf_bigops_impl.hpp for FEW series (BASELINE config 5: B = 8, T = 2048,
// d = 64): SymmetricBlockTriDiagonal.cholesky (block_tri_diag.py:423-436), upper_diagonal_lower + the posterior chain
// (block_tri_diag.py:438-545, kalman_filter.py:159-174) and LowerTriangularBlockTriDiagonal.solve (block_tri_diag.py:339-351).
// The one-workgroup-per-series kernels keep 8 of 256 CUs busy for T dependent block steps; here a series is cut into P chunks
// of L blocks and every pass is one launch of B x P workgroups on the same LDS-tile / MFMA engine:
//
//   up        chunk c eliminates its blocks in natural order, carrying the fill-in towards the block on its left as a spike
//             (the log-likelihood kernel's elimination, mf_big_impl.hpp) and leaves (Dv, GU, F): the pivot of its LAST block
//             as a function of the unknown pivot Sigma of the block before the chunk,
//                 Sigma_last = Dv - F (Sigma + GU)^-1 F^T                                   (mf_btd_par.hpp, small-d form)
//   boundary  one workgroup per series walks the P chunk ends with that map: the natural-order pivots at the chunk ends
//   emit      chunk c restarts the textbook recursion from the pivot left of it and writes the factor (or the U D U^T
//             factors and the chain's Cholesky factors) of its own blocks
//
// U D U^T is the same on the block-reversed matrix (position p <-> block n-1-p, couplings transposed).  The right-hand-side
// recursions (solve; x_k = eta_k - U_k x_{k+1} of the posterior chain) are affine in the carried vector: the emitting pass also
// composes its chunk's map (N, a), and a last pass restarts every chunk from its true boundary vector.
// Sequential depth: about 3 L + P block steps instead of n.  Included once per scalar type, after mf_bigops_impl.hpp.
namespace mf {
namespace MF_BIG_NS {

// tile <- g^T (zero padded); consecutive threads read consecutive addresses
template <int DP> __device__ __forceinline__ void load_tile_t(real* __restrict__ tile, const real* __restrict__ g, int d) {
    constexpr int LD = Geo<DP>::LD;
    for (int e = threadIdx.x; e < DP * DP; e += NTHR) {
        const int col = e / DP, row = e % DP;
        tile[row * LD + col] = (row < d && col < d) ? g[col * d + row] : real(0);
    }
}
template <int DP> __device__ __forceinline__ void add_tile(real* __restrict__ dst, const real* __restrict__ src) {
    constexpr int LD = Geo<DP>::LD;
    for (int e = threadIdx.x; e < DP * DP; e += NTHR) dst[(e / DP) * LD + (e % DP)] += src[(e / DP) * LD + (e % DP)];
}
template <int DP> __device__ __forceinline__ void identity_tile(real* __restrict__ tile) {
    constexpr int LD = Geo<DP>::LD;
    for (int e = threadIdx.x; e < DP * DP; e += NTHR) tile[(e / DP) * LD + (e % DP)] = (e / DP == e % DP) ? real(1) : real(0);
}

// ---- factorisation: up-sweep ----------------------------------------------------------------------------------------------------
// workgroup (series s, chunk c), c < P - 1 (nobody needs the map of the last chunk).  REV: positions run backwards over the blocks.
template <int DP, bool REV>
__global__ void __launch_bounds__(NTHR) bigpar_chol_up_kernel(long B, long n, int d, long P, long L, const real* __restrict__ diag,
                                                             const real* __restrict__ sub, real* __restrict__ oDv,
                                                             real* __restrict__ oGU, real* __restrict__ oF, int* info) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long s = blockIdx.x / (P - 1), c = blockIdx.x % (P - 1);
    const long k0 = c * L, dd = (long)d * d;
    long k1 = k0 + L;
    if (k1 > n) k1 = n;
    real *Phi = sm.tile(0), *X = sm.tile(1), *GU = sm.tile(2), *W = sm.tile(3), *Linv = sm.tile(4), *Fin = sm.tile(5), *V = sm.tile(6);
    const bool spike = c > 0;
    bool bad = false;
    load_tile<DP>(Phi, diag + (s * n + (REV ? n - 1 - k0 : k0)) * dd, nullptr, d, false, true);
    if (spike) {
        // coupling of position k0 with position k0 - 1, rows: k0
        if (!REV) load_tile<DP>(X, sub + (s * (n - 1) + k0 - 1) * dd, nullptr, d, false, false);
        else load_tile_t<DP>(X, sub + (s * (n - 1) + n - 1 - k0) * dd, d);
    }
    zero_tile<DP>(GU);
    __syncthreads();
    for (long k = k0 + 1; k < k1; ++k) {
        (void)factor_invert<DP, true>(Phi, Linv, bad, sm.scratch());                    // block k-1: pivot complete
        load_tile<DP>(Phi, diag + (s * n + (REV ? n - 1 - k : k)) * dd, nullptr, d, false, true);
        load_tile<DP>(Fin, sub + (s * (n - 1) + (REV ? n - 1 - k : k - 1)) * dd, nullptr, d, false, false);
        __syncthreads();
        if (!REV) gemm<DP, 0, 1, 0, K_B_UPPER, O_FULL>(Fin, Linv, W, 1.f);              // W = F_k L^-T
        else gemm<DP, 0, 0, 0, K_A_LOWER, O_FULL>(Linv, Fin, W, 1.f);                   // W^T = L^-1 S_k   (F_k = S_k^T)
        if (spike) gemm<DP, 0, 0, 0, K_A_LOWER, O_FULL>(Linv, X, V, 1.f);               // V = L^-1 X
        __syncthreads();
        if (!REV) gemm<DP, 0, 1, 1, K_FULL, O_FULL>(W, W, Phi, -1.f);                   // next pivot: D_k - W W^T
        else gemm<DP, 1, 0, 1, K_FULL, O_FULL>(W, W, Phi, -1.f);
        if (spike) {
            gemm<DP, 1, 0, 1, K_FULL, O_FULL>(V, V, GU, -1.f);                          // GU -= V^T V
            if (!REV) gemm<DP, 0, 0, 0, K_FULL, O_FULL>(W, V, X, -1.f);                 // next coupling to the left block: -W V
            else gemm<DP, 1, 0, 0, K_FULL, O_FULL>(W, V, X, -1.f);
        }
        __syncthreads();
    }
    const long id = s * P + c;
    store_tile<DP>(oDv + id * dd, Phi, d);
    if (spike) {
        store_tile<DP>(oGU + id * dd, GU, d);
        store_tile<DP>(oF + id * dd, X, d);
    }
    if (threadIdx.x == 0 && bad && info) raise_info(info);
}

// ---- factorisation: the chunk ends of one series ------------------------------------------------------------------------------------
// piv[s, c] = natural-order pivot of the last block of chunk c, c = 0 ... P-2
template <int DP>
__global__ void __launch_bounds__(NTHR) bigpar_chol_boundary_kernel(long B, int d, long P, const real* __restrict__ Dv,
                                                                   const real* __restrict__ GU, const real* __restrict__ F,
                                                                   real* __restrict__ piv, int* info) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long s = blockIdx.x, dd = (long)d * d;
    real *S = sm.tile(0), *Linv = sm.tile(1), *Fc = sm.tile(2), *W = sm.tile(3), *G = sm.tile(4);
    bool bad = false;
    load_tile<DP>(S, Dv + (s * P) * dd, nullptr, d, false, true);
    __syncthreads();
    store_tile<DP>(piv + (s * P) * dd, S, d);
    for (long c = 1; c + 1 < P; ++c) {
        load_tile<DP>(G, GU + (s * P + c) * dd, nullptr, d, false, false);
        load_tile<DP>(Fc, F + (s * P + c) * dd, nullptr, d, false, false);
        __syncthreads();
        add_tile<DP>(S, G);                                                             // pivot left of the chunk once its interior is gone
        __syncthreads();
        (void)factor_invert<DP, true>(S, Linv, bad, sm.scratch());
        load_tile<DP>(S, Dv + (s * P + c) * dd, nullptr, d, false, true);
        gemm<DP, 0, 1, 0, K_B_UPPER, O_FULL>(Fc, Linv, W, 1.f);
        __syncthreads();
        gemm<DP, 0, 1, 1, K_FULL, O_FULL>(W, W, S, -1.f);
        __syncthreads();
        store_tile<DP>(piv + (s * P + c) * dd, S, d);
    }
    if (threadIdx.x == 0 && bad && info) raise_info(info);
}

// ---- cholesky: emit ---------------------------------------------------------------------------------------------------------------
template <int DP>
__global__ void __launch_bounds__(NTHR) bigpar_chol_emit_kernel(long B, long n, int d, long P, long L, const real* __restrict__ diag,
                                                               const real* __restrict__ sub, const real* __restrict__ piv,
                                                               real* __restrict__ ldiag, real* __restrict__ lsub, int* info) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long s = blockIdx.x / P, c = blockIdx.x % P;
    const long k0 = c * L, dd = (long)d * d;
    long k1 = k0 + L;
    if (k1 > n) k1 = n;
    real *S = sm.tile(0), *Linv = sm.tile(1), *W = sm.tile(2), *Fin = sm.tile(3);
    bool bad = false;
    if (c > 0) {
        load_tile<DP>(S, piv + (s * P + c - 1) * dd, nullptr, d, false, true);
        __syncthreads();
        (void)factor_invert<DP, true>(S, Linv, bad, sm.scratch());
    }
    for (long k = k0; k < k1; ++k) {
        load_tile<DP>(S, diag + (s * n + k) * dd, nullptr, d, false, true);
        if (k > 0) {
            load_tile<DP>(Fin, sub + (s * (n - 1) + k - 1) * dd, nullptr, d, false, false);
            __syncthreads();
            gemm<DP, 0, 1, 0, K_B_UPPER, O_FULL>(Fin, Linv, W, 1.f);                    // W = S_{k-1} L_{k-1}^-T
            __syncthreads();
            store_tile<DP>(lsub + (s * (n - 1) + k - 1) * dd, W, d);
            gemm<DP, 0, 1, 1, K_FULL, O_FULL>(W, W, S, -1.f);                           // D_k - W W^T
        }
        __syncthreads();
        (void)factor_invert<DP, true>(S, Linv, bad, sm.scratch());
        store_tile_lower<DP>(ldiag + (s * n + k) * dd, S, d);
        __syncthreads();
    }
    if (threadIdx.x == 0 && bad && info) raise_info(info);
}

// ---- U D U^T + posterior chain: emit ---------------------------------------------------------------------------------------------------
// positions p = n-1-k run over chunk c; with eta the recursion x_p = eta_p - U_p x_{p-1} is started from ZERO (-> a_c) and the chunk's
// linear part N_c = prod(-U_p) is composed beside it; the chain's means need the true x and are written by the next kernel.
template <int DP>
__global__ void __launch_bounds__(NTHR) bigpar_udl_emit_kernel(long B, long n, int d, long P, long L, const real* __restrict__ diag,
                                                              const real* __restrict__ sub, const real* __restrict__ piv,
                                                              real* __restrict__ ut, real* __restrict__ chol_d,
                                                              const real* __restrict__ eta, real* __restrict__ chol_dinv,
                                                              real* __restrict__ oN, real* __restrict__ oa, int* info) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long s = blockIdx.x / P, c = blockIdx.x % P;
    const long p0 = c * L, dd = (long)d * d;
    long p1 = p0 + L;
    if (p1 > n) p1 = n;
    real *Dl = sm.tile(0), *Linv = sm.tile(1), *S = sm.tile(2), *U = sm.tile(3), *Ut = sm.tile(4), *N = sm.tile(5), *N2 = sm.tile(6);
    real *x = sm.vec(0), *xp = sm.vec(1);
    const bool compose = eta && c > 0 && c + 1 < P;
    bool bad = false;
    if (c > 0) {
        load_tile<DP>(Dl, piv + (s * P + c - 1) * dd, nullptr, d, false, true);
        __syncthreads();
        (void)factor_invert<DP, true>(Dl, Linv, bad, sm.scratch());
    }
    if (threadIdx.x < 64) xp[threadIdx.x] = 0;
    if (compose) identity_tile<DP>(N);
    __syncthreads();
    for (long p = p0; p < p1; ++p) {
        const long k = n - 1 - p;
        load_tile<DP>(Dl, diag + (s * n + k) * dd, nullptr, d, false, true);
        if (eta) load_vec_lds<DP>(x, eta + (s * n + k) * d, nullptr, d);
        const bool coupled = p > 0;
        if (coupled) load_tile<DP>(S, sub + (s * (n - 1) + k) * dd, nullptr, d, false, false);
        __syncthreads();
        if (coupled) {
            gemm<DP, 0, 0, 0, K_A_LOWER, O_FULL>(Linv, S, U, 1.f);                      // L^-1 S  (L = chol Delta_{k+1})
            __syncthreads();
            gemm<DP, 1, 0, 1, K_FULL, O_FULL>(U, U, Dl, -1.f);                          // Delta_k = D_k - S^T Delta_{k+1}^-1 S
            gemm<DP, 1, 0, 0, K_A_UPPER, O_FULL>(Linv, U, Ut, 1.f);                     // U_k^T = Delta_{k+1}^-1 S
            __syncthreads();
            store_tile<DP>(ut + (s * (n - 1) + k) * dd, Ut, d);
            if (compose) gemm<DP, 1, 0, 0, K_FULL, O_FULL>(Ut, N, N2, -1.f);            // N <- -U_k N
            if (eta) matvec<DP, 1>(Ut, xp, x, -1.f, 1.f, sm.scratch());                 // x_k = eta_k - U_k x_{k+1}
            if (compose) { real* t = N; N = N2; N2 = t; }
        }
        (void)factor_invert<DP, true>(Dl, Linv, bad, sm.scratch());
        store_tile_lower<DP>(chol_d + (s * n + k) * dd, Dl, d);
        if (eta) {
            if (threadIdx.x < 64) xp[threadIdx.x] = x[threadIdx.x];
            gemm<DP, 1, 0, 0, K_A_UPPER, O_FULL>(Linv, Linv, S, 1.f);                   // Delta_k^-1
            __syncthreads();
            (void)factor_invert<DP, true>(S, U, bad, sm.scratch());
            store_tile_lower<DP>(chol_dinv + (s * n + k) * dd, S, d);
        }
        __syncthreads();
    }
    if (eta) {
        if (threadIdx.x < d) oa[(s * P + c) * d + threadIdx.x] = xp[threadIdx.x];
        if (compose) store_tile<DP>(oN + (s * P + c) * dd, N, d);
    }
    if (threadIdx.x == 0 && bad && info) raise_info(info);
}

// ---- posterior chain: means ------------------------------------------------------------------------------------------------------------
// chunk c first walks the maps of the chunks before it (x at its left boundary), then x_p = eta_p - U_p x_{p-1} and
// m_p = Delta_p^-1 x_p = C C^T x_p with C = chol(Delta_p^-1) as written by the emit kernel.  Two tiles of LDS.
template <int DP> struct SmemVec {
    static constexpr int FLOATS = 2 * Geo<DP>::TILE + 4 * 64 + 256;
    static constexpr int BYTES = FLOATS * (int)sizeof(real);
};
template <int DP>
__global__ void __launch_bounds__(NTHR) bigpar_udl_means_kernel(long B, long n, int d, long P, long L, const real* __restrict__ ut,
                                                               const real* __restrict__ chol_dinv, const real* __restrict__ eta,
                                                               const real* __restrict__ wN, const real* __restrict__ wa,
                                                               real* __restrict__ m_post) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    real* base = reinterpret_cast<real*>(smem_raw);
    real *T0 = base, *T1 = base + Geo<DP>::TILE;
    real *x = T1 + Geo<DP>::TILE, *xp = x + 64, *t = xp + 64, *mk = t + 64, *scratch = mk + 64;
    const long s = blockIdx.x / P, c = blockIdx.x % P;
    const long p0 = c * L, dd = (long)d * d;
    long p1 = p0 + L;
    if (p1 > n) p1 = n;
    if (threadIdx.x < 64) xp[threadIdx.x] = 0;
    __syncthreads();
    for (long j = 0; j < c; ++j) {                                                       // x at the end of chunk j
        load_vec_lds<DP>(x, wa + (s * P + j) * d, nullptr, d);
        if (j > 0) load_tile<DP>(T0, wN + (s * P + j) * dd, nullptr, d, false, false);
        __syncthreads();
        if (j > 0) matvec<DP, 0>(T0, xp, x, 1.f, 1.f, scratch);                          // a_j + N_j x
        if (threadIdx.x < 64) xp[threadIdx.x] = x[threadIdx.x];
        __syncthreads();
    }
    for (long p = p0; p < p1; ++p) {
        const long k = n - 1 - p;
        load_vec_lds<DP>(x, eta + (s * n + k) * d, nullptr, d);
        load_tile<DP>(T1, chol_dinv + (s * n + k) * dd, nullptr, d, true, false);
        if (p > 0) load_tile<DP>(T0, ut + (s * (n - 1) + k) * dd, nullptr, d, false, false);
        __syncthreads();
        if (p > 0) matvec<DP, 1>(T0, xp, x, -1.f, 1.f, scratch);
        matvec<DP, 1>(T1, x, t, 1.f, 0.f, scratch);
        matvec<DP, 0>(T1, t, mk, 1.f, 0.f, scratch);
        if (threadIdx.x < d) m_post[(s * n + k) * d + threadIdx.x] = mk[threadIdx.x];
        if (threadIdx.x < 64) xp[threadIdx.x] = x[threadIdx.x];
        __syncthreads();
    }
}

// ---- solve ---------------------------------------------------------------------------------------------------------------------------
// positions p (TR: p <-> block n-1-p):  z_p = Ainv_p (r_p - C_p z_{p-1}),  Ainv = L_k^-1 / L_k^-T,  C = lsub[k-1] / lsub[k]^T.
// compose: chunk c (< P-1) runs the recursion from z = 0 (-> a_c) and composes N_c = prod(-Ainv_p C_p) beside it.
template <int DP, int TR>
__global__ void __launch_bounds__(NTHR) bigpar_solve_compose_kernel(long Bl, long Br, long n, int d, long P, long L,
                                                                   const real* __restrict__ ldiag, const real* __restrict__ lsub,
                                                                   const real* __restrict__ rhs, real* __restrict__ oN,
                                                                   real* __restrict__ oa) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long r = blockIdx.x / (P - 1), c = blockIdx.x % (P - 1), s = r % Bl;
    const long p0 = c * L, dd = (long)d * d;
    long p1 = p0 + L;
    if (p1 > n) p1 = n;
    real *Lt = sm.tile(0), *Linv = sm.tile(1), *C = sm.tile(2), *N = sm.tile(3), *T1 = sm.tile(4), *N2 = sm.tile(5);
    real *z = sm.vec(0), *x = sm.vec(1);
    bool bad = false;
    if (threadIdx.x < 64) z[threadIdx.x] = 0;
    if (c > 0) identity_tile<DP>(N);
    __syncthreads();
    for (long p = p0; p < p1; ++p) {
        const long k = TR ? n - 1 - p : p;
        load_tile<DP>(Lt, ldiag + (s * n + k) * dd, nullptr, d, true, true);
        load_vec_lds<DP>(x, rhs + (r * n + k) * d, nullptr, d);
        const bool coupled = p > 0;
        if (coupled) load_tile<DP>(C, lsub + (s * (n - 1) + (TR ? k : k - 1)) * dd, nullptr, d, false, false);
        __syncthreads();
        if (coupled) matvec<DP, TR>(C, z, x, -1.f, 1.f, sm.scratch());
        (void)factor_invert<DP, false>(Lt, Linv, bad, sm.scratch());
        matvec<DP, TR>(Linv, x, z, 1.f, 0.f, sm.scratch());
        if (c > 0) {
            gemm<DP, TR, 0, 0, K_FULL, O_FULL>(C, N, T1, 1.f);
            __syncthreads();
            gemm<DP, TR, 0, 0, TR ? K_A_UPPER : K_A_LOWER, O_FULL>(Linv, T1, N2, -1.f);
            real* t = N; N = N2; N2 = t;
        }
        __syncthreads();
    }
    if (threadIdx.x < d) oa[(r * P + c) * d + threadIdx.x] = z[threadIdx.x];
    if (c > 0) store_tile<DP>(oN + (r * P + c) * dd, N, d);
}
// emit: chunk c walks the maps of the chunks before it, then the plain recursion over its own blocks
template <int DP, int TR>
__global__ void __launch_bounds__(NTHR) bigpar_solve_emit_kernel(long Bl, long Br, long n, int d, long P, long L,
                                                                const real* __restrict__ ldiag, const real* __restrict__ lsub,
                                                                const real* __restrict__ rhs, const real* __restrict__ wN,
                                                                const real* __restrict__ wa, real* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long r = blockIdx.x / P, c = blockIdx.x % P, s = r % Bl;
    const long p0 = c * L, dd = (long)d * d;
    long p1 = p0 + L;
    if (p1 > n) p1 = n;
    real *Lt = sm.tile(0), *Linv = sm.tile(1), *C = sm.tile(2);
    real *z = sm.vec(0), *x = sm.vec(1);
    bool bad = false;
    if (threadIdx.x < 64) z[threadIdx.x] = 0;
    __syncthreads();
    for (long j = 0; j < c; ++j) {
        load_vec_lds<DP>(x, wa + (r * P + j) * d, nullptr, d);
        if (j > 0) load_tile<DP>(C, wN + (r * P + j) * dd, nullptr, d, false, false);
        __syncthreads();
        if (j > 0) matvec<DP, 0>(C, z, x, 1.f, 1.f, sm.scratch());
        if (threadIdx.x < 64) z[threadIdx.x] = x[threadIdx.x];
        __syncthreads();
    }
    for (long p = p0; p < p1; ++p) {
        const long k = TR ? n - 1 - p : p;
        load_tile<DP>(Lt, ldiag + (s * n + k) * dd, nullptr, d, true, true);
        load_vec_lds<DP>(x, rhs + (r * n + k) * d, nullptr, d);
        const bool coupled = p > 0;
        if (coupled) load_tile<DP>(C, lsub + (s * (n - 1) + (TR ? k : k - 1)) * dd, nullptr, d, false, false);
        __syncthreads();
        if (coupled) matvec<DP, TR>(C, z, x, -1.f, 1.f, sm.scratch());
        (void)factor_invert<DP, false>(Lt, Linv, bad, sm.scratch());
        matvec<DP, TR>(Linv, x, z, 1.f, 0.f, sm.scratch());
        if (threadIdx.x < d) out[(r * n + k) * d + threadIdx.x] = z[threadIdx.x];
        __syncthreads();
    }
}

// ---- marginal means / sample propagation (state_space_model.py:232-251):  x_k = A_{k-1} x_{k-1} + offs_k,  x_0 = offs_0 --------------
template <int DP, int NTILES> struct SmemLite {
    static constexpr int FLOATS = NTILES * Geo<DP>::TILE + 4 * 64 + 256;
    static constexpr int BYTES = FLOATS * (int)sizeof(real);
};
template <int DP>
__global__ void __launch_bounds__(NTHR) bigpar_means_compose_kernel(long Bl, long Br, long n, int d, long P, long L,
                                                                   const real* __restrict__ A, const real* __restrict__ offs,
                                                                   real* __restrict__ oN, real* __restrict__ oa) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    real* base = reinterpret_cast<real*>(smem_raw);
    real *At = base, *N = base + Geo<DP>::TILE, *N2 = N + Geo<DP>::TILE;
    real *x = N2 + Geo<DP>::TILE, *xn = x + 64, *scratch = xn + 192;
    const long r = blockIdx.x / (P - 1), c = blockIdx.x % (P - 1), s = r % Bl;
    const long k0 = c * L, dd = (long)d * d;
    long k1 = k0 + L;
    if (k1 > n) k1 = n;
    if (threadIdx.x < 64) x[threadIdx.x] = 0;
    if (c > 0) identity_tile<DP>(N);
    __syncthreads();
    for (long k = k0; k < k1; ++k) {
        load_vec_lds<DP>(xn, offs + (r * n + k) * d, nullptr, d);
        if (k > 0) load_tile<DP>(At, A + (s * (n - 1) + k - 1) * dd, nullptr, d, false, false);
        __syncthreads();
        if (k > 0) matvec<DP, 0>(At, x, xn, 1.f, 1.f, scratch);
        if (c > 0) {
            gemm<DP, 0, 0, 0, K_FULL, O_FULL>(At, N, N2, 1.f);
            real* t = N; N = N2; N2 = t;
        }
        if (threadIdx.x < 64) x[threadIdx.x] = xn[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x < d) oa[(r * P + c) * d + threadIdx.x] = x[threadIdx.x];
    if (c > 0) store_tile<DP>(oN + (r * P + c) * dd, N, d);
}
template <int DP>
__global__ void __launch_bounds__(NTHR) bigpar_means_emit_kernel(long Bl, long Br, long n, int d, long P, long L,
                                                                const real* __restrict__ A, const real* __restrict__ offs,
                                                                const real* __restrict__ wN, const real* __restrict__ wa,
                                                                real* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    real* base = reinterpret_cast<real*>(smem_raw);
    real* At = base;
    real *x = base + Geo<DP>::TILE, *xn = x + 64, *scratch = xn + 192;
    const long r = blockIdx.x / P, c = blockIdx.x % P, s = r % Bl;
    const long k0 = c * L, dd = (long)d * d;
    long k1 = k0 + L;
    if (k1 > n) k1 = n;
    if (threadIdx.x < 64) x[threadIdx.x] = 0;
    __syncthreads();
    for (long j = 0; j < c; ++j) {
        load_vec_lds<DP>(xn, wa + (r * P + j) * d, nullptr, d);
        if (j > 0) load_tile<DP>(At, wN + (r * P + j) * dd, nullptr, d, false, false);
        __syncthreads();
        if (j > 0) matvec<DP, 0>(At, x, xn, 1.f, 1.f, scratch);
        if (threadIdx.x < 64) x[threadIdx.x] = xn[threadIdx.x];
        __syncthreads();
    }
    for (long k = k0; k < k1; ++k) {
        load_vec_lds<DP>(xn, offs + (r * n + k) * d, nullptr, d);
        if (k > 0) load_tile<DP>(At, A + (s * (n - 1) + k - 1) * dd, nullptr, d, false, false);
        __syncthreads();
        if (k > 0) matvec<DP, 0>(At, x, xn, 1.f, 1.f, scratch);
        if (threadIdx.x < d) out[(r * n + k) * d + threadIdx.x] = xn[threadIdx.x];
        if (threadIdx.x < 64) x[threadIdx.x] = xn[threadIdx.x];
        __syncthreads();
    }
}

// ---- block_diagonal_of_inverse (block Takahashi, block_tri_diag.py:318-337) ----------------------------------------------------------------
// positions p = n-1-k:  Sigma_p = C_p + G_p^T Sigma_{p-1} G_p,  C = L^-T L^-1,  G = W L^-1 (W = lsub[k]) - a congruence recursion.
// compose: chunk c (< P-1) leaves  Sigma_end = Nc + Mc^T Sigma_start Mc  (Mc = G_{p0} ... G_{p1-1}; chunk 0 starts uncoupled: Nc only).
template <int DP>
__global__ void __launch_bounds__(NTHR) bigpar_tak_compose_kernel(long B, long n, int d, long P, long L, const real* __restrict__ ldiag,
                                                                 const real* __restrict__ lsub, real* __restrict__ oM,
                                                                 real* __restrict__ oN) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long s = blockIdx.x / (P - 1), c = blockIdx.x % (P - 1);
    const long p0 = c * L, dd = (long)d * d;
    long p1 = p0 + L;
    if (p1 > n) p1 = n;
    real *G = sm.tile(0), *Linv = sm.tile(1), *W = sm.tile(2), *Cn = sm.tile(3), *Nc = sm.tile(4), *M = sm.tile(5), *M2 = sm.tile(6);
    bool bad = false;
    zero_tile<DP>(Nc);
    if (c > 0) identity_tile<DP>(M);
    __syncthreads();
    for (long p = p0; p < p1; ++p) {
        const long k = n - 1 - p;
        load_tile<DP>(G, ldiag + (s * n + k) * dd, nullptr, d, true, true);            // L_k, replaced by G_k below
        const bool coupled = p > 0;
        if (coupled) load_tile<DP>(W, lsub + (s * (n - 1) + k) * dd, nullptr, d, false, false);
        __syncthreads();
        (void)factor_invert<DP, false>(G, Linv, bad, sm.scratch());
        gemm<DP, 1, 0, 0, K_A_UPPER, O_FULL>(Linv, Linv, Cn, 1.f);                      // L^-T L^-1
        if (coupled) {
            gemm<DP, 0, 0, 0, K_B_LOWER, O_FULL>(W, Linv, G, 1.f);                      // G = W L^-1
            __syncthreads();
            gemm<DP, 0, 0, 0, K_FULL, O_FULL>(Nc, G, W, 1.f);                           // Nc G   (W is free)
            if (c > 0) gemm<DP, 0, 0, 0, K_FULL, O_FULL>(M, G, M2, 1.f);                // M <- M G
            __syncthreads();
            gemm<DP, 1, 0, 1, K_FULL, O_FULL>(G, W, Cn, 1.f);                           // + G^T Nc G
            if (c > 0) { real* t = M; M = M2; M2 = t; }
        }
        __syncthreads();
        { real* t = Nc; Nc = Cn; Cn = t; }
    }
    store_tile<DP>(oN + (s * P + c) * dd, Nc, d);
    if (c > 0) store_tile<DP>(oM + (s * P + c) * dd, M, d);
}
// start[s, c] = Sigma at the end of chunk c - 1 (c = 1 ... P-1)
template <int DP>
__global__ void __launch_bounds__(NTHR) bigpar_tak_boundary_kernel(long B, int d, long P, const real* __restrict__ wM,
                                                                  const real* __restrict__ wN, real* __restrict__ start) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    const long s = blockIdx.x, dd = (long)d * d;
    real *S = sm.tile(0), *Mt = sm.tile(1), *T1 = sm.tile(2), *Nt = sm.tile(3);
    load_tile<DP>(S, wN + (s * P) * dd, nullptr, d, false, false);
    __syncthreads();
    store_tile<DP>(start + (s * P + 1) * dd, S, d);
    for (long c = 1; c + 1 < P; ++c) {
        load_tile<DP>(Mt, wM + (s * P + c) * dd, nullptr, d, false, false);
        load_tile<DP>(Nt, wN + (s * P + c) * dd, nullptr, d, false, false);
        __syncthreads();
        gemm<DP, 0, 0, 0, K_FULL, O_FULL>(S, Mt, T1, 1.f);
        __syncthreads();
        gemm<DP, 1, 0, 0, K_FULL, O_FULL>(Mt, T1, S, 1.f);                              // M^T Sigma M
        __syncthreads();
        add_tile<DP>(S, Nt);
        __syncthreads();
        store_tile<DP>(start + (s * P + c + 1) * dd, S, d);
    }
}
template <int DP>
__global__ void __launch_bounds__(NTHR) bigpar_tak_emit_kernel(long B, long n, int d, long P, long L, const real* __restrict__ ldiag,
                                                              const real* __restrict__ lsub, const real* __restrict__ start,
                                                              real* __restrict__ odiag, real* __restrict__ osub) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    constexpr int LD = Geo<DP>::LD;
    const long s = blockIdx.x / P, c = blockIdx.x % P;
    const long p0 = c * L, dd = (long)d * d;
    long p1 = p0 + L;
    if (p1 > n) p1 = n;
    real *G = sm.tile(0), *Linv = sm.tile(1), *W = sm.tile(2), *Sig = sm.tile(3), *Out = sm.tile(4);
    bool bad = false;
    if (c > 0) load_tile<DP>(Sig, start + (s * P + c) * dd, nullptr, d, false, false);
    for (long p = p0; p < p1; ++p) {
        const long k = n - 1 - p;
        load_tile<DP>(G, ldiag + (s * n + k) * dd, nullptr, d, true, true);
        const bool coupled = p > 0;
        if (coupled) load_tile<DP>(W, lsub + (s * (n - 1) + k) * dd, nullptr, d, false, false);
        __syncthreads();
        (void)factor_invert<DP, false>(G, Linv, bad, sm.scratch());
        gemm<DP, 1, 0, 0, K_A_UPPER, O_FULL>(Linv, Linv, Out, 1.f);
        if (coupled) {
            gemm<DP, 0, 0, 0, K_B_LOWER, O_FULL>(W, Linv, G, 1.f);
            __syncthreads();
            gemm<DP, 0, 0, 0, K_FULL, O_FULL>(Sig, G, W, 1.f);                          // Sigma_{k+1} G
            __syncthreads();
            if (osub)
                for (int e = threadIdx.x; e < d * d; e += NTHR) osub[(s * (n - 1) + k) * dd + e] = -W[(e / d) * LD + (e % d)];
            gemm<DP, 1, 0, 1, K_FULL, O_FULL>(G, W, Out, 1.f);
        }
        __syncthreads();
        store_tile<DP>(odiag + (s * n + k) * dd, Out, d);
        { real* t = Sig; Sig = Out; Out = t; }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------
// chunks per series: one round of workgroups over the 256 CUs (as many per CU as the LDS carve allows), chunks of at least 8 blocks;
// fewer than 4 chunks: the one-workgroup-per-series kernels
inline void bigpar_partition(long B, long n, int d, long& P, long& L) {
    static const long forced = [] { const char* e = mf_knob("MF_BIGPAR_CHUNKS"); return e ? std::atol(e) : 0L; }();
    const int dp = d <= 16 ? 16 : d <= 32 ? 32 : d <= 48 ? 48 : 64;
    const long lds = (7L * dp * (dp + 4) + 4000) * (long)sizeof(real);
    long per_cu = (160L * 1024) / lds;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    long want = forced > 0 ? forced : (256 * per_cu) / (B > 0 ? B : 1);
    if (want > n / 8) want = n / 8;
    if (want < 1) want = 1;
    L = cdivl(n, want);
    P = cdivl(n, L);
    if (P < 4) { P = 1; L = n; }
}
inline size_t bigpar_ws(long B, long n, int d, bool chain) {
    long P, L;
    bigpar_partition(B, n, d, P, L);
    if (P == 1) return 0;
    const size_t blk = size_t(B) * P * d * d * sizeof(real);
    return 4 * blk + (chain ? blk + align_up_big(size_t(B) * P * d * sizeof(real)) : 0);
}
// (the query does not know which engine takes the call: the larger of the tile engine's and the wave kernels' partition)
inline size_t bigpar_ws_any(long B, long n, int d, bool chain) {
    const size_t e = bigpar_ws(B, n, d, chain), w = wave_udl_ws(B, n, d, (int)sizeof(real));
    return e > w ? e : w;
}
struct BigParWs { real *Dv, *GU, *F, *piv, *N, *a; };
inline BigParWs bigpar_carve(void* ws, long B, long P, int d) {
    const size_t blk = size_t(B) * P * d * d;
    real* p = static_cast<real*>(ws);
    return BigParWs{p, p + blk, p + 2 * blk, p + 3 * blk, p + 4 * blk, p + 5 * blk};
}

template <int DP, bool REV>
inline bool bigpar_pivots(long B, long n, int d, long P, long L, const real* diag, const real* sub, const BigParWs& w, int* info, hipStream_t st) {
    static const bool ok = big_attr(&bigpar_chol_up_kernel<DP, REV>, Smem<DP>::BYTES) && big_attr(&bigpar_chol_boundary_kernel<DP>, Smem<DP>::BYTES);
    if (!ok) return false;
    // 32 < d <= 64: the up-sweep on the panel kernels (mf_panel.hpp, panel_red_kernel in operator mode: 1.64 -> 0.5 ms at config 5's shape)
    static const bool poff = std::getenv("MF_PANEL_UP_OFF") != nullptr;       // (A/B switch)
    static const bool boff = std::getenv("MF_PANEL_BOUNDARY_OFF") != nullptr;
    const int prc = poff ? -101 : panel_chol_up(B, n, d, P, L, diag, sub, w.Dv, w.GU, w.F, boff ? nullptr : w.piv, REV ? 1 : 0, info, st);
    if (prc != 0 && prc != -101) return false;
    if (prc == 0 && !boff) return true;              // (up-sweep and chunk ends both on the panel kernels)
    if (prc == -101)
    hipLaunchKernelGGL((bigpar_chol_up_kernel<DP, REV>), dim3((unsigned)(B * (P - 1))), dim3(NTHR), Smem<DP>::BYTES, st, B, n, d, P, L,
                       diag, sub, w.Dv, w.GU, w.F, info);
    hipLaunchKernelGGL((bigpar_chol_boundary_kernel<DP>), dim3((unsigned)B), dim3(NTHR), Smem<DP>::BYTES, st, B, d, P,
                       static_cast<const real*>(w.Dv), static_cast<const real*>(w.GU), static_cast<const real*>(w.F), w.piv, info);
    return true;
}

inline int op_cholesky_par(long B, long n, int d, const real* diag, const real* sub, real* ldiag, real* lsub, void* ws, size_t ws_bytes,
                           int* info, hipStream_t st) {
    if (!wave_off()) {   // 16 <= d <= 32, many series: one wavefront per series on register tiles (mf_wave_ops.hpp)
        const int rc = wave_btd_cholesky<real>(B, n, d, diag, sub, ldiag, lsub, ws, ws_bytes, info, st);
        if (rc != -101) return rc;
    }
    long P, L;
    bigpar_partition(B, n, d, P, L);
    if (P == 1 || !sub || !ws || ws_bytes < bigpar_ws(B, n, d, false)) return op_cholesky(B, n, d, diag, sub, ldiag, lsub, info, st);
    const BigParWs w = bigpar_carve(ws, B, P, d);
#define MF_C(DP)                                                                                                        \
    { static const bool ok = big_attr(&bigpar_chol_emit_kernel<DP>, Smem<DP>::BYTES);                                    \
      if (!ok || !bigpar_pivots<DP, false>(B, n, d, P, L, diag, sub, w, info, st)) return -1000;                         \
      static const bool eoff = std::getenv("MF_PANEL_EMIT_OFF") != nullptr;                                             \
      const int erc = eoff ? -101 : panel_chol_emit(B, n, d, P, L, diag, sub, static_cast<const real*>(w.piv), ldiag, lsub, info, st); \
      if (erc != 0 && erc != -101) return -1000;                                                                        \
      if (erc == -101)                                                                                                  \
      hipLaunchKernelGGL((bigpar_chol_emit_kernel<DP>), dim3((unsigned)(B * P)), dim3(NTHR), Smem<DP>::BYTES, st, B, n, d, P, L, \
                         diag, sub, static_cast<const real*>(w.piv), ldiag, lsub, info); }
    MF_BIGOP_DISPATCH(MF_C)
#undef MF_C
    return big_ok();
}

inline int op_udl_par(long B, long n, int d, const real* diag, const real* sub, real* ut, real* chol_d, const real* eta, real* m_post,
                      real* chol_dinv, void* ws, size_t ws_bytes, int* info, hipStream_t st) {
    if (!wave_off()) {
        const int rc = wave_btd_udl<real>(B, n, d, diag, sub, ut, chol_d, eta, m_post, chol_dinv, ws, ws_bytes, info, st);
        if (rc != -101) return rc;
    }
    long P, L;
    bigpar_partition(B, n, d, P, L);
    if (P == 1 || !sub || !ws || ws_bytes < bigpar_ws(B, n, d, eta != nullptr))
        return op_udl(B, n, d, diag, sub, ut, chol_d, eta, m_post, chol_dinv, info, st);
    const BigParWs w = bigpar_carve(ws, B, P, d);
#define MF_C(DP)                                                                                                        \
    { static const bool ok = big_attr(&bigpar_udl_emit_kernel<DP>, Smem<DP>::BYTES) &&                                   \
                             big_attr(&bigpar_udl_means_kernel<DP>, SmemVec<DP>::BYTES);                                 \
      if (!ok || !bigpar_pivots<DP, true>(B, n, d, P, L, diag, sub, w, info, st)) return -1000;                          \
      hipLaunchKernelGGL((bigpar_udl_emit_kernel<DP>), dim3((unsigned)(B * P)), dim3(NTHR), Smem<DP>::BYTES, st, B, n, d, P, L, \
                         diag, sub, static_cast<const real*>(w.piv), ut, chol_d, eta, chol_dinv, w.N, w.a, info);        \
      if (eta)                                                                                                          \
          hipLaunchKernelGGL((bigpar_udl_means_kernel<DP>), dim3((unsigned)(B * P)), dim3(NTHR), SmemVec<DP>::BYTES, st, B, n, d, \
                             P, L, static_cast<const real*>(ut), static_cast<const real*>(chol_dinv), eta,               \
                             static_cast<const real*>(w.N), static_cast<const real*>(w.a), m_post); }
    MF_BIGOP_DISPATCH(MF_C)
#undef MF_C
    return big_ok();
}

// solve: N [Br, P, d, d] and a [Br, P, d]
inline size_t bigpar_solve_ws(long Bl, long Br, long n, int d) {
    long P, L;
    bigpar_partition(Br, n, d, P, L);
    if (P == 1) return 0;
    return size_t(Br) * P * d * d * sizeof(real) + align_up_big(size_t(Br) * P * d * sizeof(real));
}
inline int op_solve_par(long Bl, long Br, long n, int d, const real* ldiag, const real* lsub, const real* rhs, real* out, int transpose,
                        void* ws, size_t ws_bytes, hipStream_t st) {
    {   // 16 <= d <= 32: the time axis serially inside a wavefront, the batch over the chip (mf_wave_ops.hpp)
        if (!wave_off() && Br > 0 && n > 0) {
            const int rc = wave_btd_solve(Bl, Br, n, d, ldiag, lsub, rhs, out, transpose, ws, ws_bytes, st);
            if (rc != -101) return rc;
        }
    }
    long P, L;
    bigpar_partition(Br, n, d, P, L);
    if (P == 1 || !lsub || !ws || ws_bytes < bigpar_solve_ws(Bl, Br, n, d)) return op_solve(Bl, Br, n, d, ldiag, lsub, rhs, out, transpose, st);
    real* wN = static_cast<real*>(ws);
    real* wa = wN + size_t(Br) * P * d * d;
#define MF_S(DP, TR)                                                                                                    \
    { static const bool ok = big_attr(&bigpar_solve_compose_kernel<DP, TR>, Smem<DP>::BYTES) &&                          \
                             big_attr(&bigpar_solve_emit_kernel<DP, TR>, Smem<DP>::BYTES);                               \
      if (!ok) return -1000;                                                                                            \
      hipLaunchKernelGGL((bigpar_solve_compose_kernel<DP, TR>), dim3((unsigned)(Br * (P - 1))), dim3(NTHR), Smem<DP>::BYTES, st, Bl, Br, \
                         n, d, P, L, ldiag, lsub, rhs, wN, wa);                                                          \
      hipLaunchKernelGGL((bigpar_solve_emit_kernel<DP, TR>), dim3((unsigned)(Br * P)), dim3(NTHR), Smem<DP>::BYTES, st, Bl, Br, n, d, \
                         P, L, ldiag, lsub, rhs, static_cast<const real*>(wN), static_cast<const real*>(wa), out); }
#define MF_C(DP) if (transpose) MF_S(DP, 1) else MF_S(DP, 0)
    MF_BIGOP_DISPATCH(MF_C)
#undef MF_C
#undef MF_S
    return big_ok();
}

// marginal means: the workspace of the solve (N [Br, P, d, d], a [Br, P, d])
inline int op_means_par(long Bl, long Br, long n, int d, const real* A, const real* offs, real* out, void* ws, size_t ws_bytes,
                        hipStream_t st) {
    if (!wave_off()) {   // 16 <= d <= 32: the chunk maps on the register tiles, then the readlane walk per chunk
        long Pw = 1, Lw = 0;
        const real* m_in = nullptr;
        if (wave_means_boundaries(Bl, Br, n, d, A, offs, ws, ws_bytes, &Pw, &Lw, &m_in, st) != 0) return -1000;
        if (Pw > 1) return op_means(Bl, Br, n, d, A, offs, out, st, Pw, Lw, m_in);
    }
    long P, L;
    bigpar_partition(Br, n, d, P, L);
    if (P == 1 || !ws || ws_bytes < bigpar_solve_ws(Bl, Br, n, d)) return op_means(Bl, Br, n, d, A, offs, out, st);
    real* wN = static_cast<real*>(ws);
    real* wa = wN + size_t(Br) * P * d * d;
#define MF_C(DP)                                                                                                        \
    { static const bool ok = big_attr(&bigpar_means_compose_kernel<DP>, SmemLite<DP, 3>::BYTES) &&                       \
                             big_attr(&bigpar_means_emit_kernel<DP>, SmemLite<DP, 1>::BYTES);                            \
      if (!ok) return -1000;                                                                                            \
      hipLaunchKernelGGL((bigpar_means_compose_kernel<DP>), dim3((unsigned)(Br * (P - 1))), dim3(NTHR), (SmemLite<DP, 3>::BYTES), st, Bl, \
                         Br, n, d, P, L, A, offs, wN, wa);                                                               \
      hipLaunchKernelGGL((bigpar_means_emit_kernel<DP>), dim3((unsigned)(Br * P)), dim3(NTHR), (SmemLite<DP, 1>::BYTES), st, Bl, Br, n, \
                         d, P, L, A, offs, static_cast<const real*>(wN), static_cast<const real*>(wa), out); }
    MF_BIGOP_DISPATCH(MF_C)
#undef MF_C
    return big_ok();
}

// block_diagonal_of_inverse: M, N, start [B, P, d, d]
inline size_t bigpar_tak_ws_engine(long B, long n, int d) {
    long P, L;
    bigpar_partition(B, n, d, P, L);
    return P == 1 ? 0 : 3 * size_t(B) * P * d * d * sizeof(real);
}
// (the query does not know which engine takes the call: the larger of the tile engine's and the wave kernels' partition)
inline size_t bigpar_tak_ws(long B, long n, int d) {
    const size_t e = bigpar_tak_ws_engine(B, n, d), w = wave_udl_ws(B, n, d, (int)sizeof(real));
    return e > w ? e : w;
}
inline int op_diag_of_inverse_par(long B, long n, int d, const real* ldiag, const real* lsub, real* odiag, real* osub, void* ws,
                                  size_t ws_bytes, hipStream_t st) {
    if (!wave_off()) {
        const int rc = wave_btd_diag_of_inverse<real>(B, n, d, ldiag, lsub, odiag, osub, ws, ws_bytes, st);
        if (rc != -101) return rc;
    }
    long P, L;
    bigpar_partition(B, n, d, P, L);
    if (P == 1 || !lsub || !ws || ws_bytes < bigpar_tak_ws_engine(B, n, d)) return op_diag_of_inverse(B, n, d, ldiag, lsub, odiag, osub, st);
    const size_t blk = size_t(B) * P * d * d;
    real *wM = static_cast<real*>(ws), *wN = wM + blk, *start = wN + blk;
#define MF_C(DP)                                                                                                        \
    { static const bool ok = big_attr(&bigpar_tak_compose_kernel<DP>, Smem<DP>::BYTES) &&                                \
                             big_attr(&bigpar_tak_boundary_kernel<DP>, Smem<DP>::BYTES) &&                               \
                             big_attr(&bigpar_tak_emit_kernel<DP>, Smem<DP>::BYTES);                                     \
      if (!ok) return -1000;                                                                                            \
      hipLaunchKernelGGL((bigpar_tak_compose_kernel<DP>), dim3((unsigned)(B * (P - 1))), dim3(NTHR), Smem<DP>::BYTES, st, B, n, d, P, L, \
                         ldiag, lsub, wM, wN);                                                                           \
      hipLaunchKernelGGL((bigpar_tak_boundary_kernel<DP>), dim3((unsigned)B), dim3(NTHR), Smem<DP>::BYTES, st, B, d, P,  \
                         static_cast<const real*>(wM), static_cast<const real*>(wN), start);                             \
      hipLaunchKernelGGL((bigpar_tak_emit_kernel<DP>), dim3((unsigned)(B * P)), dim3(NTHR), Smem<DP>::BYTES, st, B, n, d, P, L, ldiag, \
                         lsub, static_cast<const real*>(start), odiag, osub); }
    MF_BIGOP_DISPATCH(MF_C)
#undef MF_C
    return big_ok();
}

}  // namespace MF_BIG_NS
}  // namespace mf
