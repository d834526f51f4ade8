// Gradient of the Kalman log-likelihood for LARGE state dimension (10 <= d <= 64 fp32 / 32 fp64): the local step of Fisher's
// identity, grad log p(y) = E_{x|y}[grad log p(x, y)], on the LDS-tile / MFMA engine - the large-d form of kf_grad_kernel
// (mf_kernels.hpp; reference: TensorFlow reverse mode through kalman_filter.py:184-255, pinned by
// tests/integration/models/test_variational.py:123-132 there).  One workgroup per (series, time point k), no dependence between
// them: the smoothed moments m_k, S_k, X_{k-1} = Cov(x_k, x_{k-1}) come from the posterior chain's forward recursion
// (bigop_cov_chunk_kernel), and with them
//
//   observation k:   r = y - H m,  HS = H S:      dH = R^-1 (r m^T - HS),  dy = -R^-1 r,  Omega = r r^T + HS H^T
//   transition k-1:  e = m_k - A m_{k-1} - b,     E[e x^T] = X - A S_{k-1} + e m_{k-1}^T,
//                    Psi = S_k - A X^T - X A^T + A S_{k-1} A^T + e e^T,
//                    dA = Q^-1 E[e x^T],  db = Q^-1 e,  dC = tril(C^-T (C^-1 Psi C^-T - I)),        Q = C C^T
//   k = 0:           the same with A absent, b = mu0, C = cholP0.
//
// Every output is multiplied by the incoming weight of its series.  Eleven d x d products and one triangular inverse per point.
// Included once per scalar type, after mf_bigops_impl.hpp.
namespace mf {
namespace MF_BIG_NS {

struct BigGradArgs {
    long B, Tn;
    int d, m;
    const real *mu0, *cholP0, *A, *b, *cholQ, *H, *y, *Rinv;
    int rinv_per_step;
    const real *mean, *cov, *cross, *w;
    real *g_mu0, *g_cholP0, *g_A, *g_b, *g_cholQ, *g_H, *g_y, *g_om;
};

// tile <- rows x cols block of g (row stride cols), zero padded
template <int DP> __device__ __forceinline__ void load_rect(real* __restrict__ tile, const real* __restrict__ g, int rows, int cols) {
    constexpr int LD = Geo<DP>::LD;
    for (int e = threadIdx.x; e < DP * DP; e += NTHR) {
        const int row = e / DP, col = e % DP;
        tile[row * LD + col] = (row < rows && col < cols) ? g[row * cols + col] : real(0);
    }
}

template <int DP>
__global__ void __launch_bounds__(NTHR) biggrad_local_kernel(BigGradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem<DP> sm{reinterpret_cast<real*>(smem_raw)};
    constexpr int LD = Geo<DP>::LD;
    const long s = blockIdx.x / a.Tn, k = blockIdx.x % a.Tn;
    const int d = a.d, m = a.m;
    const long dd = (long)d * d, nt = a.Tn - 1;
    real *T0 = sm.tile(0), *T1 = sm.tile(1), *T2 = sm.tile(2), *T3 = sm.tile(3), *T4 = sm.tile(4), *T5 = sm.tile(5), *T6 = sm.tile(6);
    real *mn = sm.vec(0), *mp = sm.vec(1), *r = sm.vec(2), *e = sm.vec(3), *t = sm.vec(4), *gv = sm.vec(5);
    const real w = a.w[s];
    bool bad = false;

    load_tile<DP>(T0, a.cov + (s * a.Tn + k) * dd, nullptr, d, false, false);                    // S_k
    load_vec_lds<DP>(mn, a.mean + (s * a.Tn + k) * d, nullptr, d);
    // ---- observation k ----------------------------------------------------------------------------------------------------------
    if (a.H) {
        load_rect<DP>(T1, a.H + (s * a.Tn + k) * m * d, m, d);
        load_rect<DP>(T2, a.rinv_per_step ? a.Rinv + (s * a.Tn + k) * m * m : a.Rinv, m, m);
        if (threadIdx.x < DP) r[threadIdx.x] = threadIdx.x < m ? a.y[(s * a.Tn + k) * m + threadIdx.x] : real(0);
        __syncthreads();
        gemm<DP, 0, 0, 0, K_FULL, O_FULL>(T1, T0, T3, 1.f);                                       // HS
        matvec<DP, 0>(T1, mn, r, -1.f, 1.f, sm.scratch());                                        // r = y - H m
        for (int i = threadIdx.x; i < DP * DP; i += NTHR) {
            const int row = i / DP, col = i % DP;
            T4[row * LD + col] = r[row] * mn[col] - T3[row * LD + col];                           // r m^T - HS
        }
        gemm<DP, 0, 1, 0, K_FULL, O_FULL>(T3, T1, T6, 1.f);                                       // HS H^T
        __syncthreads();
        gemm<DP, 0, 0, 0, K_FULL, O_FULL>(T2, T4, T5, 1.f);                                       // R^-1 (r m^T - HS)
        matvec<DP, 0>(T2, r, gv, 1.f, 0.f, sm.scratch());                                         // R^-1 r
        real* gH = a.g_H + (s * a.Tn + k) * m * d;
        for (int i = threadIdx.x; i < m * d; i += NTHR) gH[i] = w * T5[(i / d) * LD + (i % d)];
        real* gO = a.g_om + (s * a.Tn + k) * m * m;
        for (int i = threadIdx.x; i < m * m; i += NTHR) gO[i] = w * (T6[(i / m) * LD + (i % m)] + r[i / m] * r[i % m]);
        if (threadIdx.x < m) a.g_y[(s * a.Tn + k) * m + threadIdx.x] = -w * gv[threadIdx.x];
        __syncthreads();
    }
    // ---- transition k-1 (k = 0: the prior) -----------------------------------------------------------------------------------------
    const bool tr = k > 0;
    const real* cq = tr ? a.cholQ + (s * nt + k - 1) * dd : a.cholP0 + s * dd;
    const real* off = tr ? a.b + (s * nt + k - 1) * d : a.mu0 + s * d;
    load_tile<DP>(T4, cq, nullptr, d, true, true);
    if (threadIdx.x < DP) e[threadIdx.x] = threadIdx.x < d ? mn[threadIdx.x] - off[threadIdx.x] : real(0);
    if (tr) {
        load_tile<DP>(T1, a.A + (s * nt + k - 1) * dd, nullptr, d, false, false);
        load_tile<DP>(T2, a.cov + (s * a.Tn + k - 1) * dd, nullptr, d, false, false);            // S_{k-1}
        load_tile<DP>(T3, a.cross + (s * nt + k - 1) * dd, nullptr, d, false, false);            // X = Cov(x_k, x_{k-1})
        load_vec_lds<DP>(mp, a.mean + (s * a.Tn + k - 1) * d, nullptr, d);
    }
    __syncthreads();
    (void)factor_invert<DP, false>(T4, T5, bad, sm.scratch());                                    // T5 = C^-1
    if (tr) {
        gemm<DP, 0, 0, 0, K_FULL, O_FULL>(T1, T2, T4, 1.f);                                       // A S_{k-1}
        gemm<DP, 0, 1, 0, K_FULL, O_FULL>(T1, T3, T6, 1.f);                                       // A X^T
        matvec<DP, 0>(T1, mp, e, -1.f, 1.f, sm.scratch());                                        // e = m_k - b - A m_{k-1}
    } else {
        __syncthreads();
    }
    for (int i = threadIdx.x; i < DP * DP; i += NTHR) {
        const int row = i / DP, col = i % DP;
        real psi = T0[row * LD + col] + e[row] * e[col];
        if (tr) {
            psi -= T6[row * LD + col] + T6[col * LD + row];
            T3[row * LD + col] += e[row] * mp[col] - T4[row * LD + col];                          // E[e x^T]
        }
        T0[row * LD + col] = psi;
    }
    __syncthreads();
    if (tr) {
        gemm<DP, 0, 1, 1, K_FULL, O_FULL>(T4, T1, T0, 1.f);                                       // Psi += A S A^T
        gemm<DP, 0, 0, 0, K_A_LOWER, O_FULL>(T5, T3, T2, 1.f);                                    // C^-1 E[e x^T]
        __syncthreads();
        gemm<DP, 1, 0, 0, K_A_UPPER, O_FULL>(T5, T2, T1, 1.f);                                    // Q^-1 E[e x^T]
    }
    gemm<DP, 0, 0, 0, K_A_LOWER, O_FULL>(T5, T0, T6, 1.f);                                        // C^-1 Psi
    matvec<DP, 0>(T5, e, t, 1.f, 0.f, sm.scratch());
    matvec<DP, 1>(T5, t, gv, 1.f, 0.f, sm.scratch());                                             // Q^-1 e
    gemm<DP, 0, 1, 0, K_B_UPPER, O_FULL>(T6, T5, T3, 1.f);                                        // C^-1 Psi C^-T
    if (tr) {
        real* gA = a.g_A + (s * nt + k - 1) * dd;
        for (int i = threadIdx.x; i < d * d; i += NTHR) gA[i] = w * T1[(i / d) * LD + (i % d)];
    }
    {
        real* gb = tr ? a.g_b + (s * nt + k - 1) * d : a.g_mu0 + s * d;
        if (threadIdx.x < d) gb[threadIdx.x] = w * gv[threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x < DP) T3[threadIdx.x * LD + threadIdx.x] -= real(1);
    __syncthreads();
    gemm<DP, 1, 0, 0, K_A_UPPER, O_FULL>(T5, T3, T2, 1.f);                                        // C^-T (C^-1 Psi C^-T - I)
    __syncthreads();
    {
        real* gC = tr ? a.g_cholQ + (s * nt + k - 1) * dd : a.g_cholP0 + s * dd;
        for (int i = threadIdx.x; i < d * d; i += NTHR) {
            const int row = i / d, col = i % d;
            gC[i] = col <= row ? w * T2[row * LD + col] : real(0);
        }
    }
}

inline int op_kf_grad(const BigGradArgs& a, hipStream_t st) {
    const int d = a.d;
    if (!wave_off()) {   // 16 <= d <= 32, up to four outputs: one wavefront per (series, time point) on register tiles (mf_wave_grad.hpp)
        const int rc = wave_kf_grad<real>(a.B, a.Tn, a.d, a.m, a.mu0, a.cholP0, a.A, a.b, a.cholQ, a.H, a.y, a.Rinv, a.rinv_per_step, a.mean,
                                          a.cov, a.cross, a.w, a.g_mu0, a.g_cholP0, a.g_A, a.g_b, a.g_cholQ, a.g_H, a.g_y, a.g_om, st);
        if (rc != -101) return rc;
    }
#define MF_C(DP)                                                                                                        \
    { if (a.H && a.m > DP) return -4;                                                                                   \
      static const bool ok = big_attr(&biggrad_local_kernel<DP>, Smem<DP>::BYTES);                                       \
      if (!ok) return -1000;                                                                                            \
      hipLaunchKernelGGL((biggrad_local_kernel<DP>), dim3((unsigned)(a.B * a.Tn)), dim3(NTHR), Smem<DP>::BYTES, st, a); }
    MF_BIGOP_DISPATCH(MF_C)
#undef MF_C
    return big_ok();
}

}  // namespace MF_BIG_NS
}  // namespace mf
