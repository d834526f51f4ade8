// Reverse-mode gradient of  KL(q1 || q2)  between two state space models with respect to the parameters of q1
// (markovflow/state_space_model.py:528-593 is differentiated by TensorFlow in the reference; pinned by
// tests/integration/models/test_variational.py:123-132).
//
// With  dA = A1 - A2,  db = b1 - b2  and the marginals (m_k, S_k) of q1 the divergence is a sum of local terms,
//   KL = 1/2 tr(P0_2^-1 (S_0 + d0 d0^T)) + 1/2 sum_k [ tr(Q2_k^-1 Q1_k) + tr(Q2_k^-1 dA_k (S_k + m_k m_k^T) dA_k^T)
//        + 2 db_k^T Q2_k^-1 dA_k m_k + db_k^T Q2_k^-1 db_k ] - H(q1) + terms of q2 alone,
// so the gradient is the local partial derivative plus the adjoint of the moment recursion  m_{k+1} = A1 m_k + b1,
// S_{k+1} = A1 S_k A1^T + Q1  - a backward recursion per series for  lam_k (d)  and  M_k = 2 dKL/dS_k (d x d, symmetric):
//   lam_k = dA_k^T Q2_k^-1 eps_k + A1_k^T lam_{k+1},      eps_k = dA_k m_k + db_k,
//   M_k   = dA_k^T Q2_k^-1 dA_k + A1_k^T M_{k+1} A1_k,    lam_{T-1} = 0, M_{T-1} = 0,
//   dKL/db1_k = Q2_k^-1 eps_k + lam_{k+1},
//   dKL/dA1_k = (Q2_k^-1 eps_k + lam_{k+1}) m_k^T + (Q2_k^-1 dA_k + M_{k+1} A1_k) S_k,
//   dKL/dC1_k = tril((Q2_k^-1 + M_{k+1}) C1_k) - diag(1 / C1_k)          (Q1 = C1 C1^T; the last term is the entropy),
//   dKL/dmu0_1 = P0_2^-1 d0 + lam_0,   dKL/dC0_1 = tril((P0_2^-1 + M_0) C0_1) - diag(1 / C0_1).
// The gradient with respect to q2 is local in time (minus the expected complete-data score of q2 under q1's marginals) and
// comes from kf_grad_kernel.  The recursion and the local parts run as three kernels (below: "the adjoint in three kernels");
// `marginals` (state_space_model.py:232-262) has the same adjoint with the incoming gradients as (N_k, n_k).
#pragma once
#include "mf_small.hpp"

namespace mf {

// N = W^T W (symmetric, stored full), n = W^T u  with  W = C2^-1 dA,  u = C2^-1 eps: the adjoint recursion's inputs of a step
template <typename T, int D> MF_DEV void kl_store_adjoint_inputs(const T (&W)[D][D], const T (&u)[D], T* __restrict__ Nout,
                                                                T* __restrict__ nout) {
    MF_UNROLL for (int j = 0; j < D; ++j) {
        T acc = T(0);
        MF_UNROLL for (int i = 0; i < D; ++i) acc += W[i][j] * u[i];
        nout[j] = acc;
    }
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j <= i; ++j) {
            T acc = T(0);
            MF_UNROLL for (int l = 0; l < D; ++l) acc += W[l][i] * W[l][j];
            Nout[i * D + j] = acc;
            Nout[j * D + i] = acc;
        }
}
template <typename T, int D> MF_DEV void kl_zero_adjoint_inputs(T* __restrict__ Nout, T* __restrict__ nout) {
    MF_UNROLL for (int e = 0; e < D * D; ++e) Nout[e] = T(0);
    MF_UNROLL for (int e = 0; e < D; ++e) nout[e] = T(0);
}

// ---- the divergence itself, fused ----------------------------------------------------------------------------------------
// The same local form gives KL(q1 || q2) in ONE forward sweep per series that carries q1's marginal (m_k, S_k) in registers:
//   KL = 1/2 [ |C0_2^-1 C0_1|_F^2 + |C0_2^-1 d0|^2 ] + sum_k 1/2 [ |C2_k^-1 C1_k|_F^2 + tr(W_k S_k W_k^T) + |C2_k^-1 eps_k|^2 ]
//        - T d / 2 + sum log|C2| - sum log|C1|,        W_k = C2_k^-1 dA_k,  eps_k = dA_k m_k + db_k.
// Reads the ten parameter tensors once ((4 d^2 + 2 d) s bytes per step) and writes one scalar per series: nothing of the
// reference's route - marginal covariances of q1, precision of q2, a block-sparse trace, two mean scans, a symmetric product
// (state_space_model.py:569-593) - is materialised.  One lane per series: used when there are enough series to fill the chip.
template <typename T, int D>
__global__ void __launch_bounds__(64) ssm_kl_kernel(long B, long Tn, const T* __restrict__ mu0_1, const T* __restrict__ C0_1,
                                                    const T* __restrict__ A_1, const T* __restrict__ b_1,
                                                    const T* __restrict__ C_1, const T* __restrict__ mu0_2,
                                                    const T* __restrict__ C0_2, const T* __restrict__ A_2,
                                                    const T* __restrict__ b_2, const T* __restrict__ C_2, T* __restrict__ out,
                                                    T* __restrict__ oN, T* __restrict__ on, T* __restrict__ omean,
                                                    T* __restrict__ ocov, T* __restrict__ ocross, int* info) {
    // omean [B,T,D], ocov [B,T,D,D], ocross [B,T-1,D,D] (nullable, all or none): q1's marginals and Cov(x_{k+1}, x_k) = A1_k S_k,
    // which this sweep carries in registers anyway - the backward needs them (mf_ssm_kl_grad, mf_kf_loglik_grad)
    // oN [B,T,D,D], on [B,T,D] (nullable): N_k = W_k^T W_k, n_k = W_k^T C2_k^-1 eps_k - the inputs of the adjoint recursion of the
    // backward (mf_ssm_kl_grad), which are by-products of this sweep
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    LogAcc<T> l1, l2;
    l1.init();
    l2.init();
    bool bad = false;
    T m[D], S[D][D];        // S: lower triangle
    T acc = T(0);
    // squared Frobenius norm of C2i * C1 (lower x lower) and S <- (C1 C1^T) (+ S if `add`)
    auto chol_terms = [&](const T (&C2i)[D][D], const T (&C1)[D][D]) {
        T sum = T(0);
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                T a = T(0);
                MF_UNROLL for (int l = j; l <= i; ++l) a += C2i[i][l] * C1[l][j];
                sum += a * a;
            }
        return sum;
    };
    {
        T C1[D][D], C2[D][D], C2i[D][D], d0[D], u[D];
        load_lower<T, D>(C0_1 + s * D * D, C1);
        load_lower<T, D>(C0_2 + s * D * D, C2);
        tri_inv_lower<T, D>(C2, C2i, l2, bad);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            l1.mul(C1[i][i]);
            bad |= !(C1[i][i] != T(0));
            m[i] = mu0_1[s * D + i];
            d0[i] = m[i] - mu0_2[s * D + i];
        }
        trimul_lower_vec<T, D>(C2i, d0, u);
        acc += chol_terms(C2i, C1) + dot_self<T, D>(u);
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                T a = T(0);
                MF_UNROLL for (int l = 0; l <= j; ++l) a += C1[i][l] * C1[j][l];
                S[i][j] = a;
            }
    }
    for (long k = 0; k + 1 < Tn; ++k) {
        const long tid = s * (Tn - 1) + k;
        if (omean) {
            store_vec<T, D>(omean + (s * Tn + k) * D, m);
            store_sym<T, D>(ocov + (s * Tn + k) * D * D, S);
        }
        T A1[D][D], W[D][D], C1[D][D], C2[D][D], C2i[D][D], eps[D], u[D], mn[D];
        load_mat<T, D, D>(A_1 + tid * D * D, A1);
        load_mat<T, D, D>(A_2 + tid * D * D, W);
        load_lower<T, D>(C_1 + tid * D * D, C1);
        load_lower<T, D>(C_2 + tid * D * D, C2);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            mn[i] = b_1[tid * D + i];
            eps[i] = mn[i] - b_2[tid * D + i];
        }
        __builtin_amdgcn_sched_barrier(0);
        tri_inv_lower<T, D>(C2, C2i, l2, bad);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            l1.mul(C1[i][i]);
            bad |= !(C1[i][i] != T(0));
            MF_UNROLL for (int j = 0; j < D; ++j) W[i][j] = A1[i][j] - W[i][j];          // dA
        }
        MF_UNROLL for (int j = 0; j < D; ++j)
            MF_UNROLL for (int i = 0; i < D; ++i) {
                eps[i] += W[i][j] * m[j];
                mn[i] += A1[i][j] * m[j];
            }
        trimul_lower_vec<T, D>(C2i, eps, u);
        acc += chol_terms(C2i, C1) + dot_self<T, D>(u);
        trimul_lower_inplace<T, D, D>(C2i, W);                                            // W = C2^-1 dA
        if (oN) kl_store_adjoint_inputs<T, D>(W, u, oN + (s * Tn + k) * D * D, on + (s * Tn + k) * D);
        // tr(W S W^T) = sum_ij (W S)_ij W_ij with S symmetric (held in its lower triangle)
        MF_UNROLL for (int i = 0; i < D; ++i) {
            T row[D];
            MF_UNROLL for (int j = 0; j < D; ++j) row[j] = T(0);
            MF_UNROLL for (int l = 0; l < D; ++l)
                MF_UNROLL for (int j = 0; j < D; ++j) row[j] += W[i][l] * ((l >= j) ? S[l][j] : S[j][l]);
            MF_UNROLL for (int j = 0; j < D; ++j) acc += row[j] * W[i][j];
        }
        // S <- A1 S A1^T + C1 C1^T,  m <- A1 m + b1
        T AS[D][D];
        MF_UNROLL for (int i = 0; i < D; ++i) {
            MF_UNROLL for (int j = 0; j < D; ++j) AS[i][j] = T(0);
            MF_UNROLL for (int l = 0; l < D; ++l)
                MF_UNROLL for (int j = 0; j < D; ++j) AS[i][j] += A1[i][l] * ((l >= j) ? S[l][j] : S[j][l]);
        }
        if (ocross) store_mat<T, D, D>(ocross + tid * D * D, AS);
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                T a = T(0);
                MF_UNROLL for (int l = 0; l < D; ++l) a += AS[i][l] * A1[j][l];
                MF_UNROLL for (int l = 0; l <= j; ++l) a += C1[i][l] * C1[j][l];
                S[i][j] = a;
            }
        MF_UNROLL for (int i = 0; i < D; ++i) m[i] = mn[i];
        l1.renorm();
        l2.renorm();
    }
    out[s] = T(0.5) * (acc - T(Tn) * T(D)) + l2.value() - l1.value();
    if (omean) {
        store_vec<T, D>(omean + (s * Tn + Tn - 1) * D, m);
        store_sym<T, D>(ocov + (s * Tn + Tn - 1) * D * D, S);
    }
    if (oN) kl_zero_adjoint_inputs<T, D>(oN + (s * Tn + Tn - 1) * D * D, on + (s * Tn + Tn - 1) * D);
    if (bad && info) raise_info(info);
}

// ---- the divergence with few series: local in time given q1's marginals ---------------------------------------------------
// With too few series for a lane each, q1's marginals (m_k, S_k) come from the scans in time (mf_btd_par.hpp) and every term of
// the sum above is local: lane (s, k) forms the term of transition k (and, for k = 0, of the initial state); a wave per series
// then adds the T partial values.  The marginals are outputs as well - the backward needs exactly these (mf_ssm_kl_grad).
template <typename T, int D>
__global__ void __launch_bounds__(64) ssm_kl_local_kernel(long B, long Tn, const T* __restrict__ mu0_1, const T* __restrict__ C0_1,
                                                          const T* __restrict__ A_1, const T* __restrict__ b_1,
                                                          const T* __restrict__ C_1, const T* __restrict__ mu0_2,
                                                          const T* __restrict__ C0_2, const T* __restrict__ A_2,
                                                          const T* __restrict__ b_2, const T* __restrict__ C_2,
                                                          const T* __restrict__ pm, const T* __restrict__ pS,
                                                          T* __restrict__ part, T* __restrict__ oN, T* __restrict__ on,
                                                          int* info) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * Tn) return;
    const long s = id / Tn, k = id % Tn;
    LogAcc<T> l1, l2;
    l1.init();
    l2.init();
    bool bad = false;
    T acc = T(0);
    auto chol_terms = [&](const T (&C2i)[D][D], const T (&C1)[D][D]) {        // |C2^-1 C1|_F^2 (lower x lower)
        T sum = T(0);
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                T a = T(0);
                MF_UNROLL for (int l = j; l <= i; ++l) a += C2i[i][l] * C1[l][j];
                sum += a * a;
            }
        return sum;
    };
    T m[D];
    load_vec<T, D>(pm + id * D, m);
    if (k == 0) {
        T C1[D][D], C2[D][D], C2i[D][D], d0[D], u[D];
        load_lower<T, D>(C0_1 + s * D * D, C1);
        load_lower<T, D>(C0_2 + s * D * D, C2);
        tri_inv_lower<T, D>(C2, C2i, l2, bad);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            l1.mul(C1[i][i]);
            bad |= !(C1[i][i] != T(0));
            d0[i] = m[i] - mu0_2[s * D + i];
        }
        trimul_lower_vec<T, D>(C2i, d0, u);
        acc += chol_terms(C2i, C1) + dot_self<T, D>(u);
    }
    if (k + 1 < Tn) {
        const long tid = s * (Tn - 1) + k;
        T W[D][D], C1[D][D], C2[D][D], C2i[D][D], eps[D], u[D];
        {
            T A1[D][D];
            load_mat<T, D, D>(A_1 + tid * D * D, A1);
            load_mat<T, D, D>(A_2 + tid * D * D, W);
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) W[i][j] = A1[i][j] - W[i][j];    // dA
        }
        load_lower<T, D>(C_1 + tid * D * D, C1);
        load_lower<T, D>(C_2 + tid * D * D, C2);
        MF_UNROLL for (int i = 0; i < D; ++i) eps[i] = b_1[tid * D + i] - b_2[tid * D + i];
        tri_inv_lower<T, D>(C2, C2i, l2, bad);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            l1.mul(C1[i][i]);
            bad |= !(C1[i][i] != T(0));
        }
        MF_UNROLL for (int j = 0; j < D; ++j) MF_UNROLL for (int i = 0; i < D; ++i) eps[i] += W[i][j] * m[j];
        trimul_lower_vec<T, D>(C2i, eps, u);
        acc += chol_terms(C2i, C1) + dot_self<T, D>(u);
        trimul_lower_inplace<T, D, D>(C2i, W);                                            // W = C2^-1 dA
        if (oN) kl_store_adjoint_inputs<T, D>(W, u, oN + id * D * D, on + id * D);
        T S[D][D];
        load_lower<T, D>(pS + id * D * D, S);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            T row[D];
            MF_UNROLL for (int j = 0; j < D; ++j) row[j] = T(0);
            MF_UNROLL for (int l = 0; l < D; ++l)
                MF_UNROLL for (int j = 0; j < D; ++j) row[j] += W[i][l] * ((l >= j) ? S[l][j] : S[j][l]);
            MF_UNROLL for (int j = 0; j < D; ++j) acc += row[j] * W[i][j];
        }
    }
    part[id] = T(0.5) * (acc - T(D)) + l2.value() - l1.value();
    if (oN && k + 1 >= Tn) kl_zero_adjoint_inputs<T, D>(oN + id * D * D, on + id * D);
    if (bad && info) raise_info(info);
}

// out[s] = sum_k part[s, k]: one wavefront per series (fixed order: deterministic)
template <typename T>
__global__ void __launch_bounds__(64) row_sum_kernel(long n, const T* __restrict__ part, T* __restrict__ out) {
    const long s = blockIdx.x;
    T a = T(0);
    for (long k = threadIdx.x; k < n; k += 64) a += part[s * n + k];
    MF_UNROLL for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if (threadIdx.x == 0) out[s] = a;
}

// (mu0, b) -> the [B, n, D] offsets of the mean recursion
template <typename T, int D>
__global__ void __launch_bounds__(256) concat_offsets_kernel(long B, long n, const T* __restrict__ mu0, const T* __restrict__ b,
                                                             T* __restrict__ offs) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * n * D) return;
    const long s = e / (n * D), r = e % (n * D);
    offs[e] = r < D ? mu0[s * D + r] : b[s * (n - 1) * D + r - D];
}

// ---- the adjoint in three kernels (round 2, second form) ------------------------------------------------------------------
// The ONE-sweep forms of the two adjoints (everything inside the sequential loop) were correct but slow where it matters: at
// BASELINE config 4's shape (B = 512, T = 1000, d = 9) a step carried four d x d matrices plus the state, spilt 3.2 KB per lane
// and took 70 us - 70 ms per backward on 8 wavefronts.  Both adjoints are the same recursion
//     M_k = N_k + A_k^T M_{k+1} A_k,      lam_k = n_k + A_k^T lam_{k+1}            (M = 2 dF/dS_k, symmetric; lam = dF/dm_k)
// driven by per-step inputs (N_k, n_k) that are LOCAL in time, and followed by parameter gradients that are local given
// (M_{k+1}, lam_{k+1}).  So:
//   1. a parallel kernel, one lane per (series, step), forms (N_k, n_k)              [KL only; `marginals` reads them directly]
//   2. a LIGHT sequential sweep, one lane per series, runs the recursion and writes M_k, lam_k: the only state is M (lower),
//      one transition matrix and one column - 180 doubles at d = 9, no spills
//   3. a parallel kernel, one lane per (series, step), turns (M_{k+1}, lam_{k+1}) into the gradients of A_k, b_k, cholQ_k
//      (the lane of step 0 also does mu0 and cholP0), row by row so that no d x d temporary is ever held whole.
// Workspace: N [B,T,d,d], n [B,T,d], M [B,T,d,d], lam [B,T,d].

template <typename T, int D> struct AdjointWs {
    T *N, *n, *M, *lam;
};

// ---- 1. KL: per-step inputs of the recursion ----------------------------------------------------------------------------------
//   N_k = dA_k^T Q2_k^-1 dA_k,   n_k = dA_k^T Q2_k^-1 eps_k   for k < T-1;   N_{T-1} = 0, n_{T-1} = 0.
template <typename T, int D>
__global__ void __launch_bounds__(64) ssm_kl_adjoint_inputs_kernel(long B, long Tn, const T* __restrict__ A_1,
                                                                   const T* __restrict__ b_1, const T* __restrict__ A_2,
                                                                   const T* __restrict__ b_2, const T* __restrict__ C_2,
                                                                   const T* __restrict__ pm, AdjointWs<T, D> ws, int* info) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * Tn) return;
    const long s = id / Tn, k = id % Tn;
    T* Nout = ws.N + id * D * D;
    T* nout = ws.n + id * D;
    if (k + 1 >= Tn) {
        MF_UNROLL for (int e = 0; e < D * D; ++e) Nout[e] = T(0);
        MF_UNROLL for (int e = 0; e < D; ++e) nout[e] = T(0);
        return;
    }
    const long tid = s * (Tn - 1) + k;
    T W[D][D], C2[D][D], C2i[D][D], mk[D], eps[D], u[D], qe[D];
    {
        T A1[D][D];
        load_mat<T, D, D>(A_1 + tid * D * D, A1);
        load_mat<T, D, D>(A_2 + tid * D * D, W);
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) W[i][j] = A1[i][j] - W[i][j];      // dA
    }
    load_lower<T, D>(C_2 + tid * D * D, C2);
    load_vec<T, D>(pm + id * D, mk);
    MF_UNROLL for (int i = 0; i < D; ++i) eps[i] = b_1[tid * D + i] - b_2[tid * D + i];
    LogAcc<T> la;
    la.init();
    bool bad = false;
    tri_inv_lower<T, D>(C2, C2i, la, bad);
    MF_UNROLL for (int j = 0; j < D; ++j) MF_UNROLL for (int i = 0; i < D; ++i) eps[i] += W[i][j] * mk[j];
    trimul_lower_vec<T, D>(C2i, eps, u);
    trimulT_lower_vec<T, D>(C2i, u, qe);                         // Q2^-1 eps
    MF_UNROLL for (int j = 0; j < D; ++j) {
        T acc = T(0);
        MF_UNROLL for (int i = 0; i < D; ++i) acc += W[i][j] * qe[i];
        nout[j] = acc;                                           // dA^T Q2^-1 eps
    }
    trimul_lower_inplace<T, D, D>(C2i, W);                       // C2^-1 dA
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j <= i; ++j) {
            T acc = T(0);
            MF_UNROLL for (int l = 0; l < D; ++l) acc += W[l][i] * W[l][j];
            Nout[i * D + j] = acc;
            Nout[j * D + i] = acc;
        }
    if (bad && info) raise_info(info);
}

// ---- 2. the recursion ---------------------------------------------------------------------------------------------------------
// Inputs either from the workspace (ws.N symmetric, ws.n) or, for `marginals`, straight from the incoming gradients:
// N_k = gS_k + gS_k^T, n_k = gm_k (either may be NULL = zero).
template <typename T, int D>
__global__ void __launch_bounds__(64) ssm_adjoint_scan_kernel(long B, long Tn, const T* __restrict__ A, const T* __restrict__ gm,
                                                              const T* __restrict__ gS, int from_ws, AdjointWs<T, D> ws) {
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    T M[D][D], lam[D];                                           // M: lower triangle
    MF_UNROLL for (int i = 0; i < D; ++i) {
        lam[i] = T(0);
        MF_UNROLL for (int j = 0; j <= i; ++j) M[i][j] = T(0);
    }
    for (long k = Tn - 1; k >= 0; --k) {
        const long id = s * Tn + k;
        T Mn[D][D], ln[D];                                       // the inputs of block k
        if (from_ws) {
            load_lower<T, D>(ws.N + id * D * D, Mn);
            load_vec<T, D>(ws.n + id * D, ln);
        } else {
            MF_UNROLL for (int i = 0; i < D; ++i) {
                ln[i] = gm ? gm[id * D + i] : T(0);
                MF_UNROLL for (int j = 0; j <= i; ++j) Mn[i][j] = gS ? gS[(id * D + i) * D + j] + gS[(id * D + j) * D + i] : T(0);
            }
        }
        if (k + 1 < Tn) {
            T Am[D][D];
            load_mat<T, D, D>(A + (s * (Tn - 1) + k) * D * D, Am);
            __builtin_amdgcn_sched_barrier(0);
            // lam <- n_k + A^T lam ;  M <- N_k + A^T (M A), one column of M A at a time
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T acc = T(0);
                MF_UNROLL for (int i = 0; i < D; ++i) acc += Am[i][j] * lam[i];
                ln[j] += acc;
            }
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T col[D];                                        // (M A)[:, j]
                MF_UNROLL for (int i = 0; i < D; ++i) col[i] = T(0);
                MF_UNROLL for (int l = 0; l < D; ++l)
                    MF_UNROLL for (int i = 0; i < D; ++i) col[i] += ((i >= l) ? M[i][l] : M[l][i]) * Am[l][j];
                MF_UNROLL for (int i = j; i < D; ++i) {          // lower triangle of A^T (M A): rows i >= j
                    T acc = T(0);
                    MF_UNROLL for (int l = 0; l < D; ++l) acc += Am[l][i] * col[l];
                    Mn[i][j] += acc;
                }
            }
        }
        MF_UNROLL for (int i = 0; i < D; ++i) {
            lam[i] = ln[i];
            MF_UNROLL for (int j = 0; j <= i; ++j) M[i][j] = Mn[i][j];
        }
        store_sym<T, D>(ws.M + id * D * D, M);
        store_vec<T, D>(ws.lam + id * D, lam);
    }
}

// the inputs of the `marginals` adjoint in the workspace layout of the scans: N = gS + gS^T (lower), n = gm
template <typename T, int D>
__global__ void __launch_bounds__(64) ssm_adjoint_sym_inputs_kernel(long blocks, const T* __restrict__ gm,
                                                                    const T* __restrict__ gS, AdjointWs<T, D> ws) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= blocks) return;
    T N[D][D], v[D];
    MF_UNROLL for (int i = 0; i < D; ++i) {
        v[i] = gm ? gm[id * D + i] : T(0);
        MF_UNROLL for (int j = 0; j <= i; ++j) N[i][j] = gS ? gS[(id * D + i) * D + j] + gS[(id * D + j) * D + i] : T(0);
    }
    store_sym<T, D>(ws.N + id * D * D, N);
    store_vec<T, D>(ws.n + id * D, v);
}

// ---- 3. parameter gradients, local in time --------------------------------------------------------------------------------------
// KL = true:  db = w (Q2^-1 eps + lam_{k+1}),  dA = w ((Q2^-1 eps + lam_{k+1}) m_k^T + (Q2^-1 dA + M_{k+1} A1) S_k),
//             dC = w (tril((Q2^-1 + M_{k+1}) C1) - diag(1 / C1));   step 0 also: dmu0 = w (P0_2^-1 d0 + lam_0),
//             dC0 = w (tril((P0_2^-1 + M_0) C0_1) - diag(1 / C0_1)).
// KL = false (`marginals`):  db = lam_{k+1},  dA = lam_{k+1} m_k^T + M_{k+1} A S_k,  dC = tril(M_{k+1} C);  dmu0 = lam_0,
//             dC0 = tril(M_0 C0).
template <typename T, int D> struct AdjointLocalArgs {
    long B, Tn;
    const T *mu0_1, *C0_1, *A_1, *b_1, *C_1;          // the chain whose gradients are formed
    const T *mu0_2, *C0_2, *A_2, *b_2, *C_2;          // KL only: the second chain
    const T *pm, *pS, *weights;
    T *gmu0, *gC0, *gA, *gb, *gC;
    int* info;
};

// out(lower) = w * (tril((Qinv + Msym) C) - ent * diag(1 / C));  Qinv = Ci^T Ci (Ci lower, may be NULL), Msym symmetric full
template <typename T, int D>
MF_DEV void adjoint_chol_grad(const T (*Ci)[D], const T (&Ms)[D][D], const T (&C)[D][D], T w, bool entropy, T* __restrict__ dst) {
    T t1[D][D];
    if (Ci) {
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T acc = T(0);
                MF_UNROLL for (int l = 0; l < D; ++l)
                    if (l <= i && j <= l) acc += Ci[i][l] * C[l][j];
                t1[i][j] = acc;
            }
    }
    T G[D][D];
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j <= i; ++j) {
            T acc = T(0);
            if (Ci) { MF_UNROLL for (int l = i; l < D; ++l) acc += Ci[l][i] * t1[l][j]; }
            MF_UNROLL for (int l = j; l < D; ++l) acc += Ms[i][l] * C[l][j];
            G[i][j] = w * (acc - ((entropy && i == j) ? t_rcp<T>(C[i][i]) : T(0)));
        }
    store_lower<T, D>(dst, G);
}

template <typename T, int D, bool KL>
__global__ void __launch_bounds__(64) ssm_adjoint_local_kernel(AdjointLocalArgs<T, D> a, AdjointWs<T, D> ws) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= a.B * a.Tn) return;
    const long s = id / a.Tn, k = id % a.Tn;
    const T w = (KL && a.weights) ? a.weights[s] : T(1);
    LogAcc<T> la;
    la.init();
    bool bad = false;
    if (k == 0) {
        // ---- prior of the first state ----------------------------------------------------------------------------------
        T M0[D][D], C1[D][D], g[D];
        load_mat<T, D, D>(ws.M + id * D * D, M0);
        load_vec<T, D>(ws.lam + id * D, g);
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) C1[i][j] = T(0);
        load_lower<T, D>(a.C0_1 + s * D * D, C1);
        if (KL) {
            T C2[D][D], C2i[D][D], d0[D], u[D], q[D];
            load_lower<T, D>(a.C0_2 + s * D * D, C2);
            tri_inv_lower<T, D>(C2, C2i, la, bad);
            MF_UNROLL for (int i = 0; i < D; ++i) {
                d0[i] = a.mu0_1[s * D + i] - a.mu0_2[s * D + i];
                bad |= !(C1[i][i] != T(0));
            }
            trimul_lower_vec<T, D>(C2i, d0, u);
            trimulT_lower_vec<T, D>(C2i, u, q);
            MF_UNROLL for (int i = 0; i < D; ++i) a.gmu0[s * D + i] = w * (q[i] + g[i]);
            adjoint_chol_grad<T, D>(C2i, M0, C1, w, true, a.gC0 + s * D * D);
        } else {
            MF_UNROLL for (int i = 0; i < D; ++i) a.gmu0[s * D + i] = g[i];
            adjoint_chol_grad<T, D>(static_cast<const T(*)[D]>(nullptr), M0, C1, T(1), false, a.gC0 + s * D * D);
        }
    }
    if (k + 1 >= a.Tn) {
        if (bad && a.info) raise_info(a.info);
        return;
    }
    // ---- transition k -> k+1 ---------------------------------------------------------------------------------------------
    const long tid = s * (a.Tn - 1) + k;
    T Mn[D][D], gl[D], A1[D][D], mk[D];
    load_mat<T, D, D>(ws.M + (id + 1) * D * D, Mn);               // M_{k+1} (symmetric, stored full)
    load_vec<T, D>(ws.lam + (id + 1) * D, gl);                    // lam_{k+1}
    load_mat<T, D, D>(a.A_1 + tid * D * D, A1);
    load_vec<T, D>(a.pm + id * D, mk);
    T C2i[D][D], W[D][D];                                         // KL only
    if (KL) {
        T C2[D][D], eps[D], u[D], qe[D];
        load_mat<T, D, D>(a.A_2 + tid * D * D, W);
        load_lower<T, D>(a.C_2 + tid * D * D, C2);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            eps[i] = a.b_1[tid * D + i] - a.b_2[tid * D + i];
            MF_UNROLL for (int j = 0; j < D; ++j) W[i][j] = A1[i][j] - W[i][j];
        }
        tri_inv_lower<T, D>(C2, C2i, la, bad);
        MF_UNROLL for (int j = 0; j < D; ++j) MF_UNROLL for (int i = 0; i < D; ++i) eps[i] += W[i][j] * mk[j];
        trimul_lower_vec<T, D>(C2i, eps, u);
        trimulT_lower_vec<T, D>(C2i, u, qe);
        MF_UNROLL for (int i = 0; i < D; ++i) gl[i] += qe[i];
        trimul_lower_inplace<T, D, D>(C2i, W);                    // C2^-1 dA   (Q2^-1 dA = C2i^T W, formed row by row below)
    }
    MF_UNROLL for (int i = 0; i < D; ++i) a.gb[tid * D + i] = w * gl[i];
    {
        // dA row by row: G[i][:] = sum_l M[i][l] A1[l][:] (+ sum_{l >= i} C2i[l][i] W[l][:]);  out[i][:] = gl[i] m^T + G[i][:] S
        T Sk[D][D];
        load_mat<T, D, D>(a.pS + id * D * D, Sk);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            T grow[D], orow[D];
            MF_UNROLL for (int j = 0; j < D; ++j) grow[j] = T(0);
            MF_UNROLL for (int l = 0; l < D; ++l) MF_UNROLL for (int j = 0; j < D; ++j) grow[j] += Mn[i][l] * A1[l][j];
            if (KL) { MF_UNROLL for (int l = i; l < D; ++l) MF_UNROLL for (int j = 0; j < D; ++j) grow[j] += C2i[l][i] * W[l][j]; }
            MF_UNROLL for (int j = 0; j < D; ++j) orow[j] = gl[i] * mk[j];
            MF_UNROLL for (int l = 0; l < D; ++l) MF_UNROLL for (int j = 0; j < D; ++j) orow[j] += grow[l] * Sk[l][j];
            MF_UNROLL for (int j = 0; j < D; ++j) a.gA[(tid * D + i) * D + j] = w * orow[j];
        }
    }
    {
        T C1[D][D];
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) C1[i][j] = T(0);
        load_lower<T, D>(a.C_1 + tid * D * D, C1);
        if (KL) {
            MF_UNROLL for (int i = 0; i < D; ++i) bad |= !(C1[i][i] != T(0));
            adjoint_chol_grad<T, D>(C2i, Mn, C1, w, true, a.gC + tid * D * D);
        } else {
            adjoint_chol_grad<T, D>(static_cast<const T(*)[D]>(nullptr), Mn, C1, T(1), false, a.gC + tid * D * D);
        }
    }
    if (bad && a.info) raise_info(a.info);
}

}  // namespace mf
