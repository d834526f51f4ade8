// Reverse-mode gradient of  KL(q1 || q2)  between two state space models with respect to the parameters of q1
// (markovflow/state_space_model.py:528-593 is differentiated by TensorFlow in the reference; pinned by
// tests/integration/models/test_variational.py:123-132).
//
// With  dA = A1 - A2,  db = b1 - b2  and the marginals (m_k, S_k) of q1 the divergence is a sum of local terms,
//   KL = 1/2 tr(P0_2^-1 (S_0 + d0 d0^T)) + 1/2 sum_k [ tr(Q2_k^-1 Q1_k) + tr(Q2_k^-1 dA_k (S_k + m_k m_k^T) dA_k^T)
//        + 2 db_k^T Q2_k^-1 dA_k m_k + db_k^T Q2_k^-1 db_k ] - H(q1) + terms of q2 alone,
// so the gradient is the local partial derivative plus the adjoint of the moment recursion  m_{k+1} = A1 m_k + b1,
// S_{k+1} = A1 S_k A1^T + Q1  - ONE backward sweep per series carrying  lam_k (d)  and  M_k = 2 dKL/dS_k (d x d, symmetric):
//   lam_k = dA_k^T Q2_k^-1 eps_k + A1_k^T lam_{k+1},      eps_k = dA_k m_k + db_k,
//   M_k   = dA_k^T Q2_k^-1 dA_k + A1_k^T M_{k+1} A1_k,    lam_{T-1} = 0, M_{T-1} = 0,
//   dKL/db1_k = Q2_k^-1 eps_k + lam_{k+1},
//   dKL/dA1_k = (Q2_k^-1 eps_k + lam_{k+1}) m_k^T + (Q2_k^-1 dA_k + M_{k+1} A1_k) S_k,
//   dKL/dC1_k = tril((Q2_k^-1 + M_{k+1}) C1_k) - diag(1 / C1_k)          (Q1 = C1 C1^T; the last term is the entropy),
//   dKL/dmu0_1 = P0_2^-1 d0 + lam_0,   dKL/dC0_1 = tril((P0_2^-1 + M_0) C0_1) - diag(1 / C0_1).
// The gradient with respect to q2 is local in time (minus the expected complete-data score of q2 under q1's marginals) and
// comes from kf_grad_kernel.  One lane per series, natural (backward) order.
#pragma once
#include "mf_small.hpp"

namespace mf {

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
                                                    int* info) {
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
    if (bad && info) raise_info(info);
}

template <typename T, int D>
struct KlGradArgs {
    long B, Tn;
    const T *mu0_1, *C0_1, *A_1, *b_1, *C_1;
    const T *mu0_2, *C0_2, *A_2, *b_2, *C_2;
    const T *pm, *pS, *weights;
    T *gmu0, *gC0, *gA, *gb, *gC;
    int* info;
};

// out(lower, incl. diagonal) = tril((C2i^T C2i + M) C1) - diag(1 / C1);   C1, C2i lower triangular, M full symmetric
template <typename T, int D>
MF_DEV void kl_chol_grad(const T (&C2i)[D][D], const T (&M)[D][D], const T (&C1)[D][D], T w, T* __restrict__ dst) {
    T t1[D][D], G[D][D];
    // t1 = C2i C1 (lower x lower); only the lower triangle is non-zero
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j < D; ++j) {
            T acc = T(0);
            MF_UNROLL for (int l = 0; l < D; ++l)
                if (l <= i && j <= l) acc += C2i[i][l] * C1[l][j];
            t1[i][j] = acc;
        }
    // G = tril(C2i^T t1 + M C1)
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j <= i; ++j) {
            T acc = T(0);
            MF_UNROLL for (int l = i; l < D; ++l) acc += C2i[l][i] * t1[l][j];
            MF_UNROLL for (int l = j; l < D; ++l) acc += M[i][l] * C1[l][j];
            G[i][j] = w * (acc - ((i == j) ? t_rcp<T>(C1[i][i]) : T(0)));
        }
    store_lower<T, D>(dst, G);
}

template <typename T, int D>
__global__ void __launch_bounds__(64) ssm_kl_grad_kernel(KlGradArgs<T, D> a) {
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.B) return;
    const T w = a.weights ? a.weights[s] : T(1);
    T lam[D], M[D][D];
    MF_UNROLL for (int i = 0; i < D; ++i) {
        lam[i] = T(0);
        MF_UNROLL for (int j = 0; j < D; ++j) M[i][j] = T(0);
    }
    LogAcc<T> la;
    la.init();
    bool bad = false;
    for (long k = a.Tn - 2; k >= 0; --k) {
        const long tid = s * (a.Tn - 1) + k, id = s * a.Tn + k;
        T A1[D][D], dA[D][D], C2[D][D], C2i[D][D], mk[D], eps[D];
        load_mat<T, D, D>(a.A_1 + tid * D * D, A1);
        load_mat<T, D, D>(a.A_2 + tid * D * D, dA);
        load_lower<T, D>(a.C_2 + tid * D * D, C2);
        load_vec<T, D>(a.pm + id * D, mk);
        MF_UNROLL for (int i = 0; i < D; ++i) {
            eps[i] = a.b_1[tid * D + i] - a.b_2[tid * D + i];
            MF_UNROLL for (int j = 0; j < D; ++j) dA[i][j] = A1[i][j] - dA[i][j];
        }
        tri_inv_lower<T, D>(C2, C2i, la, bad);
        la.init();                                              // the running product is not used here
        MF_UNROLL for (int j = 0; j < D; ++j) MF_UNROLL for (int i = 0; i < D; ++i) eps[i] += dA[i][j] * mk[j];
        T u[D], gl[D];
        trimul_lower_vec<T, D>(C2i, eps, u);
        trimulT_lower_vec<T, D>(C2i, u, gl);                    // Q2^-1 eps
        // adjoint of the mean, part 1 (needs Q2^-1 eps before lam is folded into gl)
        T lam_new[D];
        MF_UNROLL for (int j = 0; j < D; ++j) lam_new[j] = T(0);
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) lam_new[j] += dA[i][j] * gl[i] + A1[i][j] * lam[i];
        MF_UNROLL for (int i = 0; i < D; ++i) gl[i] += lam[i];
        T W[D][D], G[D][D], MA[D][D];
        trimul_lower<T, D, D>(C2i, dA, W);                      // C2^-1 dA
        trimulT_lower<T, D, D>(C2i, W, G);                      // Q2^-1 dA
        MF_UNROLL for (int i = 0; i < D; ++i) {
            MF_UNROLL for (int j = 0; j < D; ++j) MA[i][j] = M[i][0] * A1[0][j];
            MF_UNROLL for (int l = 1; l < D; ++l)
                MF_UNROLL for (int j = 0; j < D; ++j) MA[i][j] += M[i][l] * A1[l][j];
            MF_UNROLL for (int j = 0; j < D; ++j) G[i][j] += MA[i][j];
        }
        {
            // dA1 = w (gl m_k^T + G S_k),  db1 = w gl
            T Sk[D][D], out[D][D];
            load_mat<T, D, D>(a.pS + id * D * D, Sk);
            MF_UNROLL for (int i = 0; i < D; ++i) {
                MF_UNROLL for (int j = 0; j < D; ++j) out[i][j] = gl[i] * mk[j];
                MF_UNROLL for (int l = 0; l < D; ++l)
                    MF_UNROLL for (int j = 0; j < D; ++j) out[i][j] += G[i][l] * Sk[l][j];
                MF_UNROLL for (int j = 0; j < D; ++j) out[i][j] *= w;
            }
            store_mat<T, D, D>(a.gA + tid * D * D, out);
            MF_UNROLL for (int i = 0; i < D; ++i) a.gb[tid * D + i] = w * gl[i];
        }
        {
            T C1[D][D];
            MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) C1[i][j] = T(0);
            load_lower<T, D>(a.C_1 + tid * D * D, C1);
            MF_UNROLL for (int i = 0; i < D; ++i) bad |= !(C1[i][i] != T(0));
            kl_chol_grad<T, D>(C2i, M, C1, w, a.gC + tid * D * D);
        }
        // M <- W^T W + A1^T (M A1)
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T acc = T(0);
                MF_UNROLL for (int l = 0; l < D; ++l) acc += W[l][i] * W[l][j] + A1[l][i] * MA[l][j];
                M[i][j] = acc;
            }
        // keep M exactly symmetric (it is in exact arithmetic; rounding would otherwise drift over long chains)
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < i; ++j) {
                const T v = T(0.5) * (M[i][j] + M[j][i]);
                M[i][j] = v;
                M[j][i] = v;
            }
        MF_UNROLL for (int i = 0; i < D; ++i) lam[i] = lam_new[i];
    }
    {
        T C2[D][D], C2i[D][D], d0[D], u[D], g[D], C1[D][D];
        load_lower<T, D>(a.C0_2 + s * D * D, C2);
        tri_inv_lower<T, D>(C2, C2i, la, bad);
        MF_UNROLL for (int i = 0; i < D; ++i) d0[i] = a.mu0_1[s * D + i] - a.mu0_2[s * D + i];
        trimul_lower_vec<T, D>(C2i, d0, u);
        trimulT_lower_vec<T, D>(C2i, u, g);
        MF_UNROLL for (int i = 0; i < D; ++i) a.gmu0[s * D + i] = w * (g[i] + lam[i]);
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) C1[i][j] = T(0);
        load_lower<T, D>(a.C0_1 + s * D * D, C1);
        MF_UNROLL for (int i = 0; i < D; ++i) bad |= !(C1[i][i] != T(0));
        kl_chol_grad<T, D>(C2i, M, C1, w, a.gC0 + s * D * D);
    }
    if (bad && a.info) raise_info(a.info);
}


// Adjoint of the moment recursion itself: given the incoming gradients gm [B,T,d] and gS [B,T,d,d] of a scalar with respect to
// the marginal means and covariances of a chain (either may be NULL = zero), the gradients with respect to the chain's
// parameters.  This is what makes `StateSpaceModel.marginals` differentiable (the reference differentiates
// state_space_model.py:232-262 through TensorFlow; the expected log-likelihood term of every variational model goes through
// it, e.g. models/variational.py:150, sparse_variational.py:178-192):
//   lam_k = gm_k + A_k^T lam_{k+1},   L_k = sym(gS_k) + A_k^T L_{k+1} A_k,
//   db_k = lam_{k+1},  dA_k = lam_{k+1} m_k^T + 2 L_{k+1} A_k S_k,  dC_k = 2 tril(L_{k+1} C_k),  dmu0 = lam_0,  dC0 = 2 tril(L_0 C0).
template <typename T, int D>
__global__ void __launch_bounds__(64) ssm_marginals_grad_kernel(long B, long Tn, const T* __restrict__ C0,
                                                                const T* __restrict__ A, const T* __restrict__ C,
                                                                const T* __restrict__ pm, const T* __restrict__ pS,
                                                                const T* __restrict__ gm, const T* __restrict__ gS,
                                                                T* __restrict__ gmu0, T* __restrict__ gC0, T* __restrict__ gA,
                                                                T* __restrict__ gb, T* __restrict__ gC) {
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    T lam[D], L[D][D];
    auto add_incoming = [&](long id, bool first) {
        MF_UNROLL for (int i = 0; i < D; ++i) {
            const T v = gm ? gm[id * D + i] : T(0);
            lam[i] = first ? v : lam[i] + v;
        }
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                const T v = gS ? T(0.5) * (gS[(id * D + i) * D + j] + gS[(id * D + j) * D + i]) : T(0);
                const T nv = first ? v : T(0.5) * (L[i][j] + L[j][i]) + v;
                L[i][j] = nv;
                L[j][i] = nv;
            }
    };
    add_incoming(s * Tn + Tn - 1, true);
    for (long k = Tn - 2; k >= 0; --k) {
        const long tid = s * (Tn - 1) + k, id = s * Tn + k;
        T Am[D][D], Cm[D][D], mk[D], Sk[D][D], LA[D][D], out[D][D];
        load_mat<T, D, D>(A + tid * D * D, Am);
        load_vec<T, D>(pm + id * D, mk);
        load_mat<T, D, D>(pS + id * D * D, Sk);
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Cm[i][j] = T(0);
        load_lower<T, D>(C + tid * D * D, Cm);
        MF_UNROLL for (int i = 0; i < D; ++i) gb[tid * D + i] = lam[i];
        MF_UNROLL for (int i = 0; i < D; ++i) {
            MF_UNROLL for (int j = 0; j < D; ++j) LA[i][j] = L[i][0] * Am[0][j];
            MF_UNROLL for (int l = 1; l < D; ++l)
                MF_UNROLL for (int j = 0; j < D; ++j) LA[i][j] += L[i][l] * Am[l][j];
        }
        MF_UNROLL for (int i = 0; i < D; ++i) {
            MF_UNROLL for (int j = 0; j < D; ++j) out[i][j] = lam[i] * mk[j];
            MF_UNROLL for (int l = 0; l < D; ++l)
                MF_UNROLL for (int j = 0; j < D; ++j) out[i][j] += T(2) * LA[i][l] * Sk[l][j];
        }
        store_mat<T, D, D>(gA + tid * D * D, out);
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j <= i; ++j) {
                T acc = T(0);
                MF_UNROLL for (int l = j; l < D; ++l) acc += L[i][l] * Cm[l][j];
                out[i][j] = T(2) * acc;
            }
        store_lower<T, D>(gC + tid * D * D, out);
        // lam <- A^T lam,  L <- A^T (L A);  then the incoming gradients of block k
        T ln[D];
        MF_UNROLL for (int j = 0; j < D; ++j) ln[j] = T(0);
        MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) ln[j] += Am[i][j] * lam[i];
        MF_UNROLL for (int i = 0; i < D; ++i) lam[i] = ln[i];
        MF_UNROLL for (int i = 0; i < D; ++i)
            MF_UNROLL for (int j = 0; j < D; ++j) {
                T acc = T(0);
                MF_UNROLL for (int l = 0; l < D; ++l) acc += Am[l][i] * LA[l][j];
                L[i][j] = acc;
            }
        add_incoming(id, false);
    }
    MF_UNROLL for (int i = 0; i < D; ++i) gmu0[s * D + i] = lam[i];
    T Cm[D][D], out[D][D];
    MF_UNROLL for (int i = 0; i < D; ++i) MF_UNROLL for (int j = 0; j < D; ++j) Cm[i][j] = T(0);
    load_lower<T, D>(C0 + s * D * D, Cm);
    MF_UNROLL for (int i = 0; i < D; ++i)
        MF_UNROLL for (int j = 0; j <= i; ++j) {
            T acc = T(0);
            MF_UNROLL for (int l = j; l < D; ++l) acc += L[i][l] * Cm[l][j];
            out[i][j] = T(2) * acc;
        }
    store_lower<T, D>(gC0 + s * D * D, out);
}

}  // namespace mf
