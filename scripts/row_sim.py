"""
numpy emulation of the row kernels (csrc/mf_row.hpp): ONE 16-lane DPP row = one (series, chunk), lane r < D holds row r
(or column r) of every D x D matrix in D registers, lane D holds the vector quantities; the only cross-lane primitive is the
`row_newbcast` operand of a fused multiply-add.  The emulation executes the kernel's op sequence register by register with
exactly that primitive, so a layout or sign mistake shows up here, on the CPU, before anything is compiled.

    python3 scripts/row_sim.py            # log-likelihood of random chains via emulated chunks vs the numpy oracle

Test infrastructure (it imports the oracle); not part of the product.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import numpy_oracle as O  # noqa: E402

LANES = 16


class Reg:
    """one VGPR (pair) of a 16-lane row"""

    def __init__(self, v=0.0):
        self.v = np.full(LANES, v, dtype=np.float64) if np.isscalar(v) else np.array(v, dtype=np.float64)


def fmac_bc(acc, src, own, k):      # v_fmac_f64_dpp acc, src(row_newbcast:k), own
    acc.v = acc.v + src.v[k] * own.v


def fnmac_bc(acc, src, own, k):     # the same with the neg modifier on the broadcast operand
    acc.v = acc.v - src.v[k] * own.v


def mov_bc(src, k):                 # v_mov_b64_dpp
    return Reg(np.full(LANES, src.v[k]))


def regs(n, v=0.0):
    return [Reg(v) for _ in range(n)]


class ChunkRow:
    def __init__(self, D, m):
        self.D, self.m = D, m
        r = np.arange(LANES)
        self.r = r
        self.Id = [Reg((r == i).astype(float)) for i in range(D)]
        self.ED = Reg((r == D).astype(float))
        self.Phi = regs(D)
        self.Xa = regs(D)
        self.GU = regs(D)
        self.gU = Reg()
        self.quad, self.ww, self.yry = Reg(), Reg(), Reg()
        self.logC = Reg()      # distributed sum of log diag(C)
        self.logL = Reg()      # replicated

    # ---- loads as the kernel does them -----------------------------------------------------------------
    def load_transition(self, C, A, mvec):
        D, r = self.D, self.r
        rc = np.minimum(r, D - 1)
        Crow = [Reg(C[rc, k]) for k in range(D)]                     # lane r: row r of C (lanes >= D: clamped copies)
        cdiag = Reg(C[rc, rc])
        Aa = []
        for i in range(D):
            v = np.where(r < D, A[i, rc], mvec[i])                   # lane r < D: A[i][r]; lanes >= D: mvec[i]
            Aa.append(Reg(v))
        return Crow, cdiag, Aa

    def load_obs(self, H, y):
        D, r = self.D, self.r
        rc = np.minimum(r, D - 1)
        return [Reg(np.where(r < D, H[o, rc], y[o])) for o in range(self.m)]

    # ---- ops -------------------------------------------------------------------------------------------------
    def whiten(self, Crow, cdiag, Aa):
        """dinvr, Ba = C^-1 [A | mvec] (columns), CiT = C^-1 [I | mvec] (columns)"""
        D = self.D
        dinv = Reg(1.0 / cdiag.v)
        self.logC.v = self.logC.v + np.log(np.abs(cdiag.v))
        dinvr = [mov_bc(dinv, i) for i in range(D)]
        Ba, CiT = regs(D), regs(D)
        for i in range(D):
            acc = Reg(Aa[i].v.copy())
            for k in range(i):
                fnmac_bc(acc, Crow[k], Ba[k], i)
            Ba[i].v = acc.v * dinvr[i].v
        for i in range(D):
            acc = Reg(Aa[i].v * self.ED.v + self.Id[i].v)
            for k in range(i):
                fnmac_bc(acc, Crow[k], CiT[k], i)
            CiT[i].v = acc.v * dinvr[i].v
        for k in range(D):
            self.ww.v = self.ww.v + Ba[k].v * Ba[k].v
        return Ba, CiT

    def new_pivot(self, CiT, W, Ha, Rinv):
        """Phi' = Ci^T Ci - W W^T + H^T R^-1 H in the matrix lanes; rn - W z + H^T R^-1 y in lane D"""
        D, m = self.D, self.m
        Pn = regs(D)
        for j in range(D):
            for k in range(j, D):
                fmac_bc(Pn[j], CiT[k], CiT[k], j)
        if W is not None:
            for j in range(D):
                for k in range(D):
                    fnmac_bc(Pn[j], W[k], W[k], j)
        u = regs(m)
        for o in range(m):
            for p in range(m):
                u[o].v = u[o].v + Rinv[o, p] * Ha[p].v
        for o in range(m):
            self.yry.v = self.yry.v + Ha[o].v * u[o].v
        for j in range(D):
            for o in range(m):
                fmac_bc(Pn[j], Ha[o], u[o], j)
        self.Phi = Pn

    def start(self, C, A, mvec, H, y, Rinv, has_separator):
        """first block of the chunk: block 0 of the series (A = 0) or the block after the separator"""
        D = self.D
        if not has_separator:
            A = np.zeros((D, D))
        Crow, cdiag, Aa = self.load_transition(C, A, mvec)
        Ha = self.load_obs(H, y)
        Ba, CiT = self.whiten(Crow, cdiag, Aa)
        # GU = B^T B, gU = -B^T w, X = -Ci^T B in column layout
        self.GU = regs(D)
        for b in range(D):
            for k in range(D):
                fmac_bc(self.GU[b], Ba[k], Ba[k], b)
        self.gU = Reg()
        for k in range(D):
            fnmac_bc(self.gU, Ba[k], Ba[k], D)
        self.Xa = regs(D)
        for i in range(D):
            for k in range(D):
                fnmac_bc(self.Xa[i], CiT[k], Ba[k], i)
        self.new_pivot(CiT, None, Ha, Rinv)

    def step(self, C, A, mvec, H, y, Rinv):
        D = self.D
        Crow, cdiag, Aa = self.load_transition(C, A, mvec)
        Ha = self.load_obs(H, y)
        Ba, CiT = self.whiten(Crow, cdiag, Aa)
        # S rows (lanes < D) and the right-hand side row t - B^T w (lane D)
        S = [Reg(self.Phi[j].v * self.ED.v) for j in range(D)]
        for j in range(D):
            for k in range(D):
                fnmac_bc(S[j], Ba[k], CiT[k], j)
        # pivot of block k-1 complete
        for j in range(D):
            for k in range(D):
                fmac_bc(self.Phi[j], Ba[k], Ba[k], j)
        # right-looking Cholesky in place; V = L^-1 X (columns) and W = S L^-T (rows, z in lane D) ride along
        W = regs(D)
        for j in range(D):
            s_ = mov_bc(self.Phi[j], j)
            inv = Reg(1.0 / np.sqrt(s_.v))
            self.logL.v = self.logL.v + np.log(s_.v * inv.v)
            self.Phi[j].v = self.Phi[j].v * inv.v
            self.Xa[j].v = self.Xa[j].v * inv.v
            W[j].v = S[j].v * inv.v
            for k in range(j + 1, D):
                fnmac_bc(self.Phi[k], self.Phi[j], self.Phi[j], k)
                fnmac_bc(self.Xa[k], self.Phi[j], self.Xa[j], k)
                fnmac_bc(S[k], self.Phi[j], W[j], k)
        for k in range(D):
            self.quad.v = self.quad.v + W[k].v * W[k].v
        # separator: GU -= V^T V, gU -= V^T z
        for b in range(D):
            for k in range(D):
                fnmac_bc(self.GU[b], self.Xa[k], self.Xa[k], b)
        for k in range(D):
            fnmac_bc(self.gU, W[k], self.Xa[k], D)
        # X' = -W V
        Xn = regs(D)
        for i in range(D):
            for k in range(D):
                fnmac_bc(Xn[i], W[k], self.Xa[k], i)
        self.Xa = Xn
        self.new_pivot(CiT, W, Ha, Rinv)

    def result(self):
        D = self.D
        Dv = np.array([[self.Phi[j].v[i] for j in range(D)] for i in range(D)])
        tv = np.array([self.Phi[j].v[D] for j in range(D)])
        GU = np.array([[self.GU[j].v[i] for j in range(D)] for i in range(D)])
        gU = self.gU.v[:D].copy()
        F = np.array([[self.Xa[i].v[j] for j in range(D)] for i in range(D)])
        sc = -0.5 * (self.yry.v[D] + self.ww.v[D]) + 0.5 * self.quad.v[D] - self.logC.v[:D].sum() - self.logL.v[0]
        return Dv, tv, GU, gU, F, sc


def loglik_by_chunks(mu0, cholP0, A, b, cholQ, H, y, Rinv, P):
    """one series through P emulated chunks + a dense solve of the reduced system"""
    Tn, m, D = H.shape[0], H.shape[1], mu0.shape[0]
    red = []
    for c in range(P):
        k0, k1 = (c * Tn) // P, ((c + 1) * Tn) // P
        row = ChunkRow(D, m)
        for k in range(k0, k1):
            C = cholP0 if k == 0 else cholQ[k - 1]
            mv = mu0 if k == 0 else b[k - 1]
            Ak = A[k - 1] if k > 0 else None
            if k == k0:
                row.start(C, Ak, mv, H[k], y[k], Rinv, has_separator=k > 0)
            else:
                row.step(C, Ak, mv, H[k], y[k], Rinv)
        red.append(row.result())
    n = P * D
    M, rhs, sc = np.zeros((n, n)), np.zeros(n), 0.0
    for j, (Dv, tv, GU, gU, F, s) in enumerate(red):
        sl = slice(j * D, (j + 1) * D)
        M[sl, sl] += Dv
        rhs[sl] += tv
        sc += s
        if j > 0:
            pl = slice((j - 1) * D, j * D)
            M[pl, pl] += GU
            rhs[pl] += gU
            M[sl, pl] += F
            M[pl, sl] += F.T
    Lm = np.linalg.cholesky(M)
    z = np.linalg.solve(Lm, rhs)
    add_const = -0.5 * m * Tn * np.log(2 * np.pi) + 0.5 * Tn * np.linalg.slogdet(Rinv)[1]
    return add_const + sc + 0.5 * z @ z - np.log(np.diag(Lm)).sum()


def main():
    rng = np.random.default_rng(5)
    worst = 0.0
    for D, m, Tn, P in [(9, 3, 23, 4), (7, 1, 17, 3), (8, 2, 9, 9), (9, 3, 5, 1), (3, 1, 12, 2), (15, 4, 11, 2)]:
        mu0 = rng.normal(size=D)
        cholP0 = np.tril(rng.normal(size=(D, D))) * 0.3 + np.eye(D)
        A = rng.normal(size=(Tn - 1, D, D)) * 0.3
        b = rng.normal(size=(Tn - 1, D))
        cholQ = np.tril(rng.normal(size=(Tn - 1, D, D))) * 0.2 + 0.7 * np.eye(D)
        H = rng.normal(size=(Tn, m, D))
        y = rng.normal(size=(Tn, m))
        cr = np.tril(rng.normal(size=(m, m))) * 0.2 + np.eye(m)
        Rinv = np.linalg.inv(cr @ cr.T)
        ref = O.kf_log_likelihood(mu0[None], cholP0[None], A[None], b[None], cholQ[None], H[None], y[None], Rinv)
        ref = float(np.asarray(ref).reshape(-1)[0])
        got = loglik_by_chunks(mu0, cholP0, A, b, cholQ, H, y, Rinv, P)
        rel = abs(got - ref) / abs(ref)
        worst = max(worst, rel)
        print(f"D={D} m={m} T={Tn} P={P}: emulated {got:.12f}  oracle {ref:.12f}  rel {rel:.2e}")
    assert worst < 1e-10, worst
    print("row emulation agrees with the oracle")


if __name__ == "__main__":
    main()
