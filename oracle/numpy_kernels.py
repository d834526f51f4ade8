"""
CPU oracle (numpy, fp64) for the SDE-kernel -> state-space-model step.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as numpy_oracle.py).

Restates the closed forms of the reference's kernels:
  Matern12  markovflow/kernels/matern.py:66-110      A = exp(-dt/l),                        Pinf = var
  Matern32  markovflow/kernels/matern.py:299-356     A = e^{-lam dt}(I + (F + lam I) dt),   lam = sqrt(3)/l
  Matern52  markovflow/kernels/matern.py:434-501     A = e^{-lam dt}(I + N dt + N^2 dt^2/2), lam = sqrt(5)/l
  Q_k = Pinf - A_k Pinf A_k^T + jitter I             markovflow/kernels/sde_kernel.py:421-446
  Sum / IndependentMultiOutput: block-diagonal A, Pinf; H = [H1, H2, ...] / H1 (+) H2 (+) ...   sde_kernel.py:592-690,847-878
Parity pin: tests/test_oracle_golden.py checks these against tests/golden/kernels_matern_T24.npz, produced by the
reference's own numpy/scipy-expm test kernels (tests/tools/kernels/kernels.py).
"""
import numpy as np

ORDER_SIZE = {1: 1, 3: 2, 5: 3}


def matern_feedback_and_pinf(order: int, length_scale: float, variance: float):
    lam = np.sqrt(order) / length_scale
    if order == 1:
        return np.array([[-lam]]), np.array([[variance]])
    if order == 3:
        return (np.array([[0.0, 1.0], [-lam ** 2, -2 * lam]]), variance * np.diag([1.0, lam ** 2]))
    if order == 5:
        l23 = lam ** 2 / 3.0
        f = np.array([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [-lam ** 3, -3 * lam ** 2, -3 * lam]])
        p = variance * np.array([[1.0, 0.0, -l23], [0.0, l23, 0.0], [-l23, 0.0, lam ** 4]])
        return f, p
    raise ValueError(order)


def matern_transitions(order: int, length_scale: float, variance: float, dt: np.ndarray, jitter: float = 0.0):
    """(A [..., n, k, k], Q [..., n, k, k], Pinf [k, k]) for time gaps dt [..., n]."""
    f, pinf = matern_feedback_and_pinf(order, length_scale, variance)
    lam = np.sqrt(order) / length_scale
    k = f.shape[0]
    nil = f + lam * np.eye(k)                       # nilpotent of index k
    dtm = dt[..., None, None]
    a = np.eye(k) + nil * dtm
    if k == 3:
        a = a + (nil @ nil) * (0.5 * dtm ** 2)
    a = a * np.exp(-lam * dtm)
    q = pinf - a @ pinf @ np.swapaxes(a, -1, -2) + jitter * np.eye(k)
    return a, q, pinf


def concat_transitions(orders, length_scales, variances, dt, jitter=0.0):
    """Block-diagonal (A, Q, Pinf) of a Sum / IndependentMultiOutput of Matern components."""
    parts = [matern_transitions(o, l, v, dt, jitter) for o, l, v in zip(orders, length_scales, variances)]
    d = sum(ORDER_SIZE[o] for o in orders)
    a = np.zeros(dt.shape + (d, d))
    q = np.zeros(dt.shape + (d, d))
    p = np.zeros((d, d))
    off = 0
    for (ai, qi, pi), o in zip(parts, orders):
        k = ORDER_SIZE[o]
        a[..., off:off + k, off:off + k] = ai
        q[..., off:off + k, off:off + k] = qi
        p[off:off + k, off:off + k] = pi
        off += k
    return a, q, p


def emission(orders, num_points_shape, independent_outputs: bool):
    """H: [..., T, m, d]; Sum: one output reading the first state of every component; IMO: one output per component."""
    d = sum(ORDER_SIZE[o] for o in orders)
    m = len(orders) if independent_outputs else 1
    h = np.zeros((m, d))
    off = 0
    for j, o in enumerate(orders):
        h[j if independent_outputs else 0, off] = 1.0
        off += ORDER_SIZE[o]
    return np.broadcast_to(h, tuple(num_points_shape) + (m, d)).copy()


def dense_kernel_matrix(orders, length_scales, variances, t):
    """k(t, t') of the SUM of Matern components, for the dense GP marginal likelihood."""
    r = np.abs(t[:, None] - t[None, :])
    out = np.zeros_like(r)
    for o, l, v in zip(orders, length_scales, variances):
        lam = np.sqrt(o) / l
        poly = {1: 1.0, 3: 1.0 + lam * r, 5: 1.0 + lam * r + (lam * r) ** 2 / 3.0}[o]
        out += v * poly * np.exp(-lam * r)
    return out


def dense_gp_predict(orders, length_scales, variances, t, y, noise, t_new):
    """Dense GP posterior of f at t_new (mean, variance) - Rasmussen & Williams eq. 2.25-2.26; the check of
    markovflow/posterior.py:231-258 (predict_f) used by the reference's GPR tests."""
    def k(a, b):
        r = np.abs(a[:, None] - b[None, :])
        out = np.zeros_like(r)
        for o, l, v in zip(orders, length_scales, variances):
            lam = np.sqrt(o) / l
            poly = {1: 1.0, 3: 1.0 + lam * r, 5: 1.0 + lam * r + (lam * r) ** 2 / 3.0}[o]
            out += v * poly * np.exp(-lam * r)
        return out
    kn = k(t, t) + noise * np.eye(len(t))
    ks = k(t_new, t)
    mean = ks @ np.linalg.solve(kn, y)
    var = np.diag(k(t_new, t_new)) - np.einsum("ij,ji->i", ks, np.linalg.solve(kn, ks.T))
    return mean, var
