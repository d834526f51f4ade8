import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import markovflow_amd as mfa
rng = np.random.default_rng(5)
worst = 0.0
for case in range(160):
    big = case % 8 == 7
    d = int(rng.integers(10, 40)) if big else int(rng.integers(1, 10))
    t = int(rng.integers(2, 500)); bsz = int(rng.choice([1, 2, 3, 7, 40])) if case % 20 else 4100
    if bsz == 4100: t = int(rng.integers(2, 30))
    dt = torch.float64 if (not big or d <= 32) and case % 3 else torch.float32
    if big and d > 32: dt = torch.float32
    sc = 0.5 / np.sqrt(d)
    g = lambda *s: rng.normal(size=s)
    cp0 = np.tril(0.3 * g(bsz, d, d)) + np.eye(d); cq = np.tril(0.3 * g(bsz, t - 1, d, d)) + np.eye(d)
    ssm = mfa.StateSpaceModel(*(torch.tensor(x, dtype=dt, device="cuda:0") for x in (g(bsz, d), cp0, sc * g(bsz, t - 1, d, d), 0.3 * g(bsz, t - 1, d), cq)))
    m, c, s = ssm._moments(True)
    m2 = ssm._propagate(ssm.concatenated_state_offsets); c2, s2 = ssm._covariance_scan(True)
    tol = 1e-9 if dt == torch.float64 else 2e-3
    for a, b in ((m, m2), (c, c2), (s, s2)):
        err = float((a - b).abs().max() / (b.abs().max() + 1e-30)); worst = max(worst, err if dt == torch.float64 else 0.0)
        assert err < tol, (case, d, t, bsz, dt, err)
    q2 = mfa.StateSpaceModel(*(torch.tensor(x, dtype=dt, device="cuda:0") for x in (g(bsz, d), cp0 * 1.1, sc * g(bsz, t - 1, d, d), 0.3 * g(bsz, t - 1, d), cq * 0.9)))
    if not big:
        kl = ssm.kl_divergence(q2); ops = ssm._kl_divergence_operators(q2)
        err = float(((kl - ops).abs() / (ops.abs() + 1e-6)).max())
        assert err < (1e-8 if dt == torch.float64 else 5e-2), (case, d, t, bsz, dt, err, 'kl')
torch.cuda.synchronize(); mfa.check_errors()
print("moments / kl routes consistent over 160 random shapes; worst fp64 relative difference", worst)
