"""
GPU parity tests of the block-tridiagonal operator (through the C ABI) against the numpy oracle and the
golden fixtures.  They re-express /root/reference/tests/unit/test_block_tri_diag.py:29-225.
Tolerances: fp64 rtol 1e-9 (kernel vs oracle on the same inputs), fp32 rtol 2e-3 on well-conditioned inputs.
"""
import numpy as np
import pytest
import torch

import markovflow_amd as mfa
from oracle import numpy_oracle as O
from conftest import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def tt(x, dtype=torch.float64):
    return None if x is None else torch.tensor(np.ascontiguousarray(x), dtype=dtype, device=DEV)


def nn(x):
    return x.detach().cpu().numpy().astype(np.float64)


def random_spd_btd(rng, batch, n, d, has_sub, well=True):
    """SPD block tridiagonal from a random lower factor (generator of test_block_tri_diag.py:274-295)."""
    ldiag = np.tril(rng.normal(loc=1.0, size=batch + (n, d, d)))
    lsub = rng.normal(size=batch + (n - 1, d, d)) if has_sub else None
    if well:
        idx = np.arange(d)
        dg = 1.0 + np.abs(ldiag[..., idx, idx])
        ldiag = 0.3 * (ldiag - 1.0)
        ldiag[..., idx, idx] = dg
        ldiag = np.tril(ldiag)
        lsub = None if lsub is None else 0.3 * lsub
    lower = O.btd_to_dense(ldiag, lsub, symmetric=False)
    dense = lower @ np.swapaxes(lower, -1, -2)
    diag = np.stack([dense[..., i * d:(i + 1) * d, i * d:(i + 1) * d] for i in range(n)], axis=-3)
    sub = (np.stack([dense[..., (i + 1) * d:(i + 2) * d, i * d:(i + 1) * d] for i in range(n - 1)], axis=-3)
           if has_sub else None)
    return diag, sub, dense


BATCHES = [(3,), (), (2, 1)]
TOL = {torch.float64: dict(rtol=1e-9, atol=1e-11), torch.float32: dict(rtol=2e-3, atol=2e-4)}


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("batch", BATCHES)
@pytest.mark.parametrize("d,n,has_sub", [(1, 1, False), (1, 4, True), (3, 1, False), (3, 4, True), (3, 4, False),
                                         (2, 7, True), (4, 33, True), (5, 5, True), (6, 64, True), (7, 3, True),
                                         (8, 9, True), (9, 17, True)])
def test_cholesky_solve_logdet_mult_inverse(rng, dtype, batch, d, n, has_sub):
    diag, sub, dense = random_spd_btd(rng, batch, n, d, has_sub)
    rhs = rng.normal(size=batch + (n, d))
    tol = TOL[dtype]
    sym = mfa.SymmetricBlockTriDiagonal(tt(diag, dtype), tt(sub, dtype))
    chol = sym.cholesky
    ld, ls = O.btd_cholesky(diag, sub)
    np.testing.assert_allclose(nn(chol.block_diagonal), np.tril(ld), **tol)
    if has_sub:
        np.testing.assert_allclose(nn(chol.block_sub_diagonal), ls, **tol)
    # to_dense == np.linalg.cholesky(dense)   (test_block_tri_diag.py:94-107)
    np.testing.assert_allclose(nn(chol.to_dense()), np.linalg.cholesky(dense), rtol=max(tol["rtol"], 1e-7), atol=1e-7 if dtype == torch.float64 else 2e-3)
    np.testing.assert_allclose(nn(sym.to_dense()), dense, rtol=1e-6 if dtype == torch.float32 else 1e-12, atol=1e-6 if dtype == torch.float32 else 0)
    np.testing.assert_allclose(nn(chol.abs_log_det()), O.btd_abs_log_det(ld), **tol)
    r = tt(rhs, dtype)
    np.testing.assert_allclose(nn(chol.solve(r)), O.btd_solve(ld, ls, rhs), **tol)
    np.testing.assert_allclose(nn(chol.solve(r, transpose_left=True)), O.btd_solve(ld, ls, rhs, transpose_left=True), **tol)
    np.testing.assert_allclose(nn(sym.dense_mult(r)), O.btd_dense_mult(diag, sub, rhs, symmetric=True), **tol)
    np.testing.assert_allclose(nn(chol.dense_mult(r)), O.btd_dense_mult(ld, ls, rhs, symmetric=False), **tol)
    np.testing.assert_allclose(nn(chol.dense_mult(r, transpose_left=True)),
                               O.btd_dense_mult(ld, ls, rhs, symmetric=False, transpose_left=True), **tol)
    inv_d, inv_s = O.btd_block_diagonal_of_inverse(ld, ls, return_sub=True)
    np.testing.assert_allclose(nn(chol.block_diagonal_of_inverse()), inv_d, **tol)
    got_d, got_s = chol._diag_and_sub_of_inverse(want_sub=True)
    if has_sub:
        np.testing.assert_allclose(nn(got_s), inv_s, **tol)


@pytest.mark.parametrize("name", ["btd_d1_T1_sub0", "btd_d1_T4_sub1", "btd_d3_T1_sub0", "btd_d3_T4_sub0",
                                  "btd_d3_T4_sub1", "btd_d6_T64_sub1", "btd_d9_T64_sub1"])
def test_golden_fixtures(name):
    g = golden(name + ".npz")
    sub = g["sub"] if bool(g["has_sub"]) else None
    sym = mfa.SymmetricBlockTriDiagonal(tt(g["diag"]), tt(sub))
    chol = sym.cholesky
    tol = dict(rtol=2e-6, atol=1e-7)   # the fixture tolerance of tests/test_oracle_golden.py
    np.testing.assert_allclose(nn(chol.block_diagonal), g["chol_diag"], **tol)
    if sub is not None:
        np.testing.assert_allclose(nn(chol.block_sub_diagonal), g["chol_sub"], **tol)
    np.testing.assert_allclose(nn(chol.abs_log_det()), 0.5 * g["logdet"], rtol=2e-6)
    r = tt(g["rhs"])
    np.testing.assert_allclose(nn(chol.solve(r)), g["solve_l"], **tol)
    np.testing.assert_allclose(nn(chol.solve(r, transpose_left=True)), g["solve_lt"], **tol)
    np.testing.assert_allclose(nn(sym.dense_mult(r)), g["mult_sym"], **tol)
    np.testing.assert_allclose(nn(chol.dense_mult(r)), g["mult_l"], **tol)
    np.testing.assert_allclose(nn(chol.dense_mult(r, transpose_left=True)), g["mult_lt"], **tol)
    np.testing.assert_allclose(nn(chol.block_diagonal_of_inverse()), g["inv_diag"], **tol)


@pytest.mark.parametrize("batch", BATCHES)
@pytest.mark.parametrize("d,n", [(1, 3), (3, 3), (3, 5), (6, 40)])
def test_upper_diagonal_lower(rng, batch, d, n):
    # test_block_tri_diag.py:205-225
    diag, sub, dense = random_spd_btd(rng, batch, n, d, True)
    lower_t, diag_t = mfa.SymmetricBlockTriDiagonal(tt(diag), tt(sub)).upper_diagonal_lower()
    lower, dd = nn(lower_t.to_dense()), nn(diag_t.to_dense())
    chol_d_u = np.swapaxes(dd, -1, -2) @ lower
    np.testing.assert_allclose(lower, np.tril(lower))
    assert diag_t.block_sub_diagonal is None
    np.testing.assert_allclose(dense, np.swapaxes(chol_d_u, -1, -2) @ chol_d_u, rtol=1e-6, atol=1e-9)
    u_t, chol_d = O.btd_upper_diagonal_lower(diag, sub)
    np.testing.assert_allclose(nn(lower_t.block_sub_diagonal), u_t, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(nn(diag_t.block_diagonal), chol_d, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("has_sub_1", [True, False])
@pytest.mark.parametrize("has_sub_2", [True, False])
def test_add_and_sub_diag(rng, has_sub_1, has_sub_2):
    # test_block_tri_diag.py:29-76
    d1, s1, dense1 = random_spd_btd(rng, (3,), 4, 3, has_sub_1)
    d2, s2, dense2 = random_spd_btd(rng, (3,), 4, 3, has_sub_2)
    added = mfa.SymmetricBlockTriDiagonal(tt(d1), tt(s1)) + mfa.SymmetricBlockTriDiagonal(tt(d2), tt(s2))
    np.testing.assert_allclose(nn(added.to_dense()), dense1 + dense2, rtol=1e-12)
    low = mfa.LowerTriangularBlockTriDiagonal(tt(d1), tt(rng.normal(size=(3, 3, 3, 3))))
    assert low.block_sub_diagonal.shape == (3, 3, 3, 3)


def test_solve_rhs_broadcast_and_leading_dims(rng):
    # block_tri_diag.py:261-287 and the sample_shape + batch_shape use of state_space_model.py:307-322
    diag, sub, _ = random_spd_btd(rng, (2, 3), 6, 3, True)
    chol = mfa.SymmetricBlockTriDiagonal(tt(diag), tt(sub)).cholesky
    ld, ls = O.btd_cholesky(diag, sub)
    rhs = rng.normal(size=(5, 2, 3, 6, 3))
    np.testing.assert_allclose(nn(chol.solve(tt(rhs))), O.btd_solve(ld, ls, rhs), rtol=1e-9, atol=1e-11)
    rhs1 = rng.normal(size=(1, 3, 6, 3))
    np.testing.assert_allclose(nn(chol.solve(tt(rhs1))), O.btd_solve(ld, ls, rhs1), rtol=1e-9, atol=1e-11)
    with pytest.raises(ValueError):
        chol.solve(tt(rng.normal(size=(2, 3, 5, 3))))


def test_logdet_quad_partitioned_matches_natural_order(rng):
    """Fused scalar form: 1/2 |L^-1 r|^2 - log|L| from the partitioned elimination, long chain."""
    import ctypes
    from markovflow_amd import _lib
    for d, n in [(6, 1000), (3, 9), (4, 65)]:
        diag, sub, _ = random_spd_btd(rng, (5,), n, d, True)
        rhs = rng.normal(size=(5, n, d))
        ld, ls = O.btd_cholesky(diag, sub)
        ref = 0.5 * np.sum(O.btd_solve(ld, ls, rhs) ** 2, axis=(-1, -2)) - O.btd_abs_log_det(ld)
        dg, sb, rh = tt(diag), tt(sub), tt(rhs)
        lib = _lib.load()
        wsb = int(lib.mf_btd_logdet_quad_workspace_bytes(5, n, d, 8))
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        out = torch.empty(5, dtype=torch.float64, device=DEV)
        _lib.call("mf_btd_logdet_quad", torch.float64, 5, n, d, _lib.ptr(dg), _lib.ptr(sb), _lib.ptr(rh),
                  _lib.ptr(out), _lib.ptr(ws), wsb, None, _lib.stream_ptr(torch.device(DEV)))
        np.testing.assert_allclose(nn(out), ref, rtol=1e-10)


def test_non_positive_definite_sets_info(rng):
    from markovflow_amd import _lib
    diag = -np.tile(np.eye(3), (2, 4, 1, 1))
    dg = tt(diag)
    ld = torch.empty_like(dg)
    info = _lib.new_info(torch.device(DEV))
    _lib.call("mf_btd_cholesky", torch.float64, 2, 4, 3, _lib.ptr(dg), None, _lib.ptr(ld), None, None, 0,
              _lib.ptr(info), _lib.stream_ptr(torch.device(DEV)))
    # every block fails: the word names the first one (flat index 0; LAPACK info = 1)
    assert int(info.item()) > 0 and int(_lib.load().mf_info_flat_index(int(info.item()))) == 0


def test_non_positive_pivot_is_reported_like_the_reference(rng, monkeypatch):
    """TensorFlow's Cholesky op raises on a matrix that is not positive definite (block_tri_diag.py:423-436).  Here the kernels
    raise a flag in pinned host memory: by default it is looked at without synchronising (next library call, or
    check_errors()); in synchronous mode (MF_CHECK_PIVOTS=1) the factorising call itself raises."""
    from markovflow_amd import _lib
    _lib.check_errors()                                                   # start clean
    sym = mfa.SymmetricBlockTriDiagonal(tt(-np.tile(np.eye(3), (2, 4, 1, 1))))
    assert not torch.isfinite(sym.cholesky.block_diagonal).all()          # the result itself is NaN
    with pytest.raises(_lib.MarkovflowAmdError, match="SymmetricBlockTriDiagonal.cholesky"):
        _lib.check_errors()
    _lib.check_errors()                                                   # reported once, then clean again
    sym.cholesky
    torch.cuda.synchronize()
    good = mfa.SymmetricBlockTriDiagonal(tt(np.tile(np.eye(3), (2, 4, 1, 1))))
    with pytest.raises(_lib.MarkovflowAmdError):                          # the next library call notices the finished failure
        good.cholesky
    assert torch.isfinite(good.cholesky.block_diagonal).all()
    _lib.check_errors()
    monkeypatch.setattr(_lib, "CHECK_PIVOTS", True)
    with pytest.raises(_lib.MarkovflowAmdError):
        sym.cholesky
    good.cholesky                                                         # and a clean call stays clean
    monkeypatch.setattr(_lib, "CHECK_PIVOTS", False)
    with mfa.errors_as_nan():                                             # opt-out: NaN results, nothing raised, nothing left behind
        assert not torch.isfinite(sym.cholesky.block_diagonal).all()
        good.cholesky
    _lib.check_errors()


def test_a_failure_survives_later_library_calls_until_it_is_looked_at(rng):
    """failing cholesky, then a NON-factorising call (solve) while the first kernel may still be running, then
    check_errors(): the failure must be raised and must name the cholesky.  (A clean unsynchronised look used to forget the
    names on record, and check_errors() then skipped the flag.)"""
    from markovflow_amd import _lib
    _lib.check_errors()
    n = 4000                                                             # long enough to still be running at the next call
    diag = np.tile(np.eye(3), (2, n, 1, 1))
    diag[:, -1] = -np.eye(3)                                              # the LAST block fails: the flag is raised late
    sym = mfa.SymmetricBlockTriDiagonal(tt(diag))
    chol = sym.cholesky
    try:
        chol.solve(tt(np.ones((2, n, 3))))                                # may or may not see the flag already
        with pytest.raises(_lib.MarkovflowAmdError, match="SymmetricBlockTriDiagonal.cholesky"):
            _lib.check_errors()
    except _lib.MarkovflowAmdError as e:                                  # the kernel had already finished: raised by solve's look
        assert "SymmetricBlockTriDiagonal.cholesky" in str(e)
    _lib.check_errors()


def test_a_bad_log_likelihood_raises_when_the_host_reads_it(rng):
    """TensorFlow raises inside the Cholesky (block_tri_diag.py:423-436).  `float(kf.log_likelihood())` on a model whose
    posterior precision is not positive definite raises here too - at the host read, which is the synchronisation the
    caller performs anyway - without check_errors() and without MF_CHECK_PIVOTS."""
    from markovflow_amd import _lib
    from markovflow_amd import synthetic
    torch.cuda.synchronize()
    _lib.check_errors()
    inp = synthetic.make_ssm(3, 40, (3, 3), dtype=torch.float64, device=DEV)
    good = float(synthetic.kalman_filter_from(inp).log_likelihood())
    assert np.isfinite(good)
    inp["cholQ"][1, 17] = 0.0                                             # a singular process covariance in one series
    kf = synthetic.kalman_filter_from(inp)
    # no synchronisation is added: the failure surfaces at the host read of the result at the latest - or already at the
    # evaluation's second library call when the first kernel has finished by then (a 3 x 40 problem takes microseconds)
    with pytest.raises(_lib.MarkovflowAmdError, match="log_likelihood"):
        float(kf.log_likelihood())
    with mfa.errors_as_nan():                                             # opt-out: NaN, nothing raised, nothing left behind
        assert not np.isfinite(float(kf.log_likelihood()))
    _lib.check_errors()


def test_unsupported_state_dim_fails_loudly(rng):
    diag = np.tile(np.eye(40), (1, 3, 1, 1))        # fp64: register kernels to d = 9, LDS-tile kernels to d = 32
    with pytest.raises(NotImplementedError):
        mfa.SymmetricBlockTriDiagonal(tt(diag)).cholesky
    diag = np.tile(np.eye(65), (1, 3, 1, 1))        # fp32: to d = 64
    with pytest.raises(NotImplementedError):
        mfa.SymmetricBlockTriDiagonal(tt(diag, torch.float32)).cholesky
