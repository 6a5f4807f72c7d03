"""a18 held to the reference's own known-answer vector, on the device.

g2o's linear-solver test (third_party/g2o/unit_test/solver/linear_solver_test.cpp:72-85) hands every solver a
36 x 36 block-sparse SPD matrix and checks x to isApprox(1e-6); matrix, b, x and the dense inverse are literals
of unit_test/solver/sparse_system_helper.cpp:52-149,151-194,255,298 (fixture: tests/golden/g2o_sparse_system.json).
The matrix is an arrow: blocks 1..6 couple to block 11 only, blocks 7..10 to block 0 only, 0 and 11 to each other.
Taking (0, 11) as one 6-dof block makes it exactly the shape of the engine's normal equations without
regularisers -- H_pp (6x6), ten 3x3 diagonal blocks, ten 6x3 couplings -- so the product's PCG kernels
(k_trial_setup with inv3_sym / inv6_spd, k_spmv, k_pcg_update) solve the reference's system itself
(include/nrs.h nrs_debug_pcg_solve)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
POSE_BLOCKS = (0, 11)
ROW_BLOCKS = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10)


def _system():
    d = json.load(open(os.path.join(HERE, "golden", "g2o_sparse_system.json")))
    A = np.zeros((36, 36))
    for blk in d["blocks"]:
        r, c, v = blk["r"], blk["c"], np.array(blk["v"])
        A[3 * r:3 * r + 3, 3 * c:3 * c + 3] = v
        if r != c:
            A[3 * c:3 * c + 3, 3 * r:3 * r + 3] = v.T
    return A, np.array(d["b"]), np.array(d["x"]), np.array(d["inverse"]), d["tol"]


def _perm():
    idx = []
    for blk in POSE_BLOCKS + ROW_BLOCKS:
        idx += [3 * blk, 3 * blk + 1, 3 * blk + 2]
    return np.array(idx)


def _pack(A, b):
    """36x36 in the permuted order [pose(6), rows(30)] -> the tap's arrays; asserts the arrow shape."""
    p = _perm()
    Ap, bp_ = A[np.ix_(p, p)], b[p]
    rows = Ap[6:, 6:].copy()
    for i in range(10):
        rows[3 * i:3 * i + 3, 3 * i:3 * i + 3] = 0
    assert not rows.any(), "rows must couple through the 6-dof block only"
    Hpp = Ap[:6, :6]
    Hpp21 = np.array([Hpp[i, j] for i in range(6) for j in range(i, 6)])
    D6 = np.array([[Ap[6 + 3 * i + a, 6 + 3 * i + c] for a in range(3) for c in range(a, 3)] for i in range(10)])
    Hpl18 = np.array([[Ap[pp, 6 + 3 * i + c] for pp in range(6) for c in range(3)] for i in range(10)])
    return Hpp21, bp_[:6], D6, Hpl18, bp_[6:].reshape(10, 3), p


def _unperm(xp, p):
    x = np.zeros(36)
    x[p] = xp
    return x


def _is_approx(a, b, tol):          # Eigen isApprox
    return np.linalg.norm(a - b) <= tol * min(np.linalg.norm(a), np.linalg.norm(b))


def test_g2o_known_answer_through_device_pcg(ctx):
    A, b, x_ref, _, tol = _system()
    Hpp21, bp, D6, Hpl18, bl, p = _pack(A, b)
    for _ in range(2):                                   # the reference solves twice (pattern reuse)
        xs, iters = ctx.debug_pcg_solve(Hpp21, bp, D6, Hpl18, bl, 0.0)
        assert _is_approx(_unperm(xs, p), x_ref, tol), (iters, np.abs(_unperm(xs, p) - x_ref).max())
        assert 0 < iters <= 60


def test_g2o_dense_inverse_columns_through_device_pcg(ctx):
    """columns of the reference's dense inverse (sparse_system_helper.cpp:151-194) = solutions for unit right-hand sides"""
    A, _, _, inv, _ = _system()
    for k in (0, 4, 8, 17, 23, 30, 35):
        e = np.zeros(36)
        e[k] = 1.0
        Hpp21, bp, D6, Hpl18, bl, p = _pack(A, e)
        xs, _ = ctx.debug_pcg_solve(Hpp21, bp, D6, Hpl18, bl, 0.0)
        assert np.allclose(_unperm(xs, p), inv[:, k], rtol=0, atol=1e-6 * np.abs(inv[:, k]).max())


def test_block_inverses_are_exact(ctx):
    """without couplings the block-Jacobi preconditioner IS the inverse: inv3_sym / inv6_spd are exact iff PCG
    needs one step (the second launch only sees the converged residual)"""
    A, b, _, _, _ = _system()
    Hpp21, bp, D6, Hpl18, bl, p = _pack(A, b)
    xs, iters = ctx.debug_pcg_solve(Hpp21, bp, D6, 0 * Hpl18, bl, 0.0)
    assert iters <= 2
    Hpp = np.zeros((6, 6))
    Hpp[np.triu_indices(6)] = Hpp21
    Hpp = Hpp + Hpp.T - np.diag(np.diag(Hpp))
    assert np.allclose(xs[:6], np.linalg.solve(Hpp, bp), rtol=1e-12, atol=0)
    for i in range(10):
        Di = np.zeros((3, 3))
        Di[np.triu_indices(3)] = D6[i]
        Di = Di + Di.T - np.diag(np.diag(Di))
        assert np.allclose(xs[6 + 3 * i:9 + 3 * i], np.linalg.solve(Di, bl[i]), rtol=1e-12, atol=1e-18)


def test_damped_system(ctx):
    """(H + lambda I) x = b as an LM trial poses it"""
    A, b, _, _, _ = _system()
    lam = 1e-5 * np.abs(np.diag(A)).max()
    Hpp21, bp, D6, Hpl18, bl, p = _pack(A, b)
    xs, _ = ctx.debug_pcg_solve(Hpp21, bp, D6, Hpl18, bl, lam)
    ref = np.linalg.solve(A + lam * np.eye(36), b)
    assert _is_approx(_unperm(xs, p), ref, 1e-8)


def test_not_positive_definite_is_reported(ctx):
    import nrs
    A, b, _, _, _ = _system()
    Hpp21, bp, D6, Hpl18, bl, p = _pack(A, b)
    D6 = D6.copy()
    D6[3] = -D6[3]
    with pytest.raises(nrs.NrsError):
        ctx.debug_pcg_solve(Hpp21, bp, D6, Hpl18, bl, 0.0)
