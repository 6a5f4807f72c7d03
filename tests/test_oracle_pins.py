"""Pins the oracle against the only reference-owned vectors on this path:
g2o's 36x36 known-answer SPD system and the Huber-derivative property test."""
import json
import os

import numpy as np
import scipy.sparse as sp

import nrs_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def _g2o_system():
    d = json.load(open(os.path.join(HERE, "golden", "g2o_sparse_system.json")))
    A = np.zeros((36, 36))
    for blk in d["blocks"]:
        r, c, v = blk["r"], blk["c"], np.array(blk["v"])
        A[3 * r:3 * r + 3, 3 * c:3 * c + 3] = v
        if r != c:                      # g2o stores the upper block triangle only
            A[3 * c:3 * c + 3, 3 * r:3 * r + 3] = v.T
    return A, np.array(d["b"]), np.array(d["x"]), d["tol"]


def test_g2o_known_answer_dense_path():
    # reference: third_party/g2o/unit_test/solver/linear_solver_test.cpp:72-85 (isApprox 1e-6)
    A, b, x, tol = _g2o_system()
    ok, xs = O.solve_spd(sp.csc_matrix(A), b, np.zeros(36))
    assert ok
    assert np.linalg.norm(xs - x) <= tol * min(np.linalg.norm(xs), np.linalg.norm(x))


def test_g2o_known_answer_sparse_path(monkeypatch):
    A, b, x, tol = _g2o_system()
    # force the sparse-LU branch that large problems take
    big = sp.block_diag([sp.csc_matrix(A)] * 130, format="csc")
    ok, xs = O.solve_spd(big, np.tile(b, 130), np.zeros(36 * 130))
    assert ok
    xs = xs.reshape(130, 36)
    for row in xs[[0, 57, 129]]:
        assert np.linalg.norm(row - x) <= tol * np.linalg.norm(x)


def test_not_positive_definite_is_reported():
    A = np.diag([1.0, -1.0, 2.0])
    ok, xs = O.solve_spd(sp.csc_matrix(A), np.ones(3), np.full(3, 7.0))
    assert not ok and np.all(xs == 7.0)       # stale x is handed back, as g2o does


def test_huber_derivative_property():
    # reference: third_party/g2o/unit_test/general/robust_kernel_tests.cpp:104-119
    delta = 1.3
    for frac in (0.5, 0.99, 1.5):
        e = (frac * delta) ** 2
        h = 1e-6
        num = (O.huber(e + h, delta)[0] - O.huber(e - h, delta)[0]) / (2 * h)
        assert abs(num - O.huber(e, delta)[1]) < 1e-5
    # continuity at the threshold
    e = delta * delta
    assert abs(O.huber(e, delta)[0] - O.huber(e * (1 + 1e-12), delta)[0]) < 1e-9


def test_constants_are_float_arithmetic():
    # g2o_optimization.cc:63-64,197-210: float sqrt / float reciprocal, widened to double
    assert float(O.TH2) == float(np.float32(np.sqrt(np.float32(5.99))))
    assert float(O.TH2) ** 2 != 5.99
    assert float(O.INFO_REPROJ) == 4.0
    assert abs(float(O.INFO_POSITION) - 100.0) < 1e-4
    assert abs(O.info_spatial(0.05) - 1.0 / (0.005 ** 2)) / 4e4 < 1e-6
