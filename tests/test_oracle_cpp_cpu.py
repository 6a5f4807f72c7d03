"""The C++ CPU restatement (oracle/nrs_cpu.cpp: the timed CPU baseline and second checker) held to
  * g2o's 36x36 known-answer system through its block Cholesky, natural and AMD ordering
    (third_party/g2o/unit_test/solver/linear_solver_test.cpp:72-85, tolerance 1e-6),
  * the NumPy oracle on small windows: identical LM accept/reject sequence, lambda / chi2 of every trial, final state,
    with the sparse Cholesky (what the reference does) and with the PCG variant,
  * the committed full-size C2 golden (tests/golden/dba_C2_trace.npz) with the PCG variant on all cores."""
import json
import os

import numpy as np
import pytest

import nrs_cpu as CPU
import nrs_oracle as O
import nrs_synth as S

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def cpu_lib():
    CPU.build()
    return CPU.load()


@pytest.mark.parametrize("ordering", [0, 1])
def test_g2o_known_answer_through_block_cholesky(cpu_lib, ordering):
    d = json.load(open(os.path.join(HERE, "golden", "g2o_sparse_system.json")))
    br, bc = [b["r"] for b in d["blocks"]], [b["c"] for b in d["blocks"]]
    bv = np.array([b["v"] for b in d["blocks"]])
    x_ref = np.array(d["x"])
    for _ in range(2):
        ok, x = CPU.block_cholesky_solve(12, br, bc, bv, np.array(d["b"]), 0.0, ordering, cpu_lib)
        assert ok and np.linalg.norm(x - x_ref) <= d["tol"] * min(np.linalg.norm(x), np.linalg.norm(x_ref))
    inv = np.array(d["inverse"])
    for k in (0, 13, 35):
        e = np.zeros(36)
        e[k] = 1
        ok, x = CPU.block_cholesky_solve(12, br, bc, bv, e, 0.0, ordering, cpu_lib)
        assert ok and np.allclose(x, inv[:, k], rtol=0, atol=1e-6 * np.abs(inv[:, k]).max())
    bv2 = bv.copy()
    bv2[3] = -bv2[3]                                    # a negative diagonal block: "not positive definite" is reported
    ok, _ = CPU.block_cholesky_solve(12, br, bc, bv2, np.array(d["b"]), 0.0, ordering, cpu_lib)
    assert not ok


def _edges(p):
    nb = p["nbr"]
    return O.dba_build(p["kf_points"], nb["rowptr"], nb["col"], nb["w"], nb["d0"], nb["status"])


@pytest.mark.parametrize("n,k,seed,model", [(120, 3, 31, S.PINHOLE), (300, 4, 32, S.PINHOLE), (250, 4, 41, S.KB8)])
def test_cpp_restatement_matches_numpy_oracle(cpu_lib, n, k, seed, model):
    p = S.make_dba_problem(n, k, seed, model)
    e = _edges(p)
    otr = []
    oq, ot, opts, nit = O.dba_solve(p["model"], p["prm"], p["poses_q"], p["poses_t"], p["lm_xyz"], p["lm_kf"], p["lm_uv"],
                                    e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"], p["scale"], 5, otr)
    for solver, tol in ((0, 1e-10), (1, 1e-7)):
        q, t, x, tr, st = CPU.dba_solve(p["model"], p["prm"], p["poses_q"], p["poses_t"], p["lm_xyz"], p["lm_kf"], p["lm_uv"],
                                        e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"], p["scale"], 5, solver, 1e-11, 2, 0, cpu_lib)
        assert st["n_iters"] == nit and [a["accepted"] for a in tr] == [b["accepted"] for b in otr]
        for a, b in zip(tr, otr):
            assert abs(a["lam"] - b["lam"]) <= max(tol, 1e-9) * b["lam"]      # lambda follows rho: (2 rho - 1)^3
            assert abs(a["chi"] - b["chi"]) <= tol * b["chi"] and abs(a["chi_new"] - b["chi_new"]) <= tol * b["chi_new"]
        assert np.allclose(q, oq, atol=tol, rtol=0) and np.allclose(t, ot, atol=10 * tol, rtol=0)
        assert np.allclose(x, opts, atol=100 * tol, rtol=0)


def test_cpp_restatement_matches_c2_golden(cpu_lib):
    """full-size C2 (91 749 landmarks): PCG variant, all cores, against the NumPy oracle's committed trace"""
    g = np.load(os.path.join(HERE, "golden", "dba_C2_trace.npz"))
    p = S.make_dba_problem("C2")
    e = _edges(p)
    assert int(g["edge_checksum"]) == S.edge_checksum(e)
    q, t, x, tr, st = CPU.dba_solve(p["model"], p["prm"], p["poses_q"], p["poses_t"], p["lm_xyz"], p["lm_kf"], p["lm_uv"],
                                    e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"], p["scale"], 5, 1, 1e-10, 0, 0, cpu_lib)
    assert st["n_iters"] == int(g["out_iters"]) and [a["accepted"] for a in tr] == g["out_accepted"].tolist()
    for a, chi, chi_new, lam in zip(tr, g["out_chi"], g["out_chi_new"], g["out_lam"]):
        assert abs(a["lam"] - lam) <= 1e-6 * lam and abs(a["chi"] - chi) <= 1e-6 * chi and abs(a["chi_new"] - chi_new) <= 1e-6 * chi_new
    assert np.allclose(q, g["out_q"], atol=1e-6, rtol=0) and np.allclose(t, g["out_t"], atol=1e-5, rtol=0)
    assert np.allclose(x[g["sel"]], g["out_pts_sel"], atol=1e-4, rtol=0)


# ---- N2b: the embedded form of the window (oracle/nrs_cpu.cpp nrs_cpu_dba_solve_embedded) against oracle/embedded_oracle.py
def _embedded(p, n_nodes):
    import embedded_oracle as E
    flag = S.pick_nodes(p["scene"]["X0"], n_nodes) if n_nodes else np.ones(p["n_points"], np.uint8)
    nb = S.node_lists(p["scene"]["X0"], p["scene"]["sigma"], flag)
    e = E.dba_build_embedded(p["kf_points"], flag, nb["rowptr"], nb["col"], nb["w"], nb["d0"], nb["status"])
    return E, e, S.embedded_window(p, e)


def test_cpp_embedded_window_with_every_point_a_node_is_the_plain_window_bit_for_bit(cpu_lib):
    p = S.make_dba_problem(90, 3, 5)
    E, e, w = _embedded(p, 0)
    assert len(e["sk_obs"]) == 0
    for solver in (0, 1):
        a = CPU.dba_solve_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"],
                                   w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"], 5, solver, 1e-11, 1, 0, cpu_lib)
        b = CPU.dba_solve(p["model"], p["prm"], p["poses_q"], p["poses_t"], p["lm_xyz"], p["lm_kf"], p["lm_uv"], e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"],
                          p["scale"], 5, solver, 1e-11, 1, 0, cpu_lib)       # (one thread: the OpenMP reductions of chi2 are order-dependent)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[4] == b[3]


@pytest.mark.parametrize("n,m,k,seed,model", [(300, 40, 4, 9, S.PINHOLE), (400, 60, 5, 14, S.KB8), (500, 70, 6, 15, S.PINHOLE)])
def test_cpp_embedded_window_matches_numpy_oracle(cpu_lib, n, m, k, seed, model):
    """identical LM decisions, lambda / chi2 of every trial, poses, node copies and skinned points: the sparse Cholesky to 1e-10, the PCG variant to 1e-7"""
    p = S.make_dba_problem(n, k, seed, model)
    E, e, w = _embedded(p, m)
    assert len(e["sk_obs"]) > 0.6 * len(p["lm_kf"])
    otr = []
    oq, ot, ox, osk, nit = E.dba_solve_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"],
                                                e["dm_idx"], e["dm_w"], w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"], 5, otr)
    for solver, tol in ((0, 1e-10), (1, 1e-7)):
        q, t, x, sk, tr, st = CPU.dba_solve_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"],
                                                     e["dm_idx"], e["dm_w"], w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"], 5,
                                                     solver, 1e-11, 2, 0, cpu_lib)
        assert st["n_iters"] == nit and [a["accepted"] for a in tr] == [b["accepted"] for b in otr]
        for a, b in zip(tr, otr):
            assert abs(a["lam"] - b["lam"]) <= max(tol, 1e-9) * b["lam"]
            assert abs(a["chi"] - b["chi"]) <= tol * b["chi"] and abs(a["chi_new"] - b["chi_new"]) <= tol * b["chi_new"]
        assert np.allclose(q, oq, atol=tol, rtol=0) and np.allclose(t, ot, atol=10 * tol, rtol=0)
        assert np.allclose(x, ox, atol=100 * tol, rtol=0) and np.allclose(sk, osk, atol=100 * tol, rtol=0)
