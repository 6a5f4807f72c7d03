"""CPU tests of the direct solver's symbolic phase (nr-slam_amd/csrc/nrs_nd_plan.hpp: nested dissection, fronts, child maps,
entry lists -- host logic of the product) through its host reference solve (oracle/nd_host.cpp) against a dense NumPy solve.
What the solve stands for: LinearSolverEigen::solve, third_party/g2o/g2o/solvers/eigen/linear_solver_eigen.h:92-136."""
import numpy as np
import pytest

import nrs_cpu as CPU


def block_system(n, seed, pose=True, knn=8, disconnected=False):
    """An SPD block system with a2's structure: n points on a surface, each coupled to its nearest neighbours (3x3 blocks),
    optionally two `last` blocks (a pose) coupled to every point."""
    rng = np.random.default_rng(seed)
    pts = np.c_[rng.uniform(-20, 20, n), rng.uniform(-15, 15, n), 60 + rng.normal(0, 1, n)]
    if disconnected:
        pts[: n // 2, 0] -= 200.0
    d = np.linalg.norm(pts[:, None, :2] - pts[None, :, :2], axis=2)
    np.fill_diagonal(d, 1e9)
    pairs = set()
    for i in range(n):
        for j in np.argsort(d[i])[:knn]:
            if d[i, j] < 50:
                pairs.add((min(i, int(j)), max(i, int(j))))
    pairs = np.array(sorted(pairs), np.int32).reshape(-1, 2)
    flip = rng.uniform(size=len(pairs)) < 0.5                      # either node may come first in a pair
    pairs[flip] = pairs[flip][:, ::-1]
    nn = n + (2 if pose else 0)
    if pose:
        pp = np.array([(n + h, i) if (i + h) % 2 else (i, n + h) for i in range(n) for h in range(2)] + [(n, n + 1)], np.int32)
        pairs = np.r_[pairs, pp]
    pos = np.r_[pts, np.zeros((nn - n, 3))]
    last = np.zeros(nn, np.uint8)
    last[n:] = 1
    Vp = rng.normal(0, 1, (len(pairs), 3, 3)) * 0.3
    A = np.zeros((3 * nn, 3 * nn))
    for (a, b), V in zip(pairs, Vp):
        A[3 * a:3 * a + 3, 3 * b:3 * b + 3] += V
        A[3 * b:3 * b + 3, 3 * a:3 * a + 3] += V.T
    rowsum = np.abs(A).sum(1)
    Dn = np.zeros((nn, 3, 3))
    for i in range(nn):
        S = rng.normal(0, 0.2, (3, 3))
        Dn[i] = S @ S.T + np.eye(3) * (rowsum[3 * i:3 * i + 3].max() + 0.5)
        A[3 * i:3 * i + 3, 3 * i:3 * i + 3] = Dn[i]
    bn = rng.normal(0, 1, (nn, 3))
    return pos, last, pairs, Dn, Vp, bn, A


@pytest.mark.parametrize("n,seed,pose", [(5, 1, True), (40, 2, True), (300, 3, True), (300, 4, False), (1200, 5, True)])
def test_host_reference_matches_dense(n, seed, pose):
    pos, last, pairs, Dn, Vp, bn, A = block_system(n, seed, pose)
    lam = 0.37
    ok, x, st = CPU.nd_solve(pos, last, pairs, Dn, Vp, bn, lam)
    assert ok
    ref = np.linalg.solve(A + lam * np.eye(len(A)), bn.ravel())
    assert np.allclose(x.ravel(), ref, rtol=0, atol=1e-10 * np.abs(ref).max())
    assert st["max_s"] <= 96 and st["fronts"] >= 1
    if n >= 300:
        assert st["levels"] >= 4 and st["fronts"] > 8


def test_disconnected_graph_and_single_front():
    pos, last, pairs, Dn, Vp, bn, A = block_system(200, 7, pose=False, disconnected=True)     # a forest: several roots
    ok, x, st = CPU.nd_solve(pos, last, pairs, Dn, Vp, bn, 0.0)
    assert ok and np.allclose(x.ravel(), np.linalg.solve(A, bn.ravel()), atol=1e-10)
    pos, last, pairs, Dn, Vp, bn, A = block_system(3, 8, pose=True)
    ok, x, st = CPU.nd_solve(pos, last, pairs, Dn, Vp, bn, 0.0)
    assert ok and st["fronts"] == 1 and np.allclose(x.ravel(), np.linalg.solve(A, bn.ravel()), atol=1e-12)


def test_not_positive_definite_is_reported():
    pos, last, pairs, Dn, Vp, bn, A = block_system(60, 9, pose=True)
    Dn[17] = -Dn[17]                                               # linear_solver_eigen.h:124-136: the solve reports failure
    ok, x, st = CPU.nd_solve(pos, last, pairs, Dn, Vp, bn, 0.0)
    assert not ok


def g2o_block_system():
    """g2o's own solver fixture (third_party/g2o/unit_test/solver/sparse_system_helper.cpp:52-149,255,298; extracted numbers in
    tests/golden/g2o_sparse_system.json) as the direct solver's input: 12 blocks of 3, the upper block triangle as pairs."""
    import json, os
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g2o_sparse_system.json")))
    Dn, pairs, Vp = np.zeros((12, 3, 3)), [], []
    for blk in d["blocks"]:
        r, c, v = blk["r"], blk["c"], np.array(blk["v"], float)
        if r == c:
            Dn[r] = v
        else:
            pairs.append((r, c)); Vp.append(v)                     # rows = r's components
    pos = np.c_[np.arange(12.0), np.zeros(12), np.zeros(12)]        # (geometry only steers the dissection)
    return pos, np.array(pairs, np.int32), Dn, np.array(Vp), np.array(d["b"]).reshape(12, 3), np.array(d["x"]), np.array(d["inverse"]), d["tol"]


def test_g2o_known_answer_through_the_plan_and_host_reference():
    # reference: third_party/g2o/unit_test/solver/linear_solver_test.cpp:72-85 (isApprox 1e-6); leaf / chunk sizes of 32 nodes make this
    # ONE front -- the multi-front path on the same numbers: 20 copies of the system side by side as one block-diagonal problem
    pos, pairs, Dn, Vp, bn, x, inv, tol = g2o_block_system()
    ok, xs, st = CPU.nd_solve(pos, None, pairs, Dn, Vp, bn, 0.0)
    assert ok and np.linalg.norm(xs.ravel() - x) <= tol * min(np.linalg.norm(xs), np.linalg.norm(x))
    k = 20
    posk = np.concatenate([pos + [40.0 * i, 0, 0] for i in range(k)])
    pairsk = np.concatenate([pairs + 12 * i for i in range(k)])
    ok, xs, st = CPU.nd_solve(posk, None, pairsk, np.tile(Dn, (k, 1, 1)), np.tile(Vp, (k, 1, 1)), np.tile(bn, (k, 1)), 0.0)
    assert ok and st["fronts"] > 4
    for row in xs.reshape(k, 36)[[0, 7, 19]]:
        assert np.linalg.norm(row - x) <= tol * np.linalg.norm(x)
    # a column of g2o's dense inverse as a unit right-hand side
    e = np.zeros((12, 3)); e[4, 1] = 1.0
    ok, xs, st = CPU.nd_solve(pos, None, pairs, Dn, Vp, e, 0.0)
    assert ok and np.allclose(xs.ravel(), inv[:, 13], atol=1e-6 * np.abs(inv[:, 13]).max())


@pytest.mark.parametrize("n,seed,pose,knn", [(7, 1, True, 3), (90, 2, True, 11), (700, 3, True, 11), (700, 4, False, 22), (2500, 5, True, 16)])
def test_plan_invariants(n, seed, pose, knn):
    """what the device kernels take for granted about a plan (oracle/nd_host.cpp nrs_cpu_nd_plan_check): one owner per node, the
    separators separate (both ends of every coupling in the earlier node's front), boundaries sorted and inside the parent's front,
    the boundary's owner segments tile it parent-first with proper ancestors only, front sizes, the workgroup lists"""
    pos, last, pairs, Dn, Vp, bn, A = block_system(n, seed, pose, knn=knn)
    assert CPU.nd_plan_check(pos, last, pairs) == 0
    pos, last, pairs, Dn, Vp, bn, A = block_system(n, seed, pose, knn=knn, disconnected=True)     # a forest (without a pose): several roots
    assert CPU.nd_plan_check(pos, last, pairs) == 0


@pytest.mark.parametrize("n,seed,pose,knn,shape", [
    (700, 3, True, 11, dict(fronts=46, levels=7, max_s=96, max_b=132, L_doubles=415368, U_doubles=311424, flops=26193024, workgroups=214)),
    (2500, 5, True, 16, dict(fronts=131, levels=11, max_s=96, max_b=354, L_doubles=2142594, U_doubles=3868500, flops=305143704, workgroups=1448)),
    (300, 4, False, 8, dict(fronts=19, levels=5, max_s=93, max_b=66, L_doubles=151587, U_doubles=40336, flops=5480208, workgroups=52)),
])
def test_plan_shape_is_pinned(n, seed, pose, knn, shape):
    """the dissection is deterministic (coordinate bisection by (coordinate, index), minimum vertex cover by Kuhn's paths in a fixed
    order): the shape of the plan of a seeded system is a known answer -- a change of the builder that is meant to leave the plans
    alone (round 4: no sort per level, one pass for both separator candidates) must reproduce it"""
    pos, last, pairs, Dn, Vp, bn, A = block_system(n, seed, pose, knn=knn)
    ok, x, st = CPU.nd_solve(pos, last, pairs, Dn, Vp, bn, 0.1)
    assert ok and st == shape
