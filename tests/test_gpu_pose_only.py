"""GPU parity: a1 CameraPoseOptimization (reference g2o_optimization.cc:50-146) through the C ABI
against the oracle on identical seeded inputs.

Tolerances: final rotation 1e-6 (quaternion components), translation 1e-5 map
units, per-trial chi2 1e-6 relative, identical accept/reject sequence while consecutive chi2
values are separated by more than the fp32-projection noise floor, identical inlier mask."""
import numpy as np
import pytest

import nrs
import nrs_oracle as O
import nrs_synth as S

pytestmark = pytest.mark.gpu


def _problem(n, seed, model=S.PINHOLE):
    tp = S.make_tracking_problem(n, seed, model)
    m = tp["status"] == 0
    return tp, tp["uv"][m], tp["X_prev"][m]


def _compare(ctx, tp, uv, X):
    cam = nrs.make_camera(tp["model"], tp["prm"])
    tr = nrs.Trace(1024)
    q, t, inl = ctx.pose_only_solve(cam, uv, X, tp["pose_q"], tp["pose_t"], tr)
    otr = []
    q2, t2, inl2 = O.pose_only_solve(tp["model"], tp["prm"], uv, X, tp["pose_q"], tp["pose_t"], otr)
    return (q, t, inl, tr.trials), (q2, t2, inl2, [dict(r, round=i) for i, rnd in enumerate(otr) for r in rnd])


@pytest.mark.parametrize("n,seed", [(60, 1), (500, 2), (5000, 3)])
def test_pose_only_matches_oracle_pinhole(ctx, n, seed):
    tp, uv, X = _problem(n, seed)
    (q, t, inl, tr), (q2, t2, inl2, otr) = _compare(ctx, tp, uv, X)
    assert np.allclose(q, q2, atol=1e-6, rtol=0)
    assert np.allclose(t, t2, atol=1e-5, rtol=0)
    assert np.array_equal(inl, inl2)
    # every trial of every round until the oracle's own decision sits on the fp32 noise floor (tests/conftest.py)
    from conftest import compare_lm_traces
    assert compare_lm_traces(tr, otr, 3) >= 9


def test_pose_only_kb8(ctx):
    tp, uv, X = _problem(800, 4, S.KB8)
    (q, t, inl, tr), (q2, t2, inl2, otr) = _compare(ctx, tp, uv, X)
    # atan2f / sinf / cosf are defined on both sides as the double routine rounded to float,
    # so KB8 holds the pinhole bar: identical inlier mask
    assert np.allclose(q, q2, atol=1e-6, rtol=0)
    assert np.allclose(t, t2, atol=1e-5, rtol=0)
    assert np.array_equal(inl, inl2)


def test_pose_only_edge_cases(ctx):
    tp, uv, X = _problem(200, 6)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    # empty input: nothing to optimise, pose returned unchanged (g2o: optimize() on an empty graph)
    q, t, inl = ctx.pose_only_solve(cam, uv[:0], X[:0], tp["pose_q"], tp["pose_t"])
    assert np.allclose(q, O.quat_normalize(tp["pose_q"])) and np.allclose(t, tp["pose_t"]) and len(inl) == 0
    # all observations gross outliers -> every edge ends at level 1
    uv_bad = uv + 300.0
    q, t, inl = ctx.pose_only_solve(cam, uv_bad, X, tp["pose_q"], tp["pose_t"])
    q2, t2, inl2 = O.pose_only_solve(tp["model"], tp["prm"], uv_bad, X, tp["pose_q"], tp["pose_t"])
    assert np.array_equal(inl, inl2)
    # bad arguments are reported, not crashed on
    with pytest.raises(nrs.NrsError):
        ctx.pose_only_solve(nrs.make_camera(7, tp["prm"]), uv, X, tp["pose_q"], tp["pose_t"])


def test_pose_only_deterministic(ctx):
    tp, uv, X = _problem(3000, 8)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    a = ctx.pose_only_solve(cam, uv, X, tp["pose_q"], tp["pose_t"])
    b = ctx.pose_only_solve(cam, uv, X, tp["pose_q"], tp["pose_t"])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("n,seed,model", [(500, 2, S.PINHOLE), (5000, 3, S.PINHOLE), (800, 4, S.KB8)])
def test_pose_only_many_workgroups_matches_oracle(ctx, monkeypatch, n, seed, model):
    """the multi-workgroup form (k_po_pass / k_po_step: what frames of >= 32768 points take) forced onto the frames the single-workgroup
    kernel is held to the oracle on: same tolerances, identical inlier mask, traces to the noise floor; bit-reproducible between calls"""
    from conftest import compare_lm_traces
    nrs.debug_set("NRS_PO_MULTI_MIN", "1")
    tp, uv, X = _problem(n, seed, model)
    (q, t, inl, tr), (q2, t2, inl2, otr) = _compare(ctx, tp, uv, X)
    assert np.allclose(q, q2, atol=1e-6, rtol=0) and np.allclose(t, t2, atol=1e-5, rtol=0)
    assert np.array_equal(inl, inl2)
    assert compare_lm_traces(tr, otr, 3) >= 9
    cam = nrs.make_camera(tp["model"], tp["prm"])
    b = ctx.pose_only_solve(cam, uv, X, tp["pose_q"], tp["pose_t"])
    assert np.array_equal(q, b[0]) and np.array_equal(t, b[1]) and np.array_equal(inl, b[2])


def test_pose_only_on_a_100k_point_frame(ctx, monkeypatch):
    """a C5-sized frame (100 000 map points): the default path is the multi-workgroup form; it agrees with the single-workgroup kernel on
    the same frame (pose 1e-9 / 1e-8: only the order of the sums differs; inlier masks equal up to points whose chi2 sits within 1e-6 of the
    gate) and is several times faster"""
    import time
    tp, uv, X = _problem(100000, 9)
    assert len(uv) > 80000
    cam = nrs.make_camera(tp["model"], tp["prm"])

    def run():
        best, r = 1e9, None
        for _ in range(3):
            t0 = time.perf_counter()
            r = ctx.pose_only_solve(cam, uv, X, tp["pose_q"], tp["pose_t"])
            best = min(best, time.perf_counter() - t0)
        return r, best
    (q, t, inl), t_multi = run()
    nrs.debug_set("NRS_PO_MULTI_MIN", "1000000000")
    (q1, t1, inl1), t_single = run()
    print("a1 at %d points: %.2f ms on many workgroups, %.2f ms on one" % (len(uv), 1e3 * t_multi, 1e3 * t_single))
    assert np.allclose(q, q1, atol=1e-9, rtol=0) and np.allclose(t, t1, atol=1e-8, rtol=0)
    assert (inl != inl1).sum() <= 2
    assert t_multi < 0.6 * t_single
