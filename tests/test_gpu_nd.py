"""GPU parity of the direct solver's device kernels (nr-slam_amd/csrc/nrs_engine_nd.hpp: k_nd_level / k_nd_back, multifrontal
Cholesky on the nested-dissection plan, fronts on v_mfma_f64_16x16x4) through the C ABI tap nrs_debug_nd_solve, against a dense
NumPy solve and against the host reference of the same plan (oracle/nd_host.cpp).  Stands for LinearSolverEigen::solve
(third_party/g2o/g2o/solvers/eigen/linear_solver_eigen.h:92-136).  Tolerance: 1e-10 relative to the largest solution component
(fp64 Cholesky of a well-conditioned system); bit-reproducible between calls."""
import numpy as np
import pytest

import nrs
import nrs_cpu as CPU
from test_nd_cpu import block_system, g2o_block_system

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,seed,pose", [(5, 1, True), (40, 2, True), (300, 3, True), (300, 4, False), (1200, 5, True), (2500, 6, True)])
def test_device_kernels_match_dense_and_host_reference(ctx, n, seed, pose):
    pos, last, pairs, Dn, Vp, bn, A = block_system(n, seed, pose)
    lam = 0.37
    ok, x, st, _ = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, lam)
    assert ok
    okh, xh, sth = CPU.nd_solve(pos, last, pairs, Dn, Vp, bn, lam)
    assert okh and st == sth
    scale = np.abs(xh).max()
    assert np.allclose(x, xh, rtol=0, atol=1e-11 * scale)
    if n <= 1200:
        ref = np.linalg.solve(A + lam * np.eye(len(A)), bn.ravel())
        assert np.allclose(x.ravel(), ref, rtol=0, atol=1e-10 * scale)
    ok2, x2, _, _ = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, lam)
    assert ok2 and np.array_equal(x, x2)                                   # fixed summation order: the same bits


def test_forest_and_not_positive_definite(ctx):
    pos, last, pairs, Dn, Vp, bn, A = block_system(200, 7, pose=False, disconnected=True)
    ok, x, st, _ = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, 0.0)
    assert ok and np.allclose(x.ravel(), np.linalg.solve(A, bn.ravel()), atol=1e-10)
    pos, last, pairs, Dn, Vp, bn, A = block_system(60, 9, pose=True)
    Dn[17] = -Dn[17]                                               # linear_solver_eigen.h:124-136: reported, not hidden
    ok, x, st, _ = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, 0.0)
    assert not ok


def test_g2o_known_answer_through_the_device_solver(ctx):
    """a18's reference-held vector through the direct solver's KERNELS: g2o's 36 x 36 known-answer system
    (third_party/g2o/unit_test/solver/linear_solver_test.cpp:72-85, tolerance 1e-6) -- as one front, as 20 side-by-side copies
    (a forest of fronts on several levels), and columns of its dense inverse as unit right-hand sides."""
    pos, pairs, Dn, Vp, bn, x, inv, tol = g2o_block_system()
    ok, xs, st, _ = ctx.debug_nd_solve(pos, None, pairs, Dn, Vp, bn, 0.0)
    assert ok and np.linalg.norm(xs.ravel() - x) <= tol * min(np.linalg.norm(xs), np.linalg.norm(x))
    k = 20
    posk = np.concatenate([pos + [40.0 * i, 0, 0] for i in range(k)])
    pairsk = np.concatenate([pairs + 12 * i for i in range(k)])
    ok, xs, st, _ = ctx.debug_nd_solve(posk, None, pairsk, np.tile(Dn, (k, 1, 1)), np.tile(Vp, (k, 1, 1)), np.tile(bn, (k, 1)), 0.0)
    assert ok and st["fronts"] > 4
    for row in xs.reshape(k, 36)[[0, 7, 19]]:
        assert np.linalg.norm(row - x) <= tol * np.linalg.norm(x)
    for col in (0, 13, 35):
        e = np.zeros(36); e[col] = 1.0
        ok, xs, st, _ = ctx.debug_nd_solve(pos, None, pairs, Dn, Vp, e.reshape(12, 3), 0.0)
        assert ok and np.allclose(xs.ravel(), inv[:, col], atol=1e-6 * np.abs(inv[:, col]).max())


def test_plan_cache_reuse_is_invisible():
    """the context keeps the symbolic factorisation of the last few frames (nrs_engine_nd.hpp NdCache): a frame with the same
    optimised set, edges and fixed flags as an earlier one reuses it -- same result to the last bit as the call that built it,
    also after other frames have used (and evicted) slots in between, and within the parity tolerances of a context that never
    caches (the plan of the first frame bisects that frame's positions; a later hit keeps it)"""
    import os
    import nrs
    import nrs_synth as S

    def run(c, n, seed, shift=0.0):
        tp = S.make_tracking_problem(n, seed, S.PINHOLE)
        cam = nrs.make_camera(tp["model"], tp["prm"])
        fm = np.arange(n, dtype=np.int32)
        uv = tp["uv"] + np.float32(shift)
        return c.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], uv, tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], nrs.Trace(1024))

    c = nrs.Context(direct_solve=1)
    try:
        a0 = run(c, 600, 11)
        a1 = run(c, 600, 11)                                        # hit
        for k in range(6):                                          # more distinct problems than slots
            run(c, 200 + 40 * k, 20 + k)
        a2 = run(c, 600, 11)                                        # rebuilt after eviction: the same plan again
        b0 = run(c, 600, 11, shift=0.25)                            # same structure, other values: hit, and a different answer
        for k in ("pose_q", "pose_t", "f_pos", "map_pos", "f_status"):
            assert np.array_equal(a0[k], a1[k]) and np.array_equal(a0[k], a2[k]), k
        assert a0["lost"] == a1["lost"] == a2["lost"]
        assert not np.array_equal(a0["pose_t"], b0["pose_t"])
        reused, built = c.nd_cache_stats()
        assert reused >= 2 and built >= 8                           # (two problems per frame: both of a1 and b0 hit, every first sight builds)
    finally:
        c.close()
    nrs.debug_set("NRS_ND_NO_CACHE", "1")
    try:
        c = nrs.Context(direct_solve=1)
        n0 = run(c, 600, 11)
        nb = run(c, 600, 11, shift=0.25)
        c.close()
    finally:
        nrs.debug_set("NRS_ND_NO_CACHE", None)
    for k in ("pose_q", "pose_t", "f_pos", "map_pos", "f_status"):
        assert np.array_equal(a0[k], n0[k]), k
    assert np.allclose(b0["pose_t"], nb["pose_t"], atol=1e-5, rtol=0) and np.allclose(b0["f_pos"], nb["f_pos"], atol=1e-4, rtol=0)
    assert np.array_equal(b0["f_status"], nb["f_status"])


def test_chained_and_per_level_factorisation_give_the_same_bits(ctx, monkeypatch):
    """The top of the tree in one launch (NRS_ND_CHAIN=1: fronts waiting for their children's tiles on agent-scope counters, for the
    highest levels whose workgroups are all resident at once -- every level of the small systems, the top ones of the large) against one
    launch per level (the default since the levels run on 512-thread workgroups): the same arithmetic in the same order, hence the
    same bits; 256-thread workgroups (NRS_ND_THREADS=256) likewise."""
    for n, seed in ((40, 21), (120, 22), (1200, 23), (3000, 24)):
        pos, last, pairs, Dn, Vp, bn, A = block_system(n, seed, True)
        nrs.debug_set("NRS_ND_CHAIN", "1")
        ok, x, st, _ = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, 0.21)
        nrs.debug_set("NRS_ND_LEVELS", "1")
        ok1, x1, st1, _ = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, 0.21)
        nrs.debug_set("NRS_ND_LEVELS", None)
        nrs.debug_set("NRS_ND_CHAIN", None)
        ok2, x2, st2, _ = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, 0.21)
        nrs.debug_set("NRS_ND_THREADS", "256")
        ok3, x3, st3, _ = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, 0.21)
        nrs.debug_set("NRS_ND_STEP32", "0")                       # ... and the panel by 16-column steps (the round-4 form), on 256 and 512 threads
        ok4, x4, st4, _ = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, 0.21)
        nrs.debug_set("NRS_ND_THREADS", None)
        ok5, x5, st5, _ = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, 0.21)
        nrs.debug_set("NRS_ND_STEP32", None)
        assert ok and ok1 and ok2 and ok3 and st == st2 and np.array_equal(x, x1) and np.array_equal(x, x2) and np.array_equal(x, x3), n
        assert ok4 and ok5 and np.array_equal(x, x4) and np.array_equal(x, x5), n
        nrs.debug_set("NRS_ND_BACK_FLAGS", "1")                    # ... and the back pass handing over by flags instead of by the values themselves
        ok6, x6, st6, _ = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, 0.21)
        nrs.debug_set("NRS_ND_BACK_FLAGS", None)
        assert ok6 and np.array_equal(x, x6), n


def test_a_frame_beyond_the_direct_solvers_window_is_handed_to_the_pcg(monkeypatch):
    """nrs_options.direct_solve = 0 takes the direct solver up to 8000 free rows and the PCG beyond.  (i) With the window lowered
    (NRS_ND_MAX_ROWS=300) a 600-point frame is handed over and still matches the oracle like every other a2 frame; (ii) a 10 000-point
    frame (C5-sized single frame) runs on the PCG by default (inner iterations > 1) and agrees with the same frame forced onto the
    direct solver (direct_solve = 1) within the a2 tolerances: same statuses and lost set, pose 1e-6 / 1e-5, positions 1e-4."""
    import nrs
    import nrs_oracle as O
    import nrs_synth as S
    from conftest import compare_lm_traces

    def run(c, tp, tr):
        cam = nrs.make_camera(tp["model"], tp["prm"])
        fm = np.arange(len(tp["status"]), dtype=np.int32)
        return c.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tr)

    tp = S.make_tracking_problem(600, 61)
    nrs.debug_set("NRS_ND_MAX_ROWS", "300")
    c = nrs.Context()
    try:
        tr = nrs.Trace(1024)
        r = run(c, tp, tr)
    finally:
        c.close()
        nrs.debug_set("NRS_ND_MAX_ROWS", None)
    assert max(t["inner"] for t in tr.trials) > 1                   # PCG iterations: the direct solver reports 1 per trial
    otr = []
    o = O.track_deform_solve(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], np.arange(600), tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"],
                             tp["scale"], otr)
    assert np.allclose(r["pose_q"], o["pose_q"], atol=1e-6, rtol=0) and np.allclose(r["pose_t"], o["pose_t"], atol=1e-5, rtol=0)
    assert np.array_equal(r["f_status"], o["f_status"]) and r["lost"] == o["lost"] and np.allclose(r["f_pos"], o["f_pos"], atol=1e-4, rtol=0)
    assert compare_lm_traces(tr.trials, otr, len(otr)) >= 6
    big = S.make_tracking_problem(10000, 62)
    a, b = nrs.Context(), nrs.Context(direct_solve=1)
    try:
        ta, tb = nrs.Trace(1024), nrs.Trace(1024)
        ra, rb = run(a, big, ta), run(b, big, tb)
    finally:
        a.close(); b.close()
    assert (big["status"] == 0).sum() > 8000 and max(t["inner"] for t in ta.trials) > 1 and all(t["inner"] == 1 for t in tb.trials)
    assert np.allclose(ra["pose_q"], rb["pose_q"], atol=1e-6, rtol=0) and np.allclose(ra["pose_t"], rb["pose_t"], atol=1e-5, rtol=0)
    assert np.array_equal(ra["f_status"], rb["f_status"]) and ra["lost"] == rb["lost"] and np.allclose(ra["f_pos"], rb["f_pos"], atol=1e-4, rtol=0)


def test_crowded_levels_in_two_launches_give_the_same_bits(ctx, monkeypatch):
    """A level with more workgroups than CUs runs as two launches -- diagonal / inverse workgroups (k_nd_level), then the off-diagonal
    Schur tiles from the rows of L21 those left (k_nd_tile) -- against one launch per level (NRS_ND_NO_SPLIT=1): same operands, same
    matrix-core sequence, hence the same bits (systems whose lowest levels are crowded: 2500 and 4446 points)."""
    import nrs_synth as S
    for n in (2500, 4446):
        pos, last, pairs, Dn, Vp, bn = S.nd_block_system(n)
        nrs.debug_set("NRS_ND_NO_SPLIT", None)
        ok, x, st, _ = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, 0.1)
        nrs.debug_set("NRS_ND_NO_SPLIT", "1")
        ok2, x2, st2, _ = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, 0.1)
        nrs.debug_set("NRS_ND_NO_SPLIT", None)
        assert ok and ok2 and st == st2 and st["workgroups"] > 800 and np.array_equal(x, x2), n
        okh, xh, sth = CPU.nd_solve(pos, last, pairs, Dn, Vp, bn, 0.1)
        assert okh and np.allclose(x, xh, rtol=0, atol=1e-11 * np.abs(xh).max())


def _trace_key(tr):
    return [(t["round"], t["iter"], t["trial"], t["lam"], t["chi"], t["chi_new"], t["rho"], t["accepted"], t["ok"]) for t in tr]


@pytest.mark.parametrize("n,seed,embedded", [(400, 12, False), (2500, 14, False), (1500, 42, True)])
def test_speculative_trials_walk_the_same_trials_to_the_same_bits(ctx_direct, n, seed, embedded):
    """a2's LM trials inside a run of rejections go out in batches on shadow sets of the trial state and the factor storage
    (nrs_engine_types.hpp SpecSet); one at a time (NRS_SPEC_TRIALS=0), with one or three shadow sets, with the batch sized by the
    last run or fixed at 2 / 4: the same trials (damping, chi2, gain ratio, decision -- exactly) and the same results bit for bit"""
    import nrs_synth as S
    c = ctx_direct
    tp = S.make_tracking_problem(n, seed)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    fm = np.arange(n, dtype=np.int32)
    node = None
    if embedded:
        nodes = c.skin_select_nodes(tp["X_prev"], 200, tp["status"] == 0)
        node = np.zeros(n, np.uint8)
        node[nodes] = 1

    def run():
        tr = nrs.Trace(1024)
        if embedded:
            g = nrs.RGraph(c, n, tp["graph"]["sigma"], tp["graph"]["stretch_th"])
            g.add_edges(tp["X_prev"], fm, fm)
            r = c.track_deform_solve_embedded(cam, g, tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], node, tp["pose_q"], tp["pose_t"], tp["scale"], tr, 256)
            g.close()
        else:
            r = c.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tr)
        return r, tr.trials
    outs = []
    # (the last two: three trials behind EVERY first trial of an iteration -- most are discarded and drain, the worst case for the back
    # passes' turn-taking and the abort path -- and the same without the abort word)
    for sw in ({"NRS_SPEC_TRIALS": "0"}, {}, {"NRS_SPEC_TRIALS": "1"}, {"NRS_SPEC_TRIALS": "3", "NRS_SPEC_FIXED": "4"}, {"NRS_SPEC_FIXED": "2"}, {},
               {"NRS_SPEC_TRIALS": "3", "NRS_SPEC_FIRST": "3"}, {"NRS_SPEC_TRIALS": "3", "NRS_SPEC_FIRST": "3", "NRS_SPEC_NO_ABORT": "1"}):
        nrs.debug_clear()
        for k, v in sw.items():
            nrs.debug_set(k, v)
        outs.append(run())
    nrs.debug_clear()
    r0, t0 = outs[0]
    assert any(t["trial"] >= 2 for t in t0), "the frame has no run of rejections: nothing speculative ran"
    for r, t in outs[1:]:
        assert _trace_key(t) == _trace_key(t0)
        for k in ("pose_q", "pose_t", "f_pos", "f_status", "map_pos"):
            assert np.array_equal(r[k], r0[k]), k
        assert r["lost"] == r0["lost"] and r["median"] == r0["median"]


def test_symbolic_phase_on_the_helper_thread_or_inline_gives_the_same_bits(ctx_direct):
    """the plan is built on the context's helper thread next to the packing (PlanWorker) or, with NRS_HOST_THREADS=1, inline: same plan,
    same results; several frames in a row, so that the worker is reused"""
    import nrs_synth as S
    c = ctx_direct
    outs = {}
    for mode in ("worker", "inline", "worker again"):
        nrs.debug_clear()
        nrs.debug_set("NRS_ND_NO_CACHE", "1")
        if mode == "inline":
            nrs.debug_set("NRS_HOST_THREADS", "1")
        res = []
        for n, seed in ((300, 7), (900, 8), (300, 7)):
            tp = S.make_tracking_problem(n, seed)
            cam = nrs.make_camera(tp["model"], tp["prm"])
            fm = np.arange(n, dtype=np.int32)
            tr = nrs.Trace(1024)
            r = c.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tr)
            res.append((r, _trace_key(tr.trials)))
        outs[mode] = res
    nrs.debug_clear()
    for mode in ("inline", "worker again"):
        for (r, t), (r0, t0) in zip(outs[mode], outs["worker"]):
            assert t == t0
            for k in ("pose_q", "pose_t", "f_pos", "f_status", "map_pos"):
                assert np.array_equal(r[k], r0[k]), (mode, k)
