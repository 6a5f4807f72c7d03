"""GPU parity: a2 CameraPoseAndDeformationOptimization (reference g2o_optimization.cc:148-557) and the
RegularizationGraph operations a19/a20 (reference regularization_graph.cc:61-146) through the C ABI
against the oracle.

Tolerances: neighbour lists, statuses, lost sets, edge statuses: exact (integer / index work);
graph weights and max/min distances: exact (fp32, same operation sequence); pose 1e-6 / 1e-5;
positions 1e-4 map units (fp32 outputs)."""
import numpy as np
import pytest

import nrs
import nrs_oracle as O
import nrs_synth as S

pytestmark = pytest.mark.gpu


def _perturbed_graph(n, seed):
    sc = S.make_scene(n, 2, seed)
    G = sc["graph"]
    rng = np.random.default_rng(seed)
    G["e_status"][rng.uniform(size=len(G["e_status"])) < 0.15] = S.GRAPH_BAD
    G["e_status"][rng.uniform(size=len(G["e_status"])) < 0.05] = 0          # VERIFIED
    G["e_w"][rng.uniform(size=len(G["e_w"])) < 0.2] *= np.float32(0.35)
    return sc, G


@pytest.mark.parametrize("n,seed", [(50, 1), (700, 2), (4000, 3)])
def test_select_neighbours_bit_exact(ctx, n, seed):
    sc, G = _perturbed_graph(n, seed)
    rp, col, eid = ctx.graph_select_neighbours(G)
    ref = S.ordered_neighbours(G)                       # vectorised twin ...
    assert np.array_equal(rp, ref["o_rowptr"]) and np.array_equal(col, ref["o_col"]) and np.array_equal(eid, ref["o_eid"])
    for p in (0, n // 3, n - 1):                        # ... and the oracle's per-row GetEdges
        lst = O.graph_get_edges(G, p)
        assert [c for c, _ in lst] == col[rp[p]:rp[p + 1]].tolist()
        assert [e for _, e in lst] == eid[rp[p]:rp[p + 1]].tolist()


def test_select_neighbours_ties_and_empty(ctx):
    # equal weights: ties are broken by index (documented choice, SURVEY.md 7.2 hard part 4)
    g = dict(n=4, rowptr=np.array([0, 3, 4, 5, 6], np.int32), col=np.array([1, 2, 3, 0, 0, 0], np.int32),
             eid=np.array([0, 1, 2, 0, 1, 2], np.int32), e_w=np.array([0.9, 0.9, 0.9], np.float32),
             e_d0=np.ones(3, np.float32), e_max=np.ones(3, np.float32), e_min=np.ones(3, np.float32),
             e_status=np.array([2, 2, 2], np.int32), sigma=1.0, stretch_th=1.1, min_w=0.3)
    rp, col, eid = ctx.graph_select_neighbours(g)
    assert col[:3].tolist() == [1, 2, 3]
    g["e_w"] = np.array([0.9, 0.1, 0.95], np.float32)      # the cut stops at the first low weight
    rp, col, eid = ctx.graph_select_neighbours(g)
    assert col[rp[0]:rp[1]].tolist() == [3, 1]
    empty = dict(g, rowptr=np.zeros(5, np.int32), col=np.zeros(0, np.int32), eid=np.zeros(0, np.int32))
    rp, col, eid = ctx.graph_select_neighbours(empty)
    assert rp.tolist() == [0, 0, 0, 0, 0] and len(col) == 0
    bad = dict(g, col=np.array([3, 2, 1, 0, 0, 0], np.int32))   # rows must be index-ordered
    with pytest.raises(nrs.NrsError):
        ctx.graph_select_neighbours(bad)


@pytest.mark.parametrize("n,seed", [(300, 5), (3000, 6)])
def test_graph_update_bit_exact(ctx, n, seed):
    sc, G = _perturbed_graph(n, seed)
    rng = np.random.default_rng(seed)
    pos = sc["X0"] + rng.normal(0, 0.02, sc["X0"].shape).astype(np.float32)
    ids = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.int32)
    g2, good = ctx.graph_update(G, pos, ids)
    ref = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in G.items()}
    good_ref = [O.graph_update_vertex_flat(ref, int(p), pos) for p in ids]
    assert np.array_equal(good, np.array(good_ref))
    for k in ("e_status", "e_max", "e_min", "e_w"):
        assert np.array_equal(g2[k], ref[k]), k
    assert (g2["e_status"] == S.GRAPH_BAD).sum() > (G["e_status"] == S.GRAPH_BAD).sum()


def _track_compare(ctx, n, seed, model=S.PINHOLE, **kw):
    tp = S.make_tracking_problem(n, seed, model, **kw)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    fm = np.arange(n, dtype=np.int32)
    tr = nrs.Trace(1024)
    r = ctx.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"],
                               tp["pose_q"], tp["pose_t"], tp["scale"], tr)
    otr = []
    o = O.track_deform_solve(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"],
                             tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], otr)
    return tp, r, o, tr.trials, otr


@pytest.mark.parametrize("n,seed,solver", [(150, 11, "direct"), (150, 11, "pcg"), (400, 12, "direct"), (400, 12, "pcg"), (400, 12, "pcg-coarse"),
                                           (900, 13, "direct"), (900, 13, "pcg"), (900, 13, "pcg-coarse")])
def test_track_deform_matches_oracle(ctx_direct, ctx_pcg, monkeypatch, n, seed, solver):
    # BOTH linear solvers are held to the oracle: the nested-dissection Cholesky (nrs_options.direct_solve = 1; what frames of
    # 48..2600 free rows run by default) and the PCG (direct_solve = 2) -- the latter with block-Jacobi alone, as frames below
    # ~1.5k rows run it, and with the two-level preconditioner larger frames add (NRS_COARSE_MIN_TILES=0 puts a small one on it)
    if solver == "pcg-coarse":
        nrs.debug_set("NRS_COARSE_MIN_TILES", "0")
    else:
        nrs.debug_set("NRS_COARSE_MIN_TILES", None)
    tp, r, o, tr, otr = _track_compare(ctx_direct if solver == "direct" else ctx_pcg, n, seed)
    assert all(t["inner"] == 1 for t in tr) == (solver == "direct")          # (the direct path reports one "iteration" per trial)
    assert np.allclose(r["pose_q"], o["pose_q"], atol=1e-6, rtol=0)
    assert np.allclose(r["pose_t"], o["pose_t"], atol=1e-5, rtol=0)
    assert np.array_equal(r["f_status"], o["f_status"])
    assert r["lost"] == o["lost"] and len(r["lost"]) > 0
    assert np.allclose(r["f_pos"], o["f_pos"], atol=1e-4, rtol=0)
    assert np.allclose(r["map_pos"], o["map_pos"], atol=1e-4, rtol=0)
    assert abs(r["median"] - o["median"]) < 1e-5
    assert np.array_equal(r["graph"]["e_status"], o["graph"]["e_status"])
    assert np.allclose(r["graph"]["e_w"], o["graph"]["e_w"], atol=1e-5)
    # every trial of every round until the oracle's own decision sits on the fp32 noise floor (tests/conftest.py)
    from conftest import compare_lm_traces
    assert compare_lm_traces(tr, otr, 3) >= 9


@pytest.mark.parametrize("solver", ["direct", "pcg"])
def test_track_deform_kb8(ctx_direct, ctx_pcg, solver):
    tp, r, o, tr, otr = _track_compare(ctx_direct if solver == "direct" else ctx_pcg, 400, 21, S.KB8)
    assert np.allclose(r["pose_q"], o["pose_q"], atol=1e-6, rtol=0)
    assert np.allclose(r["pose_t"], o["pose_t"], atol=1e-5, rtol=0)
    # KB8 trigonometry is defined on both sides as the double routine rounded to float: the
    # classification is then as exact as the pinhole one
    assert np.array_equal(r["f_status"], o["f_status"])
    assert r["lost"] == o["lost"]
    assert np.allclose(r["f_pos"], o["f_pos"], atol=1e-4, rtol=0)
    assert np.allclose(r["map_pos"], o["map_pos"], atol=1e-4, rtol=0)
    assert np.array_equal(r["graph"]["e_status"], o["graph"]["e_status"])


def test_track_deform_no_lost_points(ctx):
    # neighbours that are JUST_TRIANGULATED are skipped without becoming "lost" (OPT:266-270):
    # stage 2 must not run
    tp = S.make_tracking_problem(200, 31, lost_frac=0.0)
    tp["status"][tp["status"] != 0] = 2
    cam = nrs.make_camera(tp["model"], tp["prm"])
    fm = np.arange(200, dtype=np.int32)
    tr = nrs.Trace(1024)
    r = ctx.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"],
                               tp["pose_q"], tp["pose_t"], tp["scale"], tr)
    o = O.track_deform_solve(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"],
                             tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"])
    assert r["lost"] == [] == o["lost"]
    assert max(x["round"] for x in tr.trials) == 1
    assert np.array_equal(r["f_status"], o["f_status"])
    assert np.allclose(r["pose_q"], o["pose_q"], atol=1e-6, rtol=0)


def test_track_deform_empty_frame(ctx):
    tp = S.make_tracking_problem(100, 41)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    st = np.full(100, 3, np.int32)                      # nothing TRACKED_WITH_3D
    r = ctx.track_deform_solve(cam, tp["graph"], tp["X_prev"], np.arange(100), st, tp["uv"], tp["X_prev"],
                               tp["pose_q"], tp["pose_t"], tp["scale"])
    assert r["lost"] == [] and np.array_equal(r["f_status"], st)
    assert np.allclose(r["pose_q"], tp["pose_q"]) and np.allclose(r["pose_t"], tp["pose_t"])
