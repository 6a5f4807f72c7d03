"""GPU parity of the embedded-deformation mode (N2 as SURVEY.md 8d words it; include/nrs.h nrs_track_deform_solve_embedded) against
oracle/embedded_oracle.py on oracle/rgraph_oracle.DenseGraph -- the generalisation of CameraPoseAndDeformationOptimization
(g2o_optimization.cc:148-557) in which only the NODES carry vertices and every other optimised landmark is skinned to <= 11 of them.
The reference has no such mode (its only skinning is OPT:476-553); what pins it: with every landmark a node the call must return the
bits of nrs_track_deform_solve_rg (and the oracle those of nrs_oracle.track_deform_solve, tests/test_oracle_embedded_cpu.py).
Tolerances of the a2 tests: statuses, lost set, graph statuses exact; pose 1e-6 / 1e-5; positions 1e-4; LM traces to the noise floor."""
import numpy as np
import pytest

import embedded_oracle as E
import nrs
import nrs_synth as S
import rgraph_oracle as RG

pytestmark = pytest.mark.gpu


def _graphs(ctx, tp, n):
    ids = np.arange(n, dtype=np.int32)
    g = nrs.RGraph(ctx, n, tp["graph"]["sigma"], tp["graph"]["stretch_th"])
    D = RG.DenseGraph(n, tp["graph"]["sigma"], tp["graph"]["stretch_th"])
    g.add_edges(tp["X_prev"], ids, ids)
    D.add_edges(tp["X_prev"], ids, ids)
    return g, D, ids


def test_all_nodes_is_the_parity_solve_bit_for_bit(ctx_direct):
    n = 500
    tp = S.make_tracking_problem(n, 31)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    g, _, ids = _graphs(ctx_direct, tp, n)
    g2, _, _ = _graphs(ctx_direct, tp, n)
    ta, tb = nrs.Trace(1024), nrs.Trace(1024)
    a = ctx_direct.track_deform_solve_embedded(cam, g, tp["X_prev"], ids, tp["status"], tp["uv"], tp["X_prev"], np.ones(n, np.uint8), tp["pose_q"], tp["pose_t"], tp["scale"], ta, 64)
    b = ctx_direct.track_deform_solve_rg(cam, g2, tp["X_prev"], ids, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tb, 64)
    for k in ("pose_q", "pose_t", "f_pos", "f_status", "map_pos"):
        assert np.array_equal(a[k], b[k]), k
    assert a["lost"] == b["lost"] and a["median"] == b["median"]
    assert [(t["lam"], t["chi"], t["chi_new"], t["accepted"]) for t in ta.trials] == [(t["lam"], t["chi"], t["chi_new"], t["accepted"]) for t in tb.trials]
    g.close(); g2.close()


@pytest.mark.parametrize("n,m,seed,model", [(600, 80, 41, S.PINHOLE), (1500, 200, 42, S.PINHOLE), (900, 120, 43, S.KB8)])
def test_embedded_mode_matches_its_oracle(ctx, n, m, seed, model):
    from conftest import compare_lm_traces
    tp = S.make_tracking_problem(n, seed, model)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    g, D, ids = _graphs(ctx, tp, n)
    nodes = ctx.skin_select_nodes(tp["X_prev"], m, tp["status"] == 0)
    node = np.zeros(n, np.uint8)
    node[nodes] = 1
    tr, otr = nrs.Trace(1024), []
    r = ctx.track_deform_solve_embedded(cam, g, tp["X_prev"], ids, tp["status"], tp["uv"], tp["X_prev"], node, tp["pose_q"], tp["pose_t"], tp["scale"], tr, 256)
    o = E.track_deform_solve_embedded(tp["model"], tp["prm"], D, tp["X_prev"], ids, tp["status"], tp["uv"], tp["X_prev"], node, tp["pose_q"], tp["pose_t"],
                                      tp["scale"], otr)
    assert o["n_nodes"] == m and o["n_skinned"] > (n - m) // 2
    assert np.allclose(r["pose_q"], o["pose_q"], atol=1e-6, rtol=0) and np.allclose(r["pose_t"], o["pose_t"], atol=1e-5, rtol=0)
    assert np.array_equal(r["f_status"], o["f_status"]) and r["lost"] == o["lost"]
    assert np.allclose(r["f_pos"], o["f_pos"], atol=1e-4, rtol=0) and np.allclose(r["map_pos"], o["map_pos"], atol=1e-4, rtol=0)
    assert abs(r["median"] - o["median"]) < 1e-5
    assert compare_lm_traces(tr.trials, otr, len(otr)) >= 6
    probe = np.arange(0, n, max(1, n // 25), dtype=np.int32)
    assert np.array_equal(g.rows(probe)[3], D.st[probe])
    g.close()


def test_reused_plan_takes_the_new_frames_skinning_weights():
    """a frame whose structure (optimised set, nodes, which nodes every point is skinned to) an earlier frame had reuses that frame's
    plan and observation lists (nrs_engine_nd.hpp NdStruct) -- with its OWN skinning weights: the same bits as a context that
    builds everything for it"""
    import os
    n, m = 700, 90
    tp = S.make_tracking_problem(n, 47, S.PINHOLE)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    ids = np.arange(n, dtype=np.int32)

    def run(c, sigma, node):
        g = nrs.RGraph(c, n, sigma, tp["graph"]["stretch_th"])
        g.add_edges(tp["X_prev"], ids, ids)
        r = c.track_deform_solve_embedded(cam, g, tp["X_prev"], ids, tp["status"], tp["uv"], tp["X_prev"], node, tp["pose_q"], tp["pose_t"], tp["scale"], nrs.Trace(1024), 256)
        g.close()
        return r

    sig = tp["graph"]["sigma"]
    c = nrs.Context()
    try:
        nodes = c.skin_select_nodes(tp["X_prev"], m, tp["status"] == 0)
        node = np.zeros(n, np.uint8)
        node[nodes] = 1
        a = run(c, sig, node)
        h0 = c.nd_cache_stats()[0]
        b = run(c, sig * 1.02, node)                                # the same neighbours in the same order, other weights
        assert c.nd_cache_stats()[0] > h0, "the second frame was expected to reuse the first one's plan"
    finally:
        c.close()
    nrs.debug_set("NRS_ND_NO_CACHE", "1")
    try:
        c = nrs.Context()
        fresh = run(c, sig * 1.02, node)
        c.close()
    finally:
        nrs.debug_set("NRS_ND_NO_CACHE", None)
    assert not np.array_equal(a["f_pos"], b["f_pos"])
    for k in ("pose_q", "pose_t", "f_pos", "map_pos", "f_status"):
        assert np.array_equal(b[k], fresh[k]), k
    assert b["lost"] == fresh["lost"] and b["median"] == fresh["median"]


@pytest.mark.parametrize("n,m,seed,model", [(600, 80, 41, S.PINHOLE), (900, 120, 43, S.KB8)])
def test_embedded_mode_on_the_pcg_matches_its_oracle(ctx_pcg, n, m, seed, model):
    """nrs_options.direct_solve = 2 (or a frame beyond the direct solver's window): the embedded mode runs on the PCG, the skinned
    observations applied as hyper-edges (k_skin_op / k_skin_op_rows, the operator form of the embedded BA window) -- same oracle,
    same tolerances as on the direct solver."""
    from conftest import compare_lm_traces
    tp = S.make_tracking_problem(n, seed, model)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    g, D, ids = _graphs(ctx_pcg, tp, n)
    nodes = ctx_pcg.skin_select_nodes(tp["X_prev"], m, tp["status"] == 0)
    node = np.zeros(n, np.uint8)
    node[nodes] = 1
    tr, otr = nrs.Trace(1024), []
    r = ctx_pcg.track_deform_solve_embedded(cam, g, tp["X_prev"], ids, tp["status"], tp["uv"], tp["X_prev"], node, tp["pose_q"], tp["pose_t"], tp["scale"], tr, 256)
    o = E.track_deform_solve_embedded(tp["model"], tp["prm"], D, tp["X_prev"], ids, tp["status"], tp["uv"], tp["X_prev"], node, tp["pose_q"], tp["pose_t"],
                                      tp["scale"], otr)
    assert max(t["inner"] for t in tr.trials) > 1                   # PCG iterations (the direct solver reports 1 per trial)
    assert np.allclose(r["pose_q"], o["pose_q"], atol=1e-6, rtol=0) and np.allclose(r["pose_t"], o["pose_t"], atol=1e-5, rtol=0)
    assert np.array_equal(r["f_status"], o["f_status"]) and r["lost"] == o["lost"]
    assert np.allclose(r["f_pos"], o["f_pos"], atol=1e-4, rtol=0) and np.allclose(r["map_pos"], o["map_pos"], atol=1e-4, rtol=0)
    assert abs(r["median"] - o["median"]) < 1e-5
    assert compare_lm_traces(tr.trials, otr, len(otr)) >= 6
    g.close()


@pytest.mark.parametrize("n,m,seed,cap", [(700, 0, 51, 64), (2500, 0, 52, 128), (1500, 200, 53, 256), (900, 0, 54, 8)])
def test_device_walk_builds_the_same_problem_as_the_host_walk(ctx_direct, monkeypatch, n, m, seed, cap):
    """The neighbour walk of OPT:252-279 runs on the device for the dense graph (nrs_rgraph.hip k_rg_walk: the sequential loop's sets as the
    fixed point of a parallel pass, the lists never leave the device) -- against the host loop over the downloaded lists
    (NRS_HOST_WALK=1): the same edges in the same order, hence the same bits everywhere: parity mode (m = 0: every point a node),
    embedded mode (m nodes), and a prefix so short (cap 8) that walks run off their lists and the driver has to fetch longer ones."""
    tp = S.make_tracking_problem(n, seed)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    node = np.ones(n, np.uint8)
    if m:
        node[:] = 0
        node[ctx_direct.skin_select_nodes(tp["X_prev"], m, tp["status"] == 0)] = 1
    out = []
    for host in (False, True):
        if host:
            nrs.debug_set("NRS_HOST_WALK", "1")
        else:
            nrs.debug_set("NRS_HOST_WALK", None)
        g, _, ids = _graphs(ctx_direct, tp, n)
        tr = nrs.Trace(1024)
        r = ctx_direct.track_deform_solve_embedded(cam, g, tp["X_prev"], ids, tp["status"], tp["uv"], tp["X_prev"], node, tp["pose_q"], tp["pose_t"], tp["scale"], tr, cap)
        st = g.edge_statuses(ids[:200], ids[:200]) if hasattr(g, "edge_statuses") else None
        out.append((r, tr.trials, st))
        g.close()
    nrs.debug_set("NRS_HOST_WALK", None)
    if cap == 64:                                     # the passes run out (capped at 2): the driver falls back to the host walk, same result
        nrs.debug_set("NRS_WALK_MAX_PASSES", "2")
        g, _, ids = _graphs(ctx_direct, tp, n)
        tr = nrs.Trace(1024)
        r = ctx_direct.track_deform_solve_embedded(cam, g, tp["X_prev"], ids, tp["status"], tp["uv"], tp["X_prev"], node, tp["pose_q"], tp["pose_t"], tp["scale"], tr, cap)
        g.close()
        nrs.debug_set("NRS_WALK_MAX_PASSES", None)
        for k in ("pose_q", "pose_t", "f_pos", "f_status", "map_pos"):
            assert np.array_equal(r[k], out[0][0][k]), k
    (a, ta, sa), (b, tb, sb) = out
    for k in ("pose_q", "pose_t", "f_pos", "f_status", "map_pos"):
        assert np.array_equal(a[k], b[k]), k
    assert a["lost"] == b["lost"] and a["median"] == b["median"] and len(ta) > 10
    assert [(t["lam"], t["chi"], t["chi_new"], t["accepted"]) for t in ta] == [(t["lam"], t["chi"], t["chi_new"], t["accepted"]) for t in tb]
