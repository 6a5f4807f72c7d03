"""GPU parity: RegularizationGraph at the reference's density (a19 / a20): the device-resident dense
graph (include/nrs.h nrs_rgraph_*) against oracle/rgraph_oracle.py -- all-pairs initialisation (map.cc:148-166),
UpdateVertex over N - 1 connections per point, GetEdges on dense rows, graph growth (mapping.cc:240-256).
Everything here is integer / index / fp32-with-fixed-operation-order work: exact."""
import numpy as np
import pytest

import nrs
import nrs_oracle as O
import rgraph_oracle as RG

pytestmark = pytest.mark.gpu


def _scene(n, seed):
    rng = np.random.default_rng(seed)
    side = np.sqrt(n / 5000.0)
    pos = np.stack([rng.uniform(-22 * side, 22 * side, n), rng.uniform(-17 * side, 17 * side, n), 60 + rng.normal(0, 1.0, n)], 1).astype(np.float32)
    return rng, pos


def _compare_rows(g, D, ids):
    mx, mn, d0, st = g.rows(ids)
    assert np.array_equal(st, D.st[ids])
    ex = D.st[ids] != RG.NONE
    assert np.array_equal(mx[ex], D.maxd[ids][ex]) and np.array_equal(mn[ex], D.mind[ids][ex]) and np.array_equal(d0[ex], D.d0[ids][ex])


def _compare_get_edges(g, D, ids, cap=256):
    cnt, col, w, d0, st = g.get_edges(ids, cap)
    for r, i in enumerate(ids):
        js, ow, od0, ost = D.get_edges(int(i))
        assert cnt[r] == len(js)
        m = min(cnt[r], cap)                                 # the first cap entries of the list are returned
        assert np.array_equal(col[r, :m], js[:m]) and np.array_equal(st[r, :m], ost[:m])
        assert np.array_equal(w[r, :m], ow[:m]) and np.array_equal(d0[r, :m], od0[:m])
    return cnt


@pytest.mark.parametrize("n,seed", [(700, 1), (2000, 2), (5000, 3)])
def test_dense_graph_matches_oracle(ctx, n, seed):
    rng, pos = _scene(n, seed)
    sigma, th = 2.2, 1.1
    n0 = n - n // 10                                     # the last tenth joins later (graph growth)
    g = nrs.RGraph(ctx, n, sigma, th)
    D = RG.DenseGraph(n, sigma, th)
    assert g.min_weight() == float(D.min_w)
    first = np.arange(n0, dtype=np.int32)
    g.add_edges(pos, first, first)                       # every pair: deg = n0 - 1
    D.add_edges(pos, first, first)
    probe = np.sort(rng.choice(n0, 40, replace=False)).astype(np.int32)
    _compare_rows(g, D, probe)
    sel = np.arange(n0, dtype=np.int32) if n <= 2000 else np.sort(rng.choice(n0, 1200, replace=False)).astype(np.int32)
    cnt = _compare_get_edges(g, D, sel)
    assert cnt.min() > 3 and (D.st[probe[0]] != RG.NONE).sum() == n0 - 1
    # two tracked frames: most points move a little, a patch stretches (its edges go BAD)
    for step in range(2):
        pos2 = pos + rng.normal(0, 0.02, pos.shape).astype(np.float32)
        patch = np.linalg.norm(pos2[:, :2] - pos2[step, :2], axis=1) < 6
        pos2[patch, :2] = pos2[step, :2] + (pos2[patch, :2] - pos2[step, :2]) * np.float32(1.8 + step)
        ids = np.sort(rng.choice(n0, int(0.9 * n0), replace=False)).astype(np.int32)
        good = g.update(pos2, ids)
        ref = np.array([D.update_vertex(pos2, int(i)) for i in ids])
        assert np.array_equal(good, ref)
        assert good.max() > n0 // 2                      # far connections count as good: what the "< 5" rule sees (OPT:468-473)
        _compare_rows(g, D, probe)
        _compare_get_edges(g, D, sel)
        pos = pos2
    assert (D.st[probe] == O.GRAPH_BAD).any()
    # graph growth: new landmarks against every current one
    new, cur = np.arange(n0, n, dtype=np.int32), np.arange(n, dtype=np.int32)
    g.add_edges(pos, new, cur)
    D.add_edges(pos, new, cur)
    both = np.concatenate([probe[:10], new[:10]]).astype(np.int32)
    _compare_rows(g, D, both)
    _compare_get_edges(g, D, both)
    e = g.edge(int(new[0]), int(probe[0]))
    assert e["status"] == O.GRAPH_NEUTRAL and e["d0"] == float(D.d0[new[0], probe[0]])
    g.close()


def test_break_semantics_and_errors(ctx):
    g = nrs.RGraph(ctx, 4, 1.0, 1.1)
    pos = np.array([[0, 0, 0], [0.5, 0, 0], [0.6, 0, 0], [9, 0, 0]], np.float32)
    g.add_edges(pos, np.arange(4), np.arange(4))
    # stretch edge (0,1) far beyond the threshold, then bring it back: BAD with a heavy-ish weight
    p2 = pos.copy()
    p2[1, 0] = 1.4
    assert g.update(p2, [1]).tolist() == [1]             # (1,0) and (1,2) stretch, (1,3) does not
    cnt, col, w, d0, st = g.get_edges([0], 8)
    assert col[0, :cnt[0]].tolist() == [2]               # the far NEUTRAL edge (0,3) cuts the list before the BAD one
    assert g.edge(0, 1)["status"] == O.GRAPH_BAD and g.edge(0, 0 + 3)["status"] == O.GRAPH_NEUTRAL
    with pytest.raises(nrs.NrsError):
        g.get_edges([0, 7], 8)
    with pytest.raises(nrs.NrsError):
        g.add_edges(pos, [0], [4])
    g.close()
    big = nrs.RGraph(ctx, 300, 50.0, 1.1)                # everything within 1.5 sigma: more neighbours than the caller's cap
    rng, p = _scene(300, 9)
    big.add_edges(p, np.arange(300), np.arange(300))
    D = RG.DenseGraph(300, 50.0, 1.1)
    D.add_edges(p, np.arange(300), np.arange(300))
    for cap in (16, 512):                                    # truncated prefix / the whole list (rank-sort path)
        cnt = _compare_get_edges(big, D, np.array([0, 7, 299], np.int32), cap)
        assert cnt[0] == 299
    big.close()
    # a sigma that covers the map on a larger graph: the extraction path (> 384 survivors per row)
    rng, p = _scene(1500, 10)
    wide = nrs.RGraph(ctx, 1500, 80.0, 1.1)
    Dw = RG.DenseGraph(1500, 80.0, 1.1)
    ids = np.arange(1500, dtype=np.int32)
    wide.add_edges(p, ids, ids)
    Dw.add_edges(p, ids, ids)
    p2 = p.copy()
    p2[:200, :2] *= np.float32(3.0)                          # some BAD edges: two status classes in the lists
    wide.update(p2, ids[:400])
    for i in ids[:400]:
        Dw.update_vertex(p2, int(i))
    cnt = _compare_get_edges(wide, Dw, np.array([0, 5, 250, 777, 1499], np.int32), 64)
    assert cnt.max() > 1000
    wide.close()
    # 1400 connections of one weight (coincident points): the selecting pass overflows its staging area on purpose and
    # every survivor of the row is staged instead; ties come out by index
    rng, p = _scene(1600, 11)
    p[100:1500] = np.float32([3.0, 1.0, 60.0])
    p[0] = np.float32([0.0, 0.0, 60.0])
    eq = nrs.RGraph(ctx, 1600, 40.0, 1.1)
    De = RG.DenseGraph(1600, 40.0, 1.1)
    ids = np.arange(1600, dtype=np.int32)
    eq.add_edges(p, ids, ids)
    De.add_edges(p, ids, ids)
    _compare_get_edges(eq, De, np.array([0, 99, 100, 1599], np.int32), 32)
    eq.close()


@pytest.mark.parametrize("n,seed,model", [(500, 21, 0), (900, 22, 1)])
def test_track_deform_on_the_dense_graph(ctx, n, seed, model):
    """a2 (CameraPoseAndDeformationOptimization, OPT:148-557) with GetEdges / UpdateVertex served from the device-resident
    all-pairs graph (nrs_track_deform_solve_rg) against the oracle's a2 on oracle/rgraph_oracle.DenseGraph: statuses, lost
    set, dense edge state exact; pose 1e-6 / 1e-5; positions 1e-4; LM traces until the noise floor."""
    import copy
    from conftest import compare_lm_traces
    import nrs_synth as S
    tp = S.make_tracking_problem(n, seed, model)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    sigma, th = tp["graph"]["sigma"], tp["graph"]["stretch_th"]
    ids = np.arange(n, dtype=np.int32)
    g = nrs.RGraph(ctx, n, sigma, th)
    D = RG.DenseGraph(n, sigma, th)
    g.add_edges(tp["X_prev"], ids, ids)
    D.add_edges(tp["X_prev"], ids, ids)
    # history: one earlier frame in which a patch stretched (some edges are BAD when a2 runs)
    rng = np.random.default_rng(seed)
    hist = tp["X_prev"].copy()
    c0 = hist[3]
    patch = np.linalg.norm(hist - c0, axis=1) < 2.5 * sigma
    hist[patch] = c0 + (hist[patch] - c0) * np.float32(2.6)
    upd = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.int32)
    assert np.array_equal(g.update(hist, upd), np.array([D.update_vertex(hist, int(i)) for i in upd]))
    fm = np.arange(n, dtype=np.int32)
    tr = nrs.Trace(1024)
    r = ctx.track_deform_solve_rg(cam, g, tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tr, 64)
    otr = []
    o = O.track_deform_solve(tp["model"], tp["prm"], D, tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"],
                             tp["scale"], otr)
    assert np.allclose(r["pose_q"], o["pose_q"], atol=1e-6, rtol=0) and np.allclose(r["pose_t"], o["pose_t"], atol=1e-5, rtol=0)
    assert np.array_equal(r["f_status"], o["f_status"]) and r["lost"] == o["lost"]
    assert np.allclose(r["f_pos"], o["f_pos"], atol=1e-4, rtol=0) and np.allclose(r["map_pos"], o["map_pos"], atol=1e-4, rtol=0)
    assert compare_lm_traces(tr.trials, otr, len(otr)) >= 6
    probe = np.sort(rng.choice(n, 30, replace=False)).astype(np.int32)
    mx, mn, d0, st = g.rows(probe)
    assert np.array_equal(st, D.st[probe])                       # the graph after OPT:457-474, all N - 1 connections per point
    ex = D.st[probe] != RG.NONE
    # max / min distances come from positions that agree to 1e-4 only (fp32 results of an fp64 solve): not bit-compared
    assert np.allclose(mx[ex], D.maxd[probe][ex], atol=2e-4, rtol=0) and np.allclose(mn[ex], D.mind[probe][ex], atol=2e-4, rtol=0)
    # all-pairs semantics: nobody falls under the "fewer than 5 good connections" rule here, unlike on a radius-cut graph
    assert (r["f_status"] == 3).sum() == (o["f_status"] == 3).sum()
    g.close()
