"""GPU parity, N2 (the skinned mode, "points x graph nodes"): node selection (nrs_skin_select_nodes) bit-exact against
oracle/skin_oracle.py, and the skinned pose-and-deformation solve -- the nodes carry stage 1, every other point of the
frame is carried by the reference's stage 2 (OPT:476-553) -- against the oracle's a2 on the same status vector."""
import numpy as np
import pytest

import nrs
import nrs_oracle as O
import rgraph_oracle as RG
import skin_oracle as K

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m,seed", [(700, 70, 1), (5000, 500, 2), (20000, 1000, 3)])
def test_node_selection_is_exact(ctx, n, m, seed):
    rng = np.random.default_rng(seed)
    pos = np.stack([rng.uniform(-22, 22, n), rng.uniform(-17, 17, n), 60 + rng.normal(0, 1.0, n)], 1).astype(np.float32)
    pos[n // 2] = pos[n // 3]                                    # a duplicate point: distance ties
    assert np.array_equal(ctx.skin_select_nodes(pos, m), K.select_nodes(pos, m))
    el = rng.uniform(size=n) < 0.6
    el[0] = False
    ids = ctx.skin_select_nodes(pos, m, el)
    assert np.array_equal(ids, K.select_nodes(pos, m, el)) and el[ids].all() and len(set(ids.tolist())) == m
    with pytest.raises(nrs.NrsError):
        ctx.skin_select_nodes(pos, int(el.sum()) + 1, el)
    # every point a node: all of them, each once
    small = pos[:300]
    assert sorted(ctx.skin_select_nodes(small, 300).tolist()) == list(range(300))


@pytest.mark.parametrize("n,m,seed,model", [(600, 120, 31, 0), (900, 150, 32, 1)])
def test_skinned_pose_and_deformation(ctx, n, m, seed, model):
    """5k x 500 in miniature: m nodes optimised, the other tracked points follow them through stage 2; dense graph"""
    from conftest import compare_lm_traces
    import nrs_synth as S
    tp = S.make_tracking_problem(n, seed, model)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    sigma, th = tp["graph"]["sigma"], tp["graph"]["stretch_th"]
    ids = np.arange(n, dtype=np.int32)
    fm = ids.copy()
    nodes = ctx.skin_select_nodes(tp["X_prev"], m, tp["status"] == 0)
    assert np.array_equal(nodes, K.select_nodes(tp["X_prev"], m, tp["status"] == 0))
    st = nrs.skinned_status(tp["status"], fm, nodes)
    assert (st == 0).sum() == m and (st == 1).sum() == (tp["status"] == 1).sum() + (tp["status"] == 0).sum() - m
    g = nrs.RGraph(ctx, n, sigma, th)
    D = RG.DenseGraph(n, sigma, th)
    g.add_edges(tp["X_prev"], ids, ids)
    D.add_edges(tp["X_prev"], ids, ids)
    tr = nrs.Trace(1024)
    r = ctx.track_deform_solve_rg(cam, g, tp["X_prev"], fm, st, tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tr, 256)
    otr = []
    o = O.track_deform_solve(tp["model"], tp["prm"], D, tp["X_prev"], fm, st, tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"],
                             tp["scale"], otr)
    assert np.allclose(r["pose_q"], o["pose_q"], atol=1e-6, rtol=0) and np.allclose(r["pose_t"], o["pose_t"], atol=1e-5, rtol=0)
    assert np.array_equal(r["f_status"], o["f_status"]) and r["lost"] == o["lost"]
    # the skinned set: every non-node point of the frame that lies within reach of a node was carried by stage 2
    assert len(r["lost"]) > 0.9 * ((st == 1).sum())
    assert np.allclose(r["f_pos"], o["f_pos"], atol=1e-4, rtol=0) and np.allclose(r["map_pos"], o["map_pos"], atol=1e-4, rtol=0)
    moved = np.linalg.norm(r["map_pos"][r["lost"]] - tp["X_prev"][r["lost"]], axis=1)
    assert np.median(moved) > 1e-3                               # the skinned points did follow the nodes
    assert compare_lm_traces(tr.trials, otr, len(otr)) >= 6
    g.close()


def test_result_does_not_depend_on_cap_per_point(ctx):
    """a walk that runs off a truncated GetEdges prefix makes the driver fetch four times as much and start over: the
    skinned solve (walks ~10x longer than in parity mode) from cap_per_point = 8 equals the one from 512, bit for bit"""
    import nrs_synth as S
    n, m = 500, 60
    tp = S.make_tracking_problem(n, 41)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    ids = np.arange(n, dtype=np.int32)
    st = nrs.skinned_status(tp["status"], ids, ctx.skin_select_nodes(tp["X_prev"], m, tp["status"] == 0))
    out = []
    for cap in (8, 512):
        g = nrs.RGraph(ctx, n, tp["graph"]["sigma"], tp["graph"]["stretch_th"])
        g.add_edges(tp["X_prev"], ids, ids)
        tr = nrs.Trace(1024)
        r = ctx.track_deform_solve_rg(cam, g, tp["X_prev"], ids, st, tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tr, cap)
        out.append((r, [(t["lam"], t["chi"], t["chi_new"], t["accepted"]) for t in tr.trials], g.rows(ids[:40])))
        g.close()
    a, b = out
    assert a[1] == b[1] and a[0]["lost"] == b[0]["lost"]
    for k in ("pose_q", "pose_t", "f_pos", "f_status", "map_pos"):
        assert np.array_equal(a[0][k], b[0][k]), k
    assert np.array_equal(a[2][3], b[2][3])                                       # statuses of the dense rows
    ex = a[2][3] != 255                                                           # (slots without an edge hold no data)
    for x, y in zip(a[2][:3], b[2][:3]):
        assert np.array_equal(x[ex], y[ex])
