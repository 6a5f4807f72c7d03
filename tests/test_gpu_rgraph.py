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
        assert np.array_equal(col[r, :cnt[r]], js) and np.array_equal(st[r, :cnt[r]], ost)
        assert np.array_equal(w[r, :cnt[r]], ow) and np.array_equal(d0[r, :cnt[r]], od0)
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
    assert cnt.min() > 3 and cnt.max() < 256 and (D.st[probe[0]] != RG.NONE).sum() == n0 - 1
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
    with pytest.raises(nrs.NrsError):
        big.get_edges([0], 16)
    cnt, col, *_ = big.get_edges([0], 512)
    assert cnt[0] == 299
    big.close()
