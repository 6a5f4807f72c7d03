"""oracle/rgraph_oracle.py: the vectorised dense restatement of RegularizationGraph held to the literal one-pair-at-a-time
walk of the reference's containers (regularization_graph.cc:38-146, map.cc:148-166, mapping.cc:240-256)."""
import numpy as np

import nrs_oracle as O
import rgraph_oracle as RG


def _scene(n, seed):
    rng = np.random.default_rng(seed)
    pos = np.stack([rng.uniform(-10, 10, n), rng.uniform(-8, 8, n), 60 + rng.normal(0, 1.5, n)], 1).astype(np.float32)
    return rng, pos


def test_dense_matches_literal_walk():
    n, n0 = 60, 45
    rng, pos = _scene(n, 3)
    sigma, th = 4.0, 1.1
    D, L = RG.DenseGraph(n, sigma, th), RG.LiteralGraph(sigma, th)
    first = np.arange(n0)
    D.add_edges(pos, first, first)                                   # map.cc:148-166: every pair of initial points
    for i in range(n0):
        for j in range(i + 1, n0):
            L.add_edge(i, j, pos[j] - pos[i])
    for step in range(3):
        pos2 = pos.copy()
        pos2[:, :2] *= np.float32(1.0 + 0.8 * (step + 1))            # stretch: edges whose length more than doubles go BAD
        pos2 += rng.normal(0, 0.05, pos.shape).astype(np.float32)
        ids = np.sort(rng.choice(n0, 30, replace=False))
        assert [D.update_vertex(pos2, i) for i in ids] == [L.update_vertex(pos2, i) for i in ids]
        if step == 1:                                                # mapping.cc:240-256: new landmarks against the current ones
            new, cur = np.arange(n0, n), np.arange(n)
            D.add_edges(pos2, new, cur)
            for a in new:
                for b in cur:
                    if a != b:
                        L.add_edge(a, b, pos2[b] - pos2[a])
            n0 = n
        for i in range(n0):
            js, w, d0, st = D.get_edges(i)
            ref = L.get_edges(i)
            assert js.tolist() == [r[0] for r in ref]
            assert np.array_equal(w, np.array([r[1] for r in ref], np.float32)) and np.array_equal(d0, np.array([r[2] for r in ref], np.float32))
            assert st.tolist() == [r[3] for r in ref]
    assert (D.st == O.GRAPH_BAD).any()


def test_break_hides_later_status_classes():
    """GetEdges stops at the FIRST weight below min_weight of the (status, weight)-sorted list: a far NEUTRAL edge cuts
    off every BAD edge, however heavy (regularization_graph.cc:78-84)"""
    D = RG.DenseGraph(4, 1.0, 1.1)
    pos = np.array([[0, 0, 0], [0.5, 0, 0], [0.6, 0, 0], [9, 0, 0]], np.float32)
    D.add_edges(pos, np.arange(4), np.arange(4))
    D.st[0, 1] = D.st[1, 0] = O.GRAPH_BAD
    js, w, d0, st = D.get_edges(0)
    assert js.tolist() == [2]                  # NEUTRAL 2 kept, NEUTRAL 3 (far) breaks, BAD 1 never reached
    D.st[0, 3] = D.st[3, 0] = RG.NONE          # without the far edge the BAD one is returned last
    assert D.get_edges(0)[0].tolist() == [2, 1]
