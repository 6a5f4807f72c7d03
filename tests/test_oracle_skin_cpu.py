"""CPU: the node-selection restatement (oracle/skin_oracle.py) against a literal farthest-point loop and the properties
that define the sampling (distinct picks, non-increasing cover radius, eligibility)."""
import numpy as np

import skin_oracle as K


def _literal(pos, m, ok):
    pos = pos.astype(np.float32)
    sel = [int(np.argmax(ok))]
    while len(sel) < m:
        best, bi = np.float32(-1), -1
        for j in range(len(pos)):
            if not ok[j] or j in sel:
                continue
            dmin = np.float32(np.inf)
            for s in sel:
                d = pos[j] - pos[s]
                d2 = np.float32(np.float32(d[0] * d[0]) + np.float32(d[1] * d[1])) + np.float32(d[2] * d[2])
                dmin = min(dmin, np.float32(d2))
            if dmin > best:
                best, bi = dmin, j
        sel.append(bi)
    return np.asarray(sel, np.int32)


def test_matches_a_literal_loop_and_the_defining_properties():
    rng = np.random.default_rng(4)
    pos = rng.uniform(-10, 10, (120, 3)).astype(np.float32)
    pos[60] = pos[20]
    ok = rng.uniform(size=120) < 0.7
    assert np.array_equal(K.select_nodes(pos, 25, ok), _literal(pos, 25, ok))
    assert np.array_equal(K.select_nodes(pos, 25), _literal(pos, 25, np.ones(120, bool)))
    big = rng.uniform(-20, 20, (3000, 3)).astype(np.float32)
    ids = K.select_nodes(big, 300)
    assert len(set(ids.tolist())) == 300 and ids[0] == 0
    # cover radius (distance of pick k to the picks before it) never grows
    rad = [np.min(np.linalg.norm(big[ids[:k]] - big[ids[k]], axis=1)) for k in range(1, 300)]
    assert all(rad[i + 1] <= rad[i] * (1 + 1e-6) for i in range(len(rad) - 1))
    try:
        K.select_nodes(pos, int(ok.sum()) + 1, ok)
        assert False
    except ValueError:
        pass
