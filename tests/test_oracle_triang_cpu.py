"""oracle/triang_oracle.py (DeformableTriangulation restated, g2o_optimization.cc:559-814): behaviour checks that need no GPU."""
import collections

import numpy as np

import nrs_oracle as O
import nrs_synth as S
import triang_oracle as T


def test_numeric_jacobian_is_quantisation_noise():
    """delta = 1e-9 through the fp32 projection -- the central difference is zero unless an estimate sits
    within 1e-9 of a float rounding boundary, where it is about one pixel ulp / 2e-9"""
    rng = np.random.default_rng(0)
    x = np.stack([rng.uniform(-1, 1, 4000), rng.uniform(-1, 1, 4000), rng.uniform(2.5, 3.5, 4000)], 1)
    G = O.Graph(S.PINHOLE, S.HAMLYN_PINHOLE, np.zeros((0, 4)), np.zeros((0, 3)), x)
    e = T.ReprojNumeric(np.zeros((4000, 2)), 4.0)
    J = e.jacobians(G, np.arange(4000))[0]
    nz = J != 0
    assert 0 < nz.mean() < 0.02                                  # a handful of spikes in 24 000 entries
    assert np.abs(J[nz]).min() > 1e3                             # ... and they are huge (one float ulp of a pixel / 2e-9)


def test_outcomes_and_determinism():
    tb = S.make_temporal_buffer(12, 3)
    res = [T.deformable_triangulation(tb, int(c), tb["model"], tb["prm"]) for c in tb["cand"][:30]]
    res2 = [T.deformable_triangulation(tb, int(c), tb["model"], tb["prm"]) for c in tb["cand"][:30]]
    assert all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(res, res2))
    cnt = collections.Counter(r[0] for r in res)
    assert cnt[T.OK] >= 15 and cnt[T.E_SHORT] > 0
    ok = np.array([r[0] == T.OK for r in res])
    d = np.linalg.norm(np.array([r[1] for r in res])[ok] - tb["truth"][:30][ok], axis=1)
    assert np.median(d) < 0.15                                   # the depth follows the neighbours' mean (OPT:653-675)
    # neighbour rule (temporal_buffer.cc:97-141): 11 closest TRACKED_WITH_3D keypoints within [20, 500] px, none closer than 20
    nb = T.closest_map_points(tb, int(tb["cand"][0]))
    assert len(nb) == 11 and all(tb["status"][j] == 0 for j in nb)
    dense = S.make_temporal_buffer(8, 9, spacing=14.0)
    assert T.closest_map_points(dense, int(dense["cand"][0])) is None
