"""GPU parity of the embedded-deformation mode of a2 (include/nrs.h nrs_track_deform_solve_embedded) AT THE SIZE bench.py MEASURES IT ON
-- 5000 points x 500 nodes, all-pairs graph, pinhole and KannalaBrandt8 -- against the output of oracle/embedded_oracle.py committed in
tests/golden/embedded5k_{pinhole,kb8}.npz (tests/golden/make_embedded5k_golden.py: the oracle ran once in the build container).  The
inputs are regenerated from the same seeds (the fixture carries their checksum); node selection on the device must reproduce the
oracle's.  Tolerances of tests/test_gpu_embedded.py: statuses, lost set, graph statuses exact; pose 1e-6 / 1e-5; positions 1e-4; LM
trials until the oracle's own decision sits on the fp32 noise floor.  ("Parity unpinned" beyond every-point-a-node: this holds the
product to its oracle at full size, not to reference output.)"""
import os
import sys

import numpy as np
import pytest

import nrs

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_embedded5k_golden import make_inputs, N_POINTS, N_NODES, CASES  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,model,seed", CASES)
def test_embedded_5k_x_500_matches_the_oracle_golden(ctx, name, model, seed):
    from conftest import compare_lm_traces
    G = np.load(os.path.join(HERE, "golden", "embedded5k_%s.npz" % name))
    n = int(G["n"])
    assert n == N_POINTS and int(G["m"]) == N_NODES
    tp, node, probe = make_inputs(seed, model)
    chk = tp["uv"].astype(np.float64).sum() + tp["X_prev"].astype(np.float64).sum() + tp["status"].sum() + node.sum()
    assert chk == float(G["in_sum"]), "the regenerated inputs are not the ones the golden was made from"
    dev_nodes = ctx.skin_select_nodes(tp["X_prev"], N_NODES, tp["status"] == 0)
    assert np.array_equal(np.sort(dev_nodes), np.where(node)[0])                     # the device's node set is the oracle's
    cam = nrs.make_camera(tp["model"], tp["prm"])
    ids = np.arange(n, dtype=np.int32)
    g = nrs.RGraph(ctx, n, tp["graph"]["sigma"], tp["graph"]["stretch_th"])
    try:
        g.add_edges(tp["X_prev"], ids, ids)
        tr = nrs.Trace(1024)
        r = ctx.track_deform_solve_embedded(cam, g, tp["X_prev"], ids, tp["status"], tp["uv"], tp["X_prev"], node, tp["pose_q"], tp["pose_t"], tp["scale"], tr, 256)
        assert np.allclose(r["pose_q"], G["pose_q"], atol=1e-6, rtol=0) and np.allclose(r["pose_t"], G["pose_t"], atol=1e-5, rtol=0)
        assert np.array_equal(r["f_status"], G["f_status"].astype(r["f_status"].dtype))
        assert r["lost"] == G["lost"].tolist()
        assert np.allclose(r["f_pos"], G["f_pos"], atol=1e-4, rtol=0) and np.allclose(r["map_pos"], G["map_pos"], atol=1e-4, rtol=0)
        assert abs(r["median"] - float(G["median"])) < 1e-5
        T = G["trials"]
        otr = [[dict(iter=int(t[1]), trial=int(t[2]), lam=t[3], chi=t[4], chi_new=t[5], rho=t[6], accepted=bool(t[7])) for t in T if int(t[0]) == rnd]
               for rnd in range(int(T[:, 0].max()) + 1)]
        assert compare_lm_traces(tr.trials, otr, len(otr)) >= 6
        assert np.array_equal(g.rows(G["probe"])[3], G["probe_status"].view(np.uint8))
        assert int(G["n_skinned"]) > 3500
    finally:
        g.close()
