"""GPU parity: a2 (CameraPoseAndDeformationOptimization, reference modules/optimization/g2o_optimization.cc:148-557) at the
size bench.py's tracked-fps figure is measured on -- ~4.4k points of one frame, the map's graph at the reference's all-pairs
density, on the default solver (the direct one) and on the PCG with its two-level preconditioner -- against the oracle's output committed
in tests/golden/track5k_{pinhole,kb8}.npz (tests/golden/make_track5k_golden.py ran oracle/nrs_oracle.track_deform_solve on
oracle/rgraph_oracle.DenseGraph once in the build container: minutes per frame).  The inputs are regenerated here from the
same seeds (nrs_synth is deterministic; the fixture carries a checksum of them).

Tolerances as in tests/test_gpu_track.py / test_gpu_rgraph.py: statuses, lost set, graph statuses exact; pose 1e-6 / 1e-5;
positions 1e-4 map units; LM trials (accept / reject, lambda, chi2) until the oracle's own decision sits on the fp32 noise
floor."""
import os
import sys

import numpy as np
import pytest

import nrs

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_track5k_golden import make_inputs, N_POINTS, CASES  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(c, name, model, seed):
    from conftest import compare_lm_traces
    G = np.load(os.path.join(HERE, "golden", "track5k_%s.npz" % name))
    n = int(G["n"])
    assert n == N_POINTS
    tp, hist, upd, probe = make_inputs(n, seed, model)
    chk = tp["uv"].astype(np.float64).sum() + tp["X_prev"].astype(np.float64).sum() + tp["status"].sum()
    assert chk == float(G["in_sum"]), "the regenerated inputs are not the ones the golden was made from"
    cam = nrs.make_camera(tp["model"], tp["prm"])
    ids = np.arange(n, dtype=np.int32)
    g = nrs.RGraph(c, n, tp["graph"]["sigma"], tp["graph"]["stretch_th"])
    try:
        g.add_edges(tp["X_prev"], ids, ids)
        assert np.array_equal(g.update(hist, upd), G["good"])          # the history frame: same good-connection counts (exact)
        tr = nrs.Trace(1024)
        r = c.track_deform_solve_rg(cam, g, tp["X_prev"], ids, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"],
                                    tp["scale"], tr, 128)
        assert np.allclose(r["pose_q"], G["pose_q"], atol=1e-6, rtol=0) and np.allclose(r["pose_t"], G["pose_t"], atol=1e-5, rtol=0)
        assert np.array_equal(r["f_status"], G["f_status"].astype(r["f_status"].dtype))
        assert r["lost"] == G["lost"].tolist() and len(r["lost"]) > 100
        assert np.allclose(r["f_pos"], G["f_pos"], atol=1e-4, rtol=0) and np.allclose(r["map_pos"], G["map_pos"], atol=1e-4, rtol=0)
        assert abs(r["median"] - float(G["median"])) < 1e-5
        T = G["trials"]
        otr = [[dict(iter=int(t[1]), trial=int(t[2]), lam=t[3], chi=t[4], chi_new=t[5], rho=t[6], accepted=bool(t[7])) for t in T if int(t[0]) == rnd]
               for rnd in range(int(T[:, 0].max()) + 1)]
        assert compare_lm_traces(tr.trials, otr, len(otr)) >= 9
        assert np.array_equal(g.rows(G["probe"])[3], G["probe_status"].view(np.uint8))     # graph after OPT:457-474 (255 = no edge)
        return tr
    finally:
        g.close()


@pytest.mark.parametrize("name,model,seed", CASES)
def test_track5k_default_mode(ctx, name, model, seed):
    tr = _run(ctx, name, model, seed)
    assert len(tr.trials) >= 30


@pytest.mark.parametrize("name,model,seed", CASES[:1])
def test_track5k_exact_trials(ctx_exact, name, model, seed):
    tr = _run(ctx_exact, name, model, seed)
    assert not any(t["early"] for t in tr.trials)


def test_track5k_default_is_the_direct_solver(ctx):
    """by default a frame of this size runs on the nested-dissection Cholesky (nrs_options.direct_solve = 0: up to 8000 free rows)"""
    tr = _run(ctx, *CASES[0])
    assert all(t["inner"] == 1 for t in tr.trials)


@pytest.mark.parametrize("name,model,seed", CASES)
def test_track5k_pcg_solver(ctx_pcg, name, model, seed):
    """the same frames on the PCG with the two-level preconditioner and early trial rejection (nrs_options.direct_solve = 2)"""
    tr = _run(ctx_pcg, name, model, seed)
    assert max(t["inner"] for t in tr.trials) > 1
