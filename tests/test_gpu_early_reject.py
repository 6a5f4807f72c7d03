"""Default (early-rejecting) LM trials against exact trials.  Early rejection only stops solving
trials that would be rejected anyway, so the accept/reject sequence, every lambda and the final state
must be those of the exact mode.  The seeds include frames whose late iterations sit on the fp32
noise floor of chi2 (where a gain ratio means nothing): there the early test must stay silent.
`tools/early_reject_sweep.py` runs the same comparison over hundreds of problems."""
import numpy as np
import pytest

import nrs
import nrs_synth as S

pytestmark = pytest.mark.gpu


def _seq(tr):
    return [(t["round"], t["iter"], t["trial"], bool(t["accepted"])) for t in tr.trials]


@pytest.mark.parametrize("seed", [5, 18, 20, 27, 30, 34, 41])
def test_tracking_frame_decisions_match_exact_mode(ctx_pcg, ctx_exact, seed):
    ctx = ctx_pcg                                                # (early rejection is the PCG's: both contexts run single-frame problems on it)
    rng = np.random.default_rng(seed)
    rng.integers(150, 1500), rng.integers(2, 7)              # same draws as the sweep tool
    n = int(rng.integers(300, 3000))
    model = S.PINHOLE if seed % 3 else S.KB8
    tp = S.make_tracking_problem(n, 2000 + seed, model)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    fm = np.arange(n, dtype=np.int32)
    out = []
    for c in (ctx, ctx_exact):
        tr = nrs.Trace(1024)
        r = c.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"],
                                 tp["pose_q"], tp["pose_t"], tp["scale"], tr)
        out.append((r, tr))
    (r0, t0), (r1, t1) = out
    assert _seq(t0) == _seq(t1)
    assert all(abs(a["lam"] - b["lam"]) <= 1e-12 * b["lam"] for a, b in zip(t0.trials, t1.trials))
    assert not any(t["early"] for t in t1.trials)
    assert r0["lost"] == r1["lost"] and np.array_equal(r0["f_status"], r1["f_status"])
    assert np.array_equal(r0["f_pos"], r1["f_pos"])          # accepted steps are the same launches on the same data


@pytest.mark.parametrize("seed", [3, 8, 13])
def test_ba_window_decisions_match_exact_mode(ctx, ctx_exact, seed):
    rng = np.random.default_rng(seed)
    n, k = int(rng.integers(150, 1500)), int(rng.integers(2, 7))
    model = S.PINHOLE if seed % 3 else S.KB8
    p = S.make_dba_problem(n, k, 1000 + seed, model)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    out = []
    for c in (ctx, ctx_exact):
        tr = nrs.Trace(256)
        pq, xyz = c.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], 8, tr)
        out.append((pq, xyz, tr))
    assert _seq(out[0][2]) == _seq(out[1][2])
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
