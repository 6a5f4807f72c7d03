"""GPU path against the committed golden vectors (tests/golden/*.npz): the same entry points as the
live-oracle parity tests, but with inputs and expected outputs read from files."""
import os

import numpy as np
import pytest

import nrs

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_pose_only_golden(ctx):
    d = np.load(os.path.join(G, "pose_only_120.npz"))
    cam = nrs.make_camera(int(d["model"]), d["prm"])
    q, t, inl = ctx.pose_only_solve(cam, d["uv"], d["X"], d["pose_q"], d["pose_t"])
    assert np.allclose(q, d["out_q"], atol=1e-6, rtol=0) and np.allclose(t, d["out_t"], atol=1e-5, rtol=0)
    assert np.array_equal(inl, d["out_inlier"])


def test_dba_golden(ctx):
    d = np.load(os.path.join(G, "dba_90x3.npz"))
    cam = nrs.make_camera(int(d["model"]), d["prm"])
    kf_points = [d["kf_pt"][d["kf_rowptr"][k]:d["kf_rowptr"][k + 1]] for k in range(len(d["kf_rowptr"]) - 1)]
    nbr = dict(rowptr=d["nbr_rowptr"], col=d["nbr_col"], w=d["nbr_w"], d0=d["nbr_d0"], status=d["nbr_status"])
    e = nrs.dba_build_edges(kf_points, nbr)
    for k in ("sp_ij", "sp_d0", "dm_idx", "dm_w"):
        assert np.array_equal(e[k], d[k])
    qt = np.concatenate([d["poses_q"], d["poses_t"]], 1)
    tr = nrs.Trace()
    pq, xyz = ctx.dba_solve(cam, qt, d["lm_xyz"], d["lm_kf"], d["lm_uv"], e, float(d["scale"]), 5, tr)
    assert [t["accepted"] for t in tr.trials] == d["out_accepted"].tolist()
    assert np.allclose(pq[:, :4], d["out_q"], atol=1e-6, rtol=0) and np.allclose(pq[:, 4:], d["out_t"], atol=1e-5, rtol=0)
    assert np.allclose(xyz, d["out_pts"], atol=1e-4, rtol=0)


def test_track_golden(ctx):
    d = np.load(os.path.join(G, "track_150.npz"))
    cam = nrs.make_camera(int(d["model"]), d["prm"])
    g = {k[2:]: d[k] for k in d.files if k.startswith("g_")}
    g.update(sigma=float(d["sigma"]), stretch_th=float(d["stretch_th"]), min_w=float(d["min_w"]))
    n = len(d["status"])
    r = ctx.track_deform_solve(cam, g, d["X_prev"], np.arange(n), d["status"], d["uv"], d["X_prev"], d["pose_q"],
                               d["pose_t"], float(d["scale"]))
    assert np.allclose(r["pose_q"], d["out_q"], atol=1e-6, rtol=0) and np.allclose(r["pose_t"], d["out_t"], atol=1e-5, rtol=0)
    assert np.array_equal(r["f_status"], d["out_f_status"]) and r["lost"] == d["out_lost"].tolist()
    assert np.allclose(r["f_pos"], d["out_f_pos"], atol=1e-4, rtol=0) and np.allclose(r["map_pos"], d["out_map_pos"], atol=1e-4, rtol=0)
    assert np.array_equal(r["graph"]["e_status"], d["out_e_status"])
    assert np.allclose(r["graph"]["e_w"], d["out_e_w"], atol=1e-5) and abs(r["median"] - float(d["out_median"])) < 1e-5


def test_lk_golden(ctx):
    d = np.load(os.path.join(G, "lk_240x180.npz"))
    ctx.klt_clear()
    ctx.klt_configure(max_level=2)
    ctx.klt_set_reference(d["im0"], d["pts"])
    for i in (0, 13, len(d["pts"]) - 1):
        t = ctx.klt_get_template(i)
        assert np.array_equal(t["gray"][0], d["tpl_gray0"][i]) and np.array_equal(t["mean"][0], d["tpl_mean0"][i])
    xy, st, good, ssim = ctx.klt_track(d["im1"], d["pts"] + np.float32(0.5), np.zeros(len(d["pts"]), np.int32))
    assert np.array_equal(st, d["out_status"]) and good == int(d["out_good"])
    assert np.array_equal(xy, d["out_xy"])
    ctx.klt_clear()
    ctx.klt_configure()
