"""The C++ side of the boundary: nr-slam_amd/host/nrs_views.hpp (the reference's function names on
flat views) compiled into nr-slam_amd/host_demo with plain g++ and run on the GPU.  Its results must
be bit-identical to the ctypes path through the same C ABI (same library, same inputs)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import nrs
import nrs_synth as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "nr-slam_amd", "host_demo")


def _w(f, a):
    a = np.ascontiguousarray(a)
    f.write(struct.pack("<q", a.nbytes))
    f.write(a.tobytes())


def _r(f, dtype):
    (n,) = struct.unpack("<q", f.read(8))
    return np.frombuffer(f.read(n), dtype=dtype).copy()


def _graph(f, g, scale):
    for k, dt in (("rowptr", np.int32), ("col", np.int32), ("eid", np.int32), ("e_w", np.float32), ("e_d0", np.float32),
                  ("e_max", np.float32), ("e_min", np.float32), ("e_status", np.int32)):
        _w(f, np.asarray(g[k], dt))


def test_cpp_host_mirror_matches_ctypes_path(ctx, tmp_path):
    if not os.path.exists(DEMO):                                # built by `make -C nr-slam_amd` / __graft_entry__.build()
        subprocess.run(["make", "-C", os.path.join(ROOT, "nr-slam_amd"), "host_demo"], check=True, capture_output=True, timeout=600)
    n = 300
    tp = S.make_tracking_problem(n, 31)
    p = S.make_dba_problem(220, 4, 32)
    fm = np.arange(n, dtype=np.int32)
    qt = np.concatenate([tp["pose_q"], tp["pose_t"]]).astype(np.float64)
    kf_rowptr = np.concatenate([[0], np.cumsum([len(k) for k in p["kf_points"]])]).astype(np.int32)
    kf_pt = np.concatenate(p["kf_points"]).astype(np.int32)
    wqt = np.concatenate([p["poses_q"], p["poses_t"]], 1).astype(np.float64)
    src, dst = tmp_path / "in.blob", tmp_path / "out.blob"
    with open(src, "wb") as f:
        _w(f, np.array([tp["model"]], np.int32)); _w(f, np.asarray(tp["prm"], np.float32))
        _w(f, tp["uv"].astype(np.float32)); _w(f, tp["X_prev"].astype(np.float32)); _w(f, tp["status"].astype(np.int32)); _w(f, fm); _w(f, qt)
        _graph(f, tp["graph"], tp["scale"]); _w(f, tp["X_prev"].astype(np.float32))
        _w(f, np.array([tp["graph"]["sigma"], tp["graph"]["stretch_th"], tp["scale"]], np.float32))
        _w(f, wqt); _w(f, kf_rowptr); _w(f, kf_pt); _w(f, p["lm_uv"].astype(np.float32)); _w(f, p["lm_xyz"].astype(np.float32))
        _graph(f, p["graph"], p["scale"])
        _w(f, np.array([p["graph"]["sigma"], p["graph"]["stretch_th"], p["scale"]], np.float32))
    r = subprocess.run([DEMO, str(src), str(dst)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    with open(dst, "rb") as f:
        a1 = _r(f, np.float64)
        a2_qt, a2_pos, a2_st, a2_lost, a2_map, a2_est = _r(f, np.float64), _r(f, np.float32), _r(f, np.int32), _r(f, np.int32), _r(f, np.float32), _r(f, np.int32)
        a3_qt, a3_xyz = _r(f, np.float64), _r(f, np.float32)
        a3e_qt, a3e_xyz = _r(f, np.float64), _r(f, np.float32)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    m = tp["status"] == 0
    q, t, _ = ctx.pose_only_solve(cam, tp["uv"][m], tp["X_prev"][m], tp["pose_q"], tp["pose_t"])
    assert np.array_equal(a1, np.concatenate([q, t]))
    r2 = ctx.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"])
    assert np.array_equal(a2_qt, np.concatenate([r2["pose_q"], r2["pose_t"]]))
    assert np.array_equal(a2_pos.reshape(-1, 3), r2["f_pos"]) and np.array_equal(a2_st, r2["f_status"])
    assert sorted(a2_lost.tolist()) == sorted(r2["lost"]) and np.array_equal(a2_map.reshape(-1, 3), r2["map_pos"])
    assert np.array_equal(a2_est, r2["graph"]["e_status"])
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    camb = nrs.make_camera(p["model"], p["prm"])
    pq, xyz = ctx.dba_solve(camb, wqt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], 5)
    assert np.array_equal(a3_qt.reshape(-1, 7), pq) and np.array_equal(a3_xyz.reshape(-1, 3), xyz)
    # the embedded form of the window (N2b): every 6th map point a node, the ordered lists cut down to nodes -- the mirror's
    # LocalDeformableBundleAdjustmentEmbedded against the ctypes path, bit for bit
    flag = np.zeros(p["n_points"], np.uint8)
    flag[::6] = 1
    nb = p["nbr"]
    keep = flag[nb["col"]] != 0
    rows = np.repeat(np.arange(p["n_points"]), np.diff(nb["rowptr"]))
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=p["n_points"]))]).astype(np.int32)
    nbn = dict(rowptr=rp, col=nb["col"][keep], w=nb["w"][keep], d0=nb["d0"][keep], status=nb["status"][keep])
    ee = nrs.dba_build_edges_embedded(p["kf_points"], flag, nbn)
    we = S.embedded_window(p, ee)
    pqe, xe, ske = ctx.dba_solve_embedded(camb, wqt, we, ee, p["scale"], 5)
    full = p["lm_xyz"].astype(np.float32).copy()
    full[ee["lm_obs"]] = xe
    full[ee["sk_obs"]] = ske
    assert len(ee["sk_obs"]) > 100 and np.array_equal(a3e_qt.reshape(-1, 7), pqe) and np.array_equal(a3e_xyz.reshape(-1, 3), full)


def test_cpp_tracker_graph_and_triangulation_mirror(tmp_path):
    """LucasKanadeTracker / RegularizationGraph / DeformableTriangulation through the compiled C++ mirror classes
    (nr-slam_amd/host/nrs_views.hpp, host_demo2.cpp; the reference's method names) against the ctypes path."""
    demo = os.path.join(ROOT, "nr-slam_amd", "host_demo2")
    if not os.path.exists(demo):
        subprocess.run(["make", "-C", os.path.join(ROOT, "nr-slam_amd"), "host_demo2"], check=True, capture_output=True, timeout=600)
    sq = S.make_lk_sequence(300, 5)
    h, w = sq["im0"].shape
    rng = np.random.default_rng(4)
    n = 400
    pos0 = np.stack([rng.uniform(-10, 10, n), rng.uniform(-8, 8, n), 60 + rng.normal(0, 1, n)], 1).astype(np.float32)
    pos1 = (pos0 * np.array([1.9, 1.9, 1.0], np.float32) + rng.normal(0, 0.02, pos0.shape)).astype(np.float32)
    upd = np.sort(rng.choice(n, 300, replace=False)).astype(np.int32)
    tb = S.make_temporal_buffer(10, 12)
    src, dst = tmp_path / "in2.blob", tmp_path / "out2.blob"
    with open(src, "wb") as f:
        _w(f, np.array([w, h], np.int32)); _w(f, sq["im0"]); _w(f, sq["im1"]); _w(f, sq["pts"].astype(np.float32))
        _w(f, np.array([3.0, 1.1], np.float32)); _w(f, pos0); _w(f, pos1); _w(f, upd)
        _w(f, np.array([tb["model"]], np.int32)); _w(f, np.asarray(tb["prm"], np.float32))
        _w(f, np.array(tb["has_kp"].shape, np.int32)); _w(f, tb["poses"].astype(np.float32)); _w(f, tb["has_kp"].astype(np.uint8))
        _w(f, tb["kp_xy"].astype(np.float32)); _w(f, tb["has_lm"].astype(np.uint8)); _w(f, tb["lm_xyz"].astype(np.float32))
        _w(f, tb["status"].astype(np.int32)); _w(f, tb["cand"].astype(np.int32))
        # skinned pose-and-deformation on the device-resident graph
        tp = S.make_tracking_problem(500, 37)
        n_nodes = 80
        fm = np.arange(500, dtype=np.int32)
        _w(f, np.array([tp["model"], n_nodes], np.int32)); _w(f, np.asarray(tp["prm"], np.float32))
        _w(f, tp["uv"].astype(np.float32)); _w(f, tp["X_prev"].astype(np.float32)); _w(f, tp["status"].astype(np.int32)); _w(f, fm)
        _w(f, np.concatenate([tp["pose_q"], tp["pose_t"]]).astype(np.float64)); _w(f, tp["X_prev"].astype(np.float32))
        _w(f, np.array([tp["graph"]["sigma"], tp["graph"]["stretch_th"], tp["scale"]], np.float32))
    r = subprocess.run([demo, str(src), str(dst)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    with open(dst, "rb") as f:
        nxt, st, good, npts, gray = _r(f, np.float32), _r(f, np.int32), _r(f, np.int32), _r(f, np.int32), _r(f, np.int16)
        goodc, flat, fw = _r(f, np.int32), _r(f, np.int32), _r(f, np.float32)
        tst, txyz = _r(f, np.int32), _r(f, np.float32)
        sk_nodes, sk_qt, sk_st, sk_lost, sk_map = _r(f, np.int32), _r(f, np.float64), _r(f, np.int32), _r(f, np.int32), _r(f, np.float32)
    c = nrs.Context()
    c.klt_configure()
    c.klt_set_reference(sq["im0"], sq["pts"])
    xy, st2, good2, _ = c.klt_track(sq["im1"], sq["pts"], np.zeros(len(sq["pts"]), np.int32), True, 0.7)
    assert np.array_equal(nxt.reshape(-1, 2), xy) and np.array_equal(st, st2) and good[0] == good2
    assert npts[0] == len(sq["pts"]) + 1 and np.array_equal(gray.reshape(-1, 21, 21), c.klt_get_template(3)["gray"])
    g = nrs.RGraph(c, n, 3.0, 1.1)
    ids = np.arange(n, dtype=np.int32)
    g.add_edges(pos0, ids, ids)
    assert np.array_equal(goodc, g.update(pos1, upd))
    cnt, col, wv, d0, stg = g.get_edges(ids, 256)
    k = kf = 0
    for i in range(n):
        assert flat[k] == cnt[i]
        row = flat[k + 1:k + 1 + 2 * cnt[i]].reshape(-1, 2)
        assert np.array_equal(row[:, 0], col[i, :cnt[i]]) and np.array_equal(row[:, 1], stg[i, :cnt[i]])
        fr = fw[kf:kf + 2 * cnt[i]].reshape(-1, 2)
        assert np.array_equal(fr[:, 0], wv[i, :cnt[i]]) and np.array_equal(fr[:, 1], d0[i, :cnt[i]])
        k += 1 + 2 * cnt[i]
        kf += 2 * cnt[i]
    g.close()
    s2, x2 = c.triangulate_batch(nrs.make_camera(tb["model"], tb["prm"]), tb, tb["cand"])
    assert np.array_equal(tst, s2) and np.array_equal(txyz.reshape(-1, 3), x2)
    nodes = c.skin_select_nodes(tp["X_prev"], n_nodes, tp["status"] == 0)
    assert np.array_equal(sk_nodes, nodes)
    g2 = nrs.RGraph(c, 500, tp["graph"]["sigma"], tp["graph"]["stretch_th"])
    g2.add_edges(tp["X_prev"], fm, fm)
    r3 = c.track_deform_solve_rg(nrs.make_camera(tp["model"], tp["prm"]), g2, tp["X_prev"], fm, nrs.skinned_status(tp["status"], fm, nodes), tp["uv"],
                                 tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], None, 256)
    assert np.array_equal(sk_qt, np.concatenate([r3["pose_q"], r3["pose_t"]])) and np.array_equal(sk_st, r3["f_status"])
    assert sorted(sk_lost.tolist()) == sorted(r3["lost"]) and np.array_equal(sk_map.reshape(-1, 3), r3["map_pos"])
    g2.close()
    c.close()
