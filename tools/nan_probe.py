import sys, time
sys.path.insert(0, "/root/repo/nr-slam_amd/py")
import numpy as np, nrs, nrs_synth as S
ctx = nrs.Context()
p = S.make_dba_problem(300, 3, 5)
e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
cam = nrs.make_camera(p["model"], p["prm"]); qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
for what in ("uv", "xyz", "pose"):
    uv, xyz, q = p["lm_uv"].copy(), p["lm_xyz"].copy(), qt.copy()
    if what == "uv": uv[5, 0] = np.nan
    if what == "xyz": xyz[7, 2] = np.nan
    if what == "pose": q[1, 4] = np.inf
    t0 = time.time()
    try:
        tr = nrs.Trace()
        pq, x = ctx.dba_solve(cam, q, xyz, p["lm_kf"], uv, e, p["scale"], 5, tr)
        print("BA", what, "returned ok in %.2fs" % (time.time() - t0), "finite:", np.isfinite(pq).all(), np.isfinite(x).all(), "trials", len(tr.trials))
    except nrs.NrsError as ex:
        print("BA", what, "error in %.2fs:" % (time.time() - t0), str(ex)[:80])
tp = S.make_tracking_problem(300, 6)
camt = nrs.make_camera(tp["model"], tp["prm"]); fm = np.arange(300, dtype=np.int32)
for what in ("uv", "pos"):
    uv, X = tp["uv"].copy(), tp["X_prev"].copy()
    if what == "uv": uv[np.where(tp["status"] == 0)[0][3], 1] = np.nan
    if what == "pos": X[np.where(tp["status"] == 0)[0][4], 0] = np.nan
    t0 = time.time()
    try:
        r = ctx.track_deform_solve(camt, tp["graph"], X, fm, tp["status"], uv, X, tp["pose_q"], tp["pose_t"], tp["scale"])
        print("a2", what, "returned ok in %.2fs" % (time.time() - t0), "finite pose:", np.isfinite(r["pose_t"]).all())
    except nrs.NrsError as ex:
        print("a2", what, "error in %.2fs:" % (time.time() - t0), str(ex)[:80])
    try:
        m = tp["status"] == 0
        q1, t1, inl = ctx.pose_only_solve(camt, uv[m], X[m], tp["pose_q"], tp["pose_t"])
        print("a1", what, "returned", np.isfinite(q1).all(), np.isfinite(t1).all())
    except nrs.NrsError as ex:
        print("a1", what, "error:", str(ex)[:80])
