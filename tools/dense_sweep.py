"""Seeded sweep: a2 on the device-resident all-pairs graph (nrs_track_deform_solve_rg), in parity mode and in skinned mode
(nrs_skin_select_nodes + demoted statuses), against the oracle's a2 on oracle/rgraph_oracle.DenseGraph -- the checks and
tolerances of tests/test_gpu_rgraph.py / test_gpu_skin.py.   python tools/dense_sweep.py [seeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, nrs, nrs_synth as S, nrs_oracle as O, rgraph_oracle as RG, skin_oracle as K
from conftest import compare_lm_traces
ctx = nrs.Context()
nseed = int(sys.argv[1]) if len(sys.argv) > 1 else 10
viol, t0 = 0, time.time()
def check(cond, msg):
    global viol
    if not cond:
        viol += 1; print("VIOLATION:", msg, flush=True)
for seed in range(nseed):
    rng = np.random.default_rng(700 + seed)
    n = int(rng.integers(250, 700)); model = S.PINHOLE if seed % 3 else S.KB8
    tp = S.make_tracking_problem(n, 800 + seed, model)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    ids = np.arange(n, dtype=np.int32)
    for mode in ("parity", "skinned"):
        st = tp["status"]
        if mode == "skinned":
            m = max(30, n // 6)
            nodes = ctx.skin_select_nodes(tp["X_prev"], m, tp["status"] == 0)
            check(np.array_equal(nodes, K.select_nodes(tp["X_prev"], m, tp["status"] == 0)), "seed %d node selection" % seed)
            st = nrs.skinned_status(tp["status"], ids, nodes)
        g = nrs.RGraph(ctx, n, tp["graph"]["sigma"], tp["graph"]["stretch_th"]); D = RG.DenseGraph(n, tp["graph"]["sigma"], tp["graph"]["stretch_th"])
        g.add_edges(tp["X_prev"], ids, ids); D.add_edges(tp["X_prev"], ids, ids)
        hist = (tp["X_prev"] * np.float32(1.0) + rng.normal(0, 0.05, tp["X_prev"].shape)).astype(np.float32)
        hist[: n // 8, :2] *= np.float32(2.4)                       # an earlier frame that stretched a patch: BAD edges exist
        upd = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.int32)
        check(np.array_equal(g.update(hist, upd), np.array([D.update_vertex(hist, int(i)) for i in upd])), "seed %d %s history update" % (seed, mode))
        tr, otr = nrs.Trace(1024), []
        r = ctx.track_deform_solve_rg(cam, g, tp["X_prev"], ids, st, tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tr, 256)
        o = O.track_deform_solve(tp["model"], tp["prm"], D, tp["X_prev"], ids, st, tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], otr)
        tag = "seed %d n %d %s" % (seed, n, mode)
        check(np.allclose(r["pose_q"], o["pose_q"], atol=1e-6, rtol=0) and np.allclose(r["pose_t"], o["pose_t"], atol=1e-5, rtol=0), tag + " pose")
        check(np.array_equal(r["f_status"], o["f_status"]) and r["lost"] == o["lost"], tag + " statuses / lost set")
        check(np.allclose(r["f_pos"], o["f_pos"], atol=1e-4, rtol=0) and np.allclose(r["map_pos"], o["map_pos"], atol=1e-4, rtol=0), tag + " positions")
        try:                                                        # (lambda follows rho^3 and rho is a difference of chi2 values: near convergence
            ncmp = compare_lm_traces(tr.trials, otr, len(otr), rtol=1e-5)   # 1e-11 in chi2 becomes 1e-6 in lambda, hence 1e-5 here)
        except AssertionError as ex:
            ncmp = -1; print("   ", str(ex)[:300])
        check(ncmp >= 4, tag + " LM traces")
        probe = np.sort(rng.choice(n, 20, replace=False)).astype(np.int32)
        check(np.array_equal(g.rows(probe)[3], D.st[probe]), tag + " graph state")
        g.close()
print("seeds %d, violations %d, %.0f s" % (nseed, viol, time.time() - t0))
