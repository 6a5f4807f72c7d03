"""Scratch probe: per-frame tracking cost (LK + pose-only + pose-and-deformation) at 5k points."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ctx = nrs.Context()
sq = S.make_lk_sequence(n, 5)
tp = S.make_tracking_problem(n, 3)
cam = nrs.make_camera(tp["model"], tp["prm"])
m = tp["status"] == 0
fm = np.arange(n, dtype=np.int32)
ctx.klt_configure(); ctx.klt_set_reference(sq["im0"], sq["pts"])
st = np.zeros(len(sq["pts"]), np.int32)
for rep in range(4):
    t0 = time.perf_counter(); ctx.klt_set_reference(sq["im0"], sq["pts"]); t1 = time.perf_counter()
    xy, st2, good, _ = ctx.klt_track(sq["im1"], sq["pts"], st); t2 = time.perf_counter()
    ctx.pose_only_solve(cam, tp["uv"][m], tp["X_prev"][m], tp["pose_q"], tp["pose_t"]); t3 = time.perf_counter()
    tr = nrs.Trace(1024)
    r = ctx.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tr); t4 = time.perf_counter()
    print("points %d/%d: set_ref %.2f ms, klt_track %.2f ms (good %d), pose_only %.2f ms, track_deform %.2f ms (trials %d, pcg %d, lost %d)" % (
        len(sq["pts"]), m.sum(), 1e3*(t1-t0), 1e3*(t2-t1), good, 1e3*(t3-t2), 1e3*(t4-t3), len(tr.trials), sum(x["inner"] for x in tr.trials), len(r["lost"])))
print([ (x["round"], x["iter"], x["trial"], x["inner"], x["accepted"]) for x in tr.trials])
