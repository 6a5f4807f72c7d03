"""Scratch probe: optimize(5) on a named synthetic config (C2..C5), resident problem, prints ms per call."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
w = sys.argv[1] if len(sys.argv) > 1 else "C3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n_points, n_kf, seed, model = S.CONFIGS[w]
p = S.make_dba_problem(n_points, n_kf, seed, model)
ctx = nrs.Context()
cam = nrs.make_camera(p["model"], p["prm"])
e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
for r in range(reps):
    ctx.dba_reset()
    tr = nrs.Trace()
    t0 = time.perf_counter(); ctx.dba_optimize(5, tr); t1 = time.perf_counter()
    print("%s optimize(5): %.2f ms, trials %d, pcg %d" % (w, 1e3 * (t1 - t0), len(tr.trials), sum(t["inner"] for t in tr.trials)))
