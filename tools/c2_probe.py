"""Scratch probe: time the resident C2 solve and print the LM/PCG trace."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
p = S.make_dba_problem(name)
e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
cam = nrs.make_camera(p["model"], p["prm"])
qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
ctx = nrs.Context()
print(ctx.device_name(), "n_lm", len(p["lm_kf"]), "springs", len(e["sp_ij"]), "dampers", len(e["dm_idx"]))
t = time.time(); ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"]); print("upload %.3fs" % (time.time() - t))
for rep in range(3):
    ctx.dba_reset(); tr = nrs.Trace()
    t = time.time(); ctx.dba_optimize(5, tr); dt = time.time() - t
    print("optimize(5): %.2f ms, iters %d, trials %d, pcg %s" % (dt * 1e3, tr.iterations, len(tr.trials), [x["inner"] for x in tr.trials]))
for x in tr.trials: print(x)
