"""Scratch probe for PMC passes: one optimize(2) on a named config (default C4), nothing else."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
w = sys.argv[1] if len(sys.argv) > 1 else "C4"
n_points, n_kf, seed, model = S.CONFIGS[w]
p = S.make_dba_problem(n_points, n_kf, seed, model)
e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
ctx = nrs.Context()
ctx.dba_upload(nrs.make_camera(p["model"], p["prm"]), np.concatenate([p["poses_q"], p["poses_t"]], 1), p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
tr = nrs.Trace()
ctx.dba_optimize(int(sys.argv[2]) if len(sys.argv) > 2 else 2, tr)
print(w, "trials", len(tr.trials), "pcg", sum(t["inner"] for t in tr.trials))
ctx.close()
