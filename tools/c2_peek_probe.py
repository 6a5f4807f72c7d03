import sys, os
sys.path.insert(0, "/root/repo/nr-slam_amd/py")
import numpy as np, nrs, nrs_synth as S
n_points, n_kf, seed, model = S.CONFIGS["C2"]
p = S.make_dba_problem(n_points, n_kf, seed, model)
e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
cam = nrs.make_camera(p["model"], p["prm"]); qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
ctx = nrs.Context(pcg_batch=1)
ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
ctx.dba_optimize(5)
