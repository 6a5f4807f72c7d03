import sys, os
sys.path.insert(0, "/root/repo/nr-slam_amd/py")
import numpy as np, nrs, nrs_synth as S
p = S.make_dba_problem(1500, 8, 75)
e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
cam = nrs.make_camera(p["model"], p["prm"]); qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
nrs.debug_set("NRS_FUSED_MAX_ROWS", "0")
for ecd in (0, 1):
    if ecd: nrs.debug_set("NRS_NO_ECD", None)
    else: nrs.debug_set("NRS_NO_ECD", "1")
    ctx = nrs.Context(exact_trials=1)
    tr = nrs.Trace()
    pq, xyz = ctx.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], 3, tr)
    print("ecd", ecd, [(t["inner"], t["accepted"]) for t in tr.trials], float(xyz.sum()))
