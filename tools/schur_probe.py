"""N1 (north star: "Schur-complement LM solve"): does eliminating the pose blocks buy PCG iterations on the windows
where it could matter (K = 50 / 200 keyframes)?  C++ restatement (oracle/nrs_cpu.cpp), same LM, same tolerance:
solver 1 = block-Jacobi PCG on the full system, solver 2 = PCG on the pose-eliminated landmark system with its exact
diagonal blocks.  Prints PCG iterations per LM trial for both.   python tools/schur_probe.py C3|<points>x<keyframes> [max_trials] [threads]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, nrs, nrs_synth as S, nrs_cpu as CPU
w = sys.argv[1] if len(sys.argv) > 1 else "C3"
mt = int(sys.argv[2]) if len(sys.argv) > 2 else 0
th = int(sys.argv[3]) if len(sys.argv) > 3 else min(32, CPU.max_threads())
if "x" in w:                                                      # "2500x200": points x keyframes, pinhole (a K = 200 window at C3 cost)
    n_pts, n_kf = (int(v) for v in w.split("x"))
    p = S.make_dba_problem(n_pts, n_kf, 7, 0)
else:
    p = S.make_dba_problem(w)
e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
out = {}
for solver in (1, 2):
    q, t, x, tr, st = CPU.dba_solve(p["model"], p["prm"], p["poses_q"], p["poses_t"], p["lm_xyz"], p["lm_kf"], p["lm_uv"], e["sp_ij"], e["sp_d0"],
                                    e["dm_idx"], e["dm_w"], p["scale"], 5, solver, 1e-10, th, mt)
    out["full" if solver == 1 else "pose_eliminated"] = dict(inner=[a["inner"] for a in tr], accepted=[a["accepted"] for a in tr],
                                                            chi_new=[a["chi_new"] for a in tr], seconds=st["t_total"])
a, b = out["full"], out["pose_eliminated"]
print(json.dumps(dict(workload=w, keyframes=int(p["n_kf"]), landmarks=len(p["lm_kf"]), threads=th, pcg_iters_full=a["inner"], pcg_iters_pose_eliminated=b["inner"],
                      total_full=sum(a["inner"]), total_pose_eliminated=sum(b["inner"]), same_decisions=a["accepted"] == b["accepted"],
                      max_rel_chi_diff=max(abs(x - y) / y for x, y in zip(a["chi_new"], b["chi_new"])), s_full=a["seconds"], s_pose_eliminated=b["seconds"])))
