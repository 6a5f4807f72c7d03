"""bench.py's tracked-fps legs on their own: python tools/tracked_fps_probe.py [points] [frames]  (all-pairs graph; both a2 solvers)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("", "nr-slam_amd/py", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
fr = int(sys.argv[2]) if len(sys.argv) > 2 else 7
for name, kw in (("direct, all-pairs graph", dict(dense_graph=True, direct_solve=1)), ("pcg, all-pairs graph", dict(dense_graph=True, direct_solve=2)),
                 ("direct, flat graph", dict(direct_solve=1)), ("pcg, flat graph", dict(direct_solve=2))):
    r = bench.tracked_fps(n, fr, **kw)
    print("%-24s %.2f frames/s, a2 %.1f ms, trials/frame %.1f, plans %s" % (name, r["value"], r["ms_pose_and_deformation"], r["lm_trials_per_frame"],
                                                                        json.dumps(r["a2_solver"].get("symbolic_plans"))), flush=True)
