import sys, os, cProfile, pstats, io
for p in ("", "nr-slam_amd/py", "oracle", "tests", "tools"): sys.path.insert(0, os.path.join(os.getcwd(), p))
import bench
bench.tracked_fps(5000, 3, dense_graph=True, direct_solve=1)   # warm
pr = cProfile.Profile(); pr.enable()
r = bench.tracked_fps(5000, 7, dense_graph=True, direct_solve=1)
pr.disable()
print(r["value"], r["ms_pose_and_deformation"], r["ms_klt_track"], r["ms_pose_only"], r["ms_point_reuse"])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
