import os, sys, cProfile, pstats
for p in ("", "nr-slam_amd/py", "oracle", "tests"):
    sys.path.insert(0, os.path.join(os.getcwd(), p))
import bench
bench.tracked_fps(5000, 4, dense_graph=True, direct_solve=1)
pr = cProfile.Profile(); pr.enable()
r = bench.tracked_fps(5000, 12, dense_graph=True, direct_solve=1)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
print(r["ms_pose_and_deformation"], r["ms_per_frame_median"])
