NRS_DEBUG=TIMING=1 python - <<'PY' 2>&1 | grep "\[nrs\] a2" | tail -40
import os,sys
for p in ("", "nr-slam_amd/py", "oracle", "tests"):
    sys.path.insert(0, os.path.join(os.getcwd(), p))
import bench
r = bench.tracked_fps(5000, 8, dense_graph=True, direct_solve=1)
print(r["ms_pose_and_deformation"])
PY
