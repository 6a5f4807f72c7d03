# tracked-fps leg (all-pairs graph, direct solver) with the set-up stages of a frame's engines on 1 / 3 / 4 / 6 host threads (HostPool, nrs_engine_setup.hpp)
for n in 1 4 3 6 1 4; do
  echo "== NRS_HOST_THREADS_SMALL=$n"
  NRS_DEBUG=HOST_THREADS_SMALL=$n python - <<'PY' 2>&1 | tail -2
import os,sys
for p in ("", "nr-slam_amd/py", "oracle", "tests"):
    sys.path.insert(0, os.path.join(os.getcwd(), p))
import bench
r = bench.tracked_fps(5000, 20, dense_graph=True, direct_solve=1)
print("%.2f frames/s, a2 %.2f ms, median frame %.2f ms" % (r["value"], r["ms_pose_and_deformation"], r["ms_per_frame_median"]))
PY
done
echo "== marks at 4 threads"
NRS_DEBUG=TIMING=1 python - <<'PY' 2>&1 | grep "\[nrs\]" | grep -E "engine_create|a2 engine" | tail -22
import os,sys
for p in ("", "nr-slam_amd/py", "oracle", "tests"):
    sys.path.insert(0, os.path.join(os.getcwd(), p))
import bench
r = bench.tracked_fps(5000, 8, dense_graph=True, direct_solve=1)
PY
