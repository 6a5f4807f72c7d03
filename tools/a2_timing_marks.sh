cd /root/repo
NRS_DEBUG=TIMING=1 python - <<'PY' 2>&1 | grep "\[nrs\]" | tail -60
import os,sys
for p in ("", "nr-slam_amd/py", "oracle", "tests"):
    sys.path.insert(0, os.path.join("/root/repo", p))
import bench
r = bench.tracked_fps(5000, 8, dense_graph=True, direct_solve=1)
PY
