"""Speculative LM trials of a2 (nrs_engine_types.hpp SpecSet) against one trial at a time: bench.py's tracked-fps leg under debug
switches, one line per setting.
  python tools/spec_trials_probe.py <frames> <dense graph 0|1> <points> SETTING...      SETTING = NAME=VALUE[,NAME=VALUE...]
e.g. python tools/spec_trials_probe.py 30 1 5000 NRS_SPEC_TRIALS=0 NRS_SPEC_TRIALS=1 NRS_SPEC_TRIALS=3 NRS_SPEC_TRIALS=3,NRS_SPEC_FIXED=4
NRS_SPEC_DBG=1 prints the host clocks of every batch (enqueue of each set, arrival of each result) on stderr."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("", "nr-slam_amd/py", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import bench
import nrs
fr = int(sys.argv[1]); dense = int(sys.argv[2]); npts = int(sys.argv[3])
for env in sys.argv[4:]:
    nrs.debug_clear()
    for kv in env.split(","):
        k, v = kv.split("=")
        nrs.debug_set(k, v)
    r = bench.tracked_fps(npts, fr, dense_graph=bool(dense))
    print("%s: %.2f frames/s (median %.2f ms), a2 %.2f ms, trials/frame %.1f, tracked %d" % (env, r["value"], r["ms_per_frame_median"], r["ms_pose_and_deformation"],
                                                                                        r["lm_trials_per_frame"], r["tracked_last_frame"]), flush=True)
