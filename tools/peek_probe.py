"""Scratch probe: gain ratio seen at the PCG milestones vs the final one (NRS_PEEK_DEBUG=1), single frame a2."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = nrs.Context()
tp = S.make_tracking_problem(n, seed)
cam = nrs.make_camera(tp["model"], tp["prm"])
fm = np.arange(n, dtype=np.int32)
tr = nrs.Trace(1024)
ctx.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tr)
