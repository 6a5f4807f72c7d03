"""Scratch probe: the embedded-deformation frame alone (N2: n points x m nodes, every observation in the problem), for kernel traces.
usage: embedded_probe.py [n] [m] [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nrs
import nrs_synth as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 500
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
tp = S.make_tracking_problem(n, 3)
cam = nrs.make_camera(tp["model"], tp["prm"])
ctx = nrs.Context()
fm = np.arange(n, dtype=np.int32)
nodes = ctx.skin_select_nodes(tp["X_prev"], m, tp["status"] == 0)
node = np.zeros(n, np.uint8)
node[nodes] = 1
g = nrs.RGraph(ctx, n, tp["graph"]["sigma"], tp["graph"]["stretch_th"])
ms = []
for rep in range(reps):
    g.add_edges(tp["X_prev"], fm, fm)
    tr = nrs.Trace(1024)
    t0 = time.perf_counter()
    r = ctx.track_deform_solve_embedded(cam, g, tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], node, tp["pose_q"], tp["pose_t"], tp["scale"], tr)
    ms.append(1e3 * (time.perf_counter() - t0))
print("n %d nodes %d: %.2f ms per call (median of %d after the first), %d LM trials" % (n, m, float(np.median(ms[1:])), reps - 1, len(tr.trials)))
g.close()
ctx.close()
