"""a2 (nrs_track_deform_solve) per frame size on both linear solvers: the nested-dissection Cholesky (nrs_options.direct_solve = 1) and the
PCG (= 2); decides the size window of the default (direct_solve = 0).   python tools/nd_crossover.py [--dense] [n ...]
(--dense: on the device-resident all-pairs graph, nrs_track_deform_solve_rg, as bench.py's tracked_fps)"""
import os, sys, time
os.environ["NRS_ND_NO_CACHE"] = "1"          # the frame is repeated: every call builds its symbolic factorisation, as a new frame does
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
ctxs = {"direct": nrs.Context(direct_solve=1), "pcg": nrs.Context(direct_solve=2)}
dense = "--dense" in sys.argv
for n in [int(a) for a in sys.argv[1:] if a != "--dense"] or [150, 300, 600, 1150, 1700, 2500, 3500, 5000]:
    tp = S.make_tracking_problem(n, 5)
    cam = nrs.make_camera(tp["model"], tp["prm"]); fm = np.arange(n, dtype=np.int32)
    out = {}
    for name, ctx in ctxs.items():
        best = 1e9
        for rep in range(4):
            if dense:
                g = nrs.RGraph(ctx, n, tp["graph"]["sigma"], tp["graph"]["stretch_th"]); g.add_edges(tp["X_prev"], fm, fm)
            tr = nrs.Trace(1024); t0 = time.perf_counter()
            if dense:
                r = ctx.track_deform_solve_rg(cam, g, tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tr, 128)
            else:
                r = ctx.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tr)
            best = min(best, time.perf_counter() - t0)
            if dense:
                g.close()
        out[name] = (best, len(tr.trials), sum(x["inner"] for x in tr.trials), r)
    d, p = out["direct"], out["pcg"]
    same = np.array_equal(d[3]["f_status"], p[3]["f_status"]) and d[3]["lost"] == p[3]["lost"]
    print("n %5d tracked %5d: direct %6.1f ms (%d trials) | pcg %6.1f ms (%d trials, %d iterations) | pcg/direct %.2f | same statuses %s, max |dpos| %.1e" % (
        n, int((tp["status"] == 0).sum()), 1e3 * d[0], d[1], 1e3 * p[0], p[1], p[2], p[0] / d[0], same, np.abs(d[3]["f_pos"] - p[3]["f_pos"]).max()), flush=True)
