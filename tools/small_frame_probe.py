"""a2 (nrs_track_deform_solve) on single frames of several sizes: ms per frame, LM trials, linear-solver iterations.  The frame is
repeated, so the time is reported twice: `ms` with the direct solver's plan cache off (every call builds its symbolic
factorisation, as a frame with a new structure does) and `ms_plan_reused` with it on (a frame whose structure equals an earlier one's)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
ctx = nrs.Context()
for n in [int(a) for a in sys.argv[1:]] or [600, 1150, 2500, 5000]:
    tp = S.make_tracking_problem(n, 5)
    cam = nrs.make_camera(tp["model"], tp["prm"]); fm = np.arange(n, dtype=np.int32)
    def run(reps):
        global tr
        b = 1e9
        for rep in range(reps):
            tr = nrs.Trace(1024); t0 = time.perf_counter()
            ctx.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tr)
            b = min(b, time.perf_counter() - t0)
        return b
    nrs.debug_set("NRS_ND_NO_CACHE", "1")
    best = run(3)
    nrs.debug_set("NRS_ND_NO_CACHE", None)
    best_hit = run(3)
    inner = [x["inner"] for x in tr.trials]
    per_round = {}
    for x in tr.trials: per_round[x["round"]] = per_round.get(x["round"], 0) + x["inner"]
    print("n %d tracked %d: %.1f ms (plan reused: %.1f ms), trials %d, pcg %d (per round %s), max inner %d, us/iter all-in %.1f" % (
        n, int((tp["status"] == 0).sum()), 1e3 * best, 1e3 * best_hit, len(inner), sum(inner), per_round, max(inner), 1e6 * best / max(1, sum(inner))), flush=True)
    print("   accepted-trial iterations:", [x["inner"] for x in tr.trials if x["accepted"]])
