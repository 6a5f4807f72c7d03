"""Embedded BA window: the keyframe-block factorisation (nrs_options.embedded_solver = 0) against the block-Jacobi PCG (= 2): LM trace, inner
iterations, LM iterations / s.  usage: python tools/kft_probe.py [n_points n_nodes n_kf] [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import nrs, nrs_synth as S
a = [int(x) for x in sys.argv[1:]]
n, m, k = a[:3] if len(a) >= 3 else (5000, 500, 20)
reps = a[3] if len(a) > 3 else 5
p = S.make_dba_problem("C2") if (n, k) == (5000, 20) else S.make_dba_problem(n, k, 53)
flag, nb = S.embedded_problem(p, m)
e = nrs.dba_build_edges_embedded(p["kf_points"], flag, nb)
w = S.embedded_window(p, e)
cam = nrs.make_camera(p["model"], p["prm"])
qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
res = {}
for mode in (1, 2):
    ctx = nrs.Context(embedded_solver=mode, exact_trials=1)
    t0 = time.perf_counter()
    ctx.dba_upload_embedded(cam, qt, w, e, p["scale"])
    t_up = time.perf_counter() - t0
    tr = nrs.Trace()
    ctx.dba_optimize(5, tr)
    ts = []
    for _ in range(reps):
        ctx.dba_reset()
        t0 = time.perf_counter()
        ctx.dba_optimize(5)
        ts.append(time.perf_counter() - t0)
    pq, xyz = ctx.dba_download()
    res[mode] = (pq, xyz, tr)
    print("embedded_solver %d: upload %.1f ms, optimize(5) %.2f ms (min of %d) = %.1f LM it/s; trials: %s" % (
        mode, 1e3 * t_up, 1e3 * min(ts), reps, tr.iterations / min(ts), [(t["accepted"], t["inner"], "%.6e" % t["chi_new"]) for t in tr.trials]), flush=True)
    ctx.close()
print("max |pose diff| %.3e  max |node diff| %.3e" % (np.abs(res[1][0] - res[2][0]).max(), np.abs(res[1][1] - res[2][1]).max()))
