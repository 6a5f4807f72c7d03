"""Scratch probe: M independent C2 windows solved concurrently on ONE GPU (one context + stream + host thread each):
aggregate LM iterations/s against M."""
import sys, os, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
n_points, n_kf, seed, model = S.CONFIGS["C2"]
steps = 20
for M in (1, 2, 3, 4, 6, 8):
    ctxs = []
    for m in range(M):
        p = S.make_dba_problem(n_points, n_kf, seed + 1000 * m, model)
        e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
        cam = nrs.make_camera(p["model"], p["prm"])
        qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
        c = nrs.Context()
        c.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
        c.dba_optimize(5)
        ctxs.append(c)
    its = [0] * M
    bar = threading.Barrier(M + 1)
    def work(m):
        bar.wait()
        for _ in range(steps):
            ctxs[m].dba_reset(); tr = nrs.Trace(64); ctxs[m].dba_optimize(5, tr); its[m] += tr.iterations
        bar.wait()
    th = [threading.Thread(target=work, args=(m,)) for m in range(M)]
    for t in th: t.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    for t in th: t.join()
    print("M=%d: %.0f LM iters/s aggregate, %.2f ms per optimize(5) per window" % (M, sum(its) / dt, 1e3 * dt / steps))
    for c in ctxs: c.close()
