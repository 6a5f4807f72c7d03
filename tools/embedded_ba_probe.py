"""N2b probe: the embedded form of a BA window (points x nodes x keyframes) resident on the GPU -- LM iterations / s, trials, PCG iterations,
next to the plain window of the same scene.  usage: python tools/embedded_ba_probe.py [config, default C2] [n_nodes, default 500]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
m = int(sys.argv[2]) if len(sys.argv) > 2 else 500
p = S.make_dba_problem(name)
flag, nb = S.embedded_problem(p, m)
t0 = time.perf_counter()
e = nrs.dba_build_edges_embedded(p["kf_points"], flag, nb)
t_build = time.perf_counter() - t0
w = S.embedded_window(p, e)
cam = nrs.make_camera(p["model"], p["prm"]); qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
for exact in (0, 1):
    ctx = nrs.Context(exact_trials=exact)
    t0 = time.perf_counter(); ctx.dba_upload_embedded(cam, qt, w, e, p["scale"]); t_up = time.perf_counter() - t0
    ctx.dba_optimize(5)
    ts, its = [], 0
    for r in range(5):
        ctx.dba_reset(); tr = nrs.Trace(64); t0 = time.perf_counter(); ctx.dba_optimize(5, tr); ts.append(time.perf_counter() - t0); its = tr.iterations
    ctx.close()
    print(json.dumps(dict(workload="%s embedded, %d nodes" % (name, m), exact_trials=exact, node_copies=len(e["lm_obs"]), skinned_obs=len(e["sk_obs"]), springs=len(e["sp_ij"]),
                          dampers=len(e["dm_idx"]), build_ms=1e3 * t_build, upload_ms=1e3 * t_up, ms_per_optimize5=1e3 * min(ts), lm_iters_per_s=its / min(ts), trials=len(tr.trials),
                          pcg=sum(t["inner"] for t in tr.trials), chi0=tr.trials[0]["chi"], chi_end=[t for t in tr.trials if t["accepted"]][-1]["chi_new"])), flush=True)
