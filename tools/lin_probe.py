"""Scratch probe: lineariser / operator timings (HIP events, nrs_options.profile) on a named config."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py")); sys.path.insert(0, ROOT)
import numpy as np, nrs, nrs_synth as S
import bench
for w in sys.argv[1:] or ["C2"]:
    n_points, n_kf, seed, model = S.CONFIGS[w]
    p = S.make_dba_problem(n_points, n_kf, seed, model)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"]); qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    ctx = nrs.Context(profile=1)
    ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    ctx.dba_optimize(5); ctx.reset_profile(); ctx.dba_reset(); ctx.dba_optimize(5)
    pr = ctx.profile(); ctx.close()
    n_lm, n_sp, n_dm = len(p["lm_kf"]), len(e["sp_ij"]), len(e["dm_idx"])
    lin_b, spmv_b = bench.algorithmic_bytes(n_lm, n_sp, n_dm, bench.unique_blocks(n_lm, e["sp_ij"], e["dm_idx"]))
    lu = 1e3 * pr["linearize_ms"] / max(1, pr["linearize_launches"]); su = 1e3 * pr["spmv_ms"] / max(1, pr["spmv_launches"])
    ctx = nrs.Context()
    ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    ctx.dba_optimize(5)
    ts = []
    for r in range(5 if w in ("C2", "C3") else 2):
        ctx.dba_reset(); tr = nrs.Trace(); t0 = time.perf_counter(); ctx.dba_optimize(5, tr); ts.append(time.perf_counter() - t0)
    ctx.close()
    print(json.dumps(dict(workload=w, lin_us=lu, lin_frac=lin_b / (lu * 1e-6) / 8e12, lin_bytes=lin_b, spmv_us=su, spmv_frac=spmv_b / (su * 1e-6) / 8e12,
                          ms_per_step=1e3 * min(ts), trials=len(tr.trials), pcg=sum(t["inner"] for t in tr.trials))), flush=True)
