"""PCIe-inclusive one-shot timings (host buffers in, host buffers out) for DESIGN.md section 6."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
ctx = nrs.Context()
for name, args in (("C2 (20 kf)", ("C2",)), ("reference window (5 kf x 5k pts)", (5000, 5, 1))):
    p = S.make_dba_problem(*args)
    t0 = time.perf_counter(); e = nrs.dba_build_edges(p["kf_points"], p["nbr"]); t1 = time.perf_counter()
    cam = nrs.make_camera(p["model"], p["prm"]); qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    for rep in range(3):
        t2 = time.perf_counter(); ctx.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], 5); t3 = time.perf_counter()
    ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    t4 = time.perf_counter(); ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"]); t5 = time.perf_counter()
    ctx.dba_optimize(5); t6 = time.perf_counter()
    print("%s: landmarks %d, build_edges %.1f ms, one-shot nrs_dba_solve %.1f ms (upload+pack %.1f ms, optimize %.1f ms)" % (
        name, len(p["lm_kf"]), 1e3*(t1-t0), 1e3*(t3-t2), 1e3*(t5-t4), 1e3*(t6-t5)))
