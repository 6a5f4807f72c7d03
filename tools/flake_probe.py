"""Stress: nrs_dba_solve_window against nrs_dba_build_edges + nrs_dba_solve on the reference-sized window, trial by trial."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
p = S.make_dba_problem(int(os.environ.get('FP_N', '5000')), int(os.environ.get('FP_K', '5')), 1, S.PINHOLE)
e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
cam = nrs.make_camera(p["model"], p["prm"])
qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
ref = None
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    c = nrs.Context(exact_trials=int(os.environ.get('FP_EXACT', '0')))
    tr = nrs.Trace(64)
    if mode == "window" or (mode == "both" and rep % 2 == 0):
        c.dba_solve_window(cam, qt, p["kf_points"], p["lm_xyz"], p["lm_uv"], p["nbr"], p["scale"], 5, tr)
    else:
        out = c.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], 5, tr)
    c.close()
    t = [(x["accepted"], x["inner"], x["early"], x["lam"], x["chi"], x["chi_new"]) for x in tr.trials]
    if ref is None:
        ref = t
        ref_out = out if mode == 'solve' else None
    elif t != ref:
        for i, (a, b) in enumerate(zip(t, ref)):
            if a != b:
                print("rep %d trial %d: %s != %s" % (rep, i, a, b), flush=True)
                if mode == 'solve': print('   final diff: poses %.3e points %.3e; trials equal after this one: %s' % (np.abs(out[0] - ref_out[0]).max(), np.abs(out[1] - ref_out[1]).max(), t[i + 1:] == ref[i + 1:]), flush=True)
                break
print("done", flush=True)
