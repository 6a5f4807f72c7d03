"""Scratch: lineariser timing with a piece removed (NRS_LIN_EXP = 1 no factor stores, 2 no neighbour gathers (own row), 3 no record loads, 4 no loops; round 5, right results except 5: 5 half the damper slots, 6 non-temporal streams, 7 / 8 two waves per SIMD with 8 / 10-slot
request batches, 9 three waves with 6, 10 = 7 + 6, 11 the product kernel through the probe switch).  Needs make PROBES=1."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
w = sys.argv[1] if len(sys.argv) > 1 else "C4"
n_points, n_kf, seed, model = S.CONFIGS[w]
p = S.make_dba_problem(n_points, n_kf, seed, model)
e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
ctx = nrs.Context(profile=1)
ctx.dba_upload(nrs.make_camera(p["model"], p["prm"]), np.concatenate([p["poses_q"], p["poses_t"]], 1), p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
for exp in (sys.argv[2:] or ("0", "1", "2", "3", "4", "5", "6", "7", "8", "9", "10", "11")):
    if exp == "0": nrs.debug_set("NRS_LIN_EXP", None)
    else: nrs.debug_set("NRS_LIN_EXP", exp)
    ctx.dba_reset(); ctx.reset_profile()
    try:
        ctx.dba_optimize(1)
    except Exception as ex:
        pass
    pr = ctx.profile()
    print(w, "EXP", exp, "lineariser %.1f us" % (1e3 * pr["linearize_ms"] / max(1, pr["linearize_launches"])), flush=True)
ctx.close()
