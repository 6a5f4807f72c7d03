"""Scratch probe: one oracle_sweep BA seed, GPU (default and exact trials) against the oracle, trial by trial."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, nrs, nrs_synth as S, nrs_oracle as O
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 188
rng = np.random.default_rng(100 + seed)
model = S.PINHOLE if seed % 3 else S.KB8
n, k = int(rng.integers(60, 260)), int(rng.integers(1, 5))
p = S.make_dba_problem(n, k, 3000 + seed, model)
e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
cam = nrs.make_camera(p["model"], p["prm"]); qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
otr = []
oq, ot, opts, nit = O.dba_solve(p["model"], p["prm"], p["poses_q"], p["poses_t"], p["lm_xyz"], p["lm_kf"], p["lm_uv"], e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"], p["scale"], 5, otr)
print("oracle :", [(t["iter"], t["trial"], bool(t["accepted"]), "%.4g" % t["rho"], "%.6e" % t["chi"]) for t in otr])
for name, c in (("default", nrs.Context()), ("exact  ", nrs.Context(exact_trials=1))):
    tr = nrs.Trace(256)
    pq, xyz = c.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], 5, tr)
    print(name, ":", [(t["iter"], t["trial"], bool(t["accepted"]), "%.4g" % t["rho"], "%.6e" % t["chi"], t["inner"], int(t["early"])) for t in tr.trials])
    print("   max |dpose|", np.abs(pq[:, :4] - oq).max(), np.abs(pq[:, 4:] - ot).max(), "max |dxyz|", np.abs(xyz - opts).max())
