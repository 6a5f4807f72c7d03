"""The keyframe-block factorisation applied to a vector, M^-1 (H + lam I) x = x, with the pivot block swept in 16-pivot steps (default) and in
the 4-pivot register form (NRS_KFT_SCALAR_SWEEP=1): relative error of each against x.  usage: python tools/kft_sweep_ab.py [n k m seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nrs, nrs_synth as S
import embedded_oracle as E
a = [int(x) for x in sys.argv[1:]]
n, k, m, seed = a if len(a) == 4 else (260, 5, 36, 75)
p = S.make_dba_problem(n, k, seed)
flag, nb = S.embedded_problem(p, m)
e = nrs.dba_build_edges_embedded(p["kf_points"], flag, nb)
w = S.embedded_window(p, e)
cam = nrs.make_camera(p["model"], p["prm"])
qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
G, _ = E.dba_graph_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"],
                            e["dm_idx"], e["dm_w"], w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"])
G.initialize(0); G.compute_active_errors()
H, b = G.build_system(); H = H.tocsr()
c = nrs.Context(embedded_solver=1)
c.dba_upload_embedded(cam, qt, w, e, p["scale"])
print(c.debug_kft_info())
rng = np.random.default_rng(k)
x = rng.normal(0, 1, H.shape[0])
for scl in (1e-5, 1e-7, 1e-9):
    lam = scl * np.abs(H.diagonal()).max()
    r = H @ x + lam * x
    for name, v in (("16-pivot steps", None), ("4-pivot register form", "1")):
        nrs.debug_set("NRS_KFT_SCALAR_SWEEP", v)
        try:
            u = c.debug_kft_apply(lam, r)
            print("lam = %.0e max diag  %-22s |u - x| / |x| = %.3e" % (scl, name, np.linalg.norm(u - x) / np.linalg.norm(x)))
        except Exception as ex:
            print("lam = %.0e max diag  %-22s %s" % (scl, name, ex))
    nrs.debug_set("NRS_KFT_SCALAR_SWEEP", None)
