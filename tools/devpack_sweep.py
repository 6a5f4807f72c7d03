"""Seeded sweep: device-side problem construction + device edge construction against the host forms on random windows
(sizes, keyframe counts, dropout, camera model drawn per seed): pack checksums, edge lists and solve results must be identical.
    python tools/devpack_sweep.py [n_seeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad, dev, t0 = 0, 0, time.time()
for seed in range(n_seeds):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(300, 9000)); k = int(rng.integers(2, 14)); model = int(rng.integers(0, 2))
    p = S.make_dba_problem(n, k, 100 + seed, model, dropout=float(rng.uniform(0.0, 0.4)))
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"]); qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    res = []
    for host in (False, True):
        if host: nrs.debug_set("NRS_HOST_PACK", "1")
        else: nrs.debug_set("NRS_HOST_PACK", None)
        c = nrs.Context()
        c.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
        h = c.dba_pack_hash(); c.dba_optimize(3); out = c.dba_download(); c.close()
        res.append((h, out))
    nrs.debug_set("NRS_HOST_PACK", None)
    c = nrs.Context()
    wq, wx = c.dba_solve_window(cam, qt, p["kf_points"], p["lm_xyz"], p["lm_uv"], p["nbr"], p["scale"], 3)
    ed = c.dba_window_edges(); c.close()
    (hd, od), (hh, oh) = res
    same_pack = all(hd[i] == hh[i] for i in range(24) if i != 21)
    same_solve = np.array_equal(od[0], oh[0]) and np.array_equal(od[1], oh[1])
    same_edges = ed is None or all(np.array_equal(ed[key], e[key]) for key in ("sp_ij", "sp_d0", "dm_idx", "dm_w"))
    same_window = np.array_equal(wq, od[0]) and np.array_equal(wx, od[1].astype(np.float32))
    dev += hd[21]
    if not (same_pack and same_solve and same_edges and same_window):
        bad += 1
        print("MISMATCH seed", seed, n, k, model, same_pack, same_solve, same_edges, same_window, flush=True)
print("devpack_sweep: %d windows (%d built on the device), %d mismatches, %.0f s" % (n_seeds, dev, bad, time.time() - t0))
