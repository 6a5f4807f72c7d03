"""Sweep: default (early-rejecting) trials vs exact trials on many seeded problems.  For every
problem the accept/reject sequence, lambdas and final state must agree; prints the mismatches."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
ctx = nrs.Context()
ctx_exact = nrs.Context(exact_trials=1)
bad = 0; total = 0; early = 0; trials = 0
def seq(tr): return [(t["round"], t["iter"], t["trial"], bool(t["accepted"])) for t in tr.trials]
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    rng = np.random.default_rng(seed)
    # --- BA windows of varied size
    n, k = int(rng.integers(150, 1500)), int(rng.integers(2, 7))
    model = S.PINHOLE if seed % 3 else S.KB8
    p = S.make_dba_problem(n, k, 1000 + seed, model)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"]); qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    out = []
    for c in (ctx, ctx_exact):
        tr = nrs.Trace(256)
        pq, xyz = c.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], 8, tr)
        out.append((pq, xyz, tr))
    total += 1
    trials += len(out[0][2].trials); early += sum(1 for t in out[0][2].trials if t["early"])
    ok = seq(out[0][2]) == seq(out[1][2]) and np.allclose(out[0][0], out[1][0], atol=1e-7) and np.allclose(out[0][1], out[1][1], atol=1e-6)
    if not ok:
        bad += 1; print("BA mismatch seed", seed, n, k)
    # --- tracking frames
    n = int(rng.integers(300, 3000))
    tp = S.make_tracking_problem(n, 2000 + seed, model)
    cam = nrs.make_camera(tp["model"], tp["prm"]); fm = np.arange(n, dtype=np.int32)
    out = []
    for c in (ctx, ctx_exact):
        tr = nrs.Trace(1024)
        r = c.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tr)
        out.append((r, tr))
    total += 1
    trials += len(out[0][1].trials); early += sum(1 for t in out[0][1].trials if t["early"])
    r0, r1 = out[0][0], out[1][0]
    ok = seq(out[0][1]) == seq(out[1][1]) and r0["lost"] == r1["lost"] and np.allclose(r0["f_pos"], r1["f_pos"], atol=1e-6) and np.array_equal(r0["f_status"], r1["f_status"])
    if not ok:
        bad += 1; print("a2 mismatch seed", seed, n, "lost equal", r0["lost"] == r1["lost"], "status equal", np.array_equal(r0["f_status"], r1["f_status"]), "max dpos", np.abs(r0["f_pos"] - r1["f_pos"]).max())
        for a, b in zip(out[0][1].trials, out[1][1].trials):
            if (a["round"], a["iter"], a["trial"], a["accepted"]) != (b["round"], b["iter"], b["trial"], b["accepted"]) or abs(a["lam"] - b["lam"]) > 1e-9 * b["lam"]:
                print("   first difference: default", {k: a[k] for k in ("round", "iter", "trial", "accepted", "early", "inner", "lam", "chi", "chi_new", "rho")})
                print("                     exact  ", {k: b[k] for k in ("round", "iter", "trial", "accepted", "early", "inner", "lam", "chi", "chi_new", "rho")})
                break
        else:
            print("   traces identical over the common prefix; lengths", len(out[0][1].trials), len(out[1][1].trials))
print("problems %d, mismatches %d, trials %d, early-rejected %d" % (total, bad, trials, early))
