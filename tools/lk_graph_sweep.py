"""Sweep: LK tracker (a21-a23) and graph kernels (a19/a20) against the oracle, bit-exact, many seeds."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, nrs, nrs_synth as S, nrs_oracle as O, lk_oracle as LK
ctx = nrs.Context()
nseed = int(sys.argv[1]) if len(sys.argv) > 1 else 10
viol = 0
def check(cond, msg):
    global viol
    if not cond:
        viol += 1; print("VIOLATION:", msg, flush=True)
t0 = time.time()
for seed in range(nseed):
    rng = np.random.default_rng(500 + seed)
    # ---- LK
    n = int(rng.integers(30, 200)); flow = float(rng.uniform(1.0, 20.0))
    sq = S.make_lk_sequence(n, 700 + seed, flow_px=flow)
    ctx.klt_configure(); ctx.klt_set_reference(sq["im0"], sq["pts"])
    lk = LK.LucasKanadeOracle(); lk.set_reference(sq["im0"], sq["pts"])
    st = np.zeros(len(sq["pts"]), np.int32); st[::13] = 3
    guess = sq["pts"] + np.float32(rng.uniform(-1.5, 1.5))
    init = bool(seed % 2)
    xy, st2, good, ssim = ctx.klt_track(sq["im1"], guess, st, initial_flow=init, min_ssim=0.8)
    oxy, ost, ogood, ossim = lk.track(sq["im1"], guess.copy(), st, initial_flow=init, min_ssim=0.8)
    tag = "LK seed %d (%d pts, flow %.1f, init %d)" % (seed, len(sq["pts"]), flow, init)
    check(np.array_equal(st2, ost) and good == ogood, tag + " status")
    check(np.array_equal(xy, oxy), tag + " positions (max diff %.3g)" % np.abs(xy - oxy).max())
    ok = np.isin(ost, (0, 1, 2))
    check(np.allclose(ssim[ok], ossim[ok], atol=1e-5), tag + " ssim")
    # ---- graph kernels
    n = int(rng.integers(40, 3000))
    sc = S.make_scene(n, 2, 900 + seed)
    G = sc["graph"]
    G["e_status"][rng.uniform(size=len(G["e_status"])) < 0.15] = S.GRAPH_BAD
    G["e_status"][rng.uniform(size=len(G["e_status"])) < 0.05] = 0
    G["e_w"][rng.uniform(size=len(G["e_w"])) < 0.2] *= np.float32(0.35)
    rp, col, eid = ctx.graph_select_neighbours(G)
    ref = S.ordered_neighbours(G)
    check(np.array_equal(rp, ref["o_rowptr"]) and np.array_equal(col, ref["o_col"]) and np.array_equal(eid, ref["o_eid"]), "graph select seed %d (%d pts)" % (seed, n))
    for p in rng.choice(n, 5, replace=False):
        lst = O.graph_get_edges(G, int(p))
        check([c for c, _ in lst] == col[rp[p]:rp[p + 1]].tolist(), "graph GetEdges row %d seed %d" % (p, seed))
    pos = sc["X0"] + rng.normal(0, 0.02, sc["X0"].shape).astype(np.float32)
    ids = np.sort(rng.choice(n, max(1, n // 2), replace=False)).astype(np.int32)
    g2, goodv = ctx.graph_update(G, pos, ids)
    refg = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in G.items()}
    good_ref = [O.graph_update_vertex_flat(refg, int(p), pos) for p in ids]
    check(np.array_equal(goodv, np.array(good_ref)), "graph update flags seed %d" % seed)
    for k in ("e_status", "e_max", "e_min", "e_w"):
        check(np.array_equal(g2[k], refg[k]), "graph update %s seed %d" % (k, seed))
print("seeds %d, violations %d, %.0f s" % (nseed, viol, time.time() - t0))
