"""N2b: would the nested-dissection direct solver (nrs_nd_plan.hpp, built for a2's single-frame systems) take the embedded BA window?
Builds the window's block structure (node copies + two halves per pose; pairs from springs, dampers, skinned node sets, pose couplings)
and asks the HOST plan builder (oracle/nd_host.cpp = csrc/nrs_nd_plan.hpp) for fronts / levels / flops; the third coordinate handed to the
dissection is the keyframe index scaled by tscale (the plan only bisects coordinates: any choice is valid, some are cheaper).
usage: python tools/proto/embedded_ba_nd_plan_probe.py [config|n_points] [n_nodes] [n_kf]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import nrs, nrs_cpu as CPU, nrs_synth as S
a = sys.argv[1:]
if a and not a[0].isdigit():
    p = S.make_dba_problem(a[0]); m = int(a[1]) if len(a) > 1 else 500
else:
    n, m, k = (int(x) for x in a[:3]) if len(a) >= 3 else (1500, 150, 6)
    p = S.make_dba_problem(n, k, 7)
flag, nb = S.embedded_problem(p, m)
e = nrs.dba_build_edges_embedded(p["kf_points"], flag, nb)
w = S.embedded_window(p, e)
nl, K = len(e["lm_obs"]), p["n_kf"]
pairs = set()
def add(i, j):
    if i != j: pairs.add((min(i, j), max(i, j)))
for i, j in e["sp_ij"]: add(int(i), int(j))
for d in e["dm_idx"]:
    for x in range(4):
        for y in range(x + 1, 4): add(int(d[x]), int(d[y]))
for row in e["sk_node"]:
    r = [int(v) for v in row if v >= 0]
    for x in range(len(r)):
        for y in range(x + 1, len(r)): add(r[x], r[y])
# poses: two halves each, coupled to each other and to every node copy of the keyframe (own observation or skinned observations)
kf = np.asarray(w["lm_kf"])
for c in range(K):
    add(nl + 2 * c, nl + 2 * c + 1)
    for i in np.where(kf == c)[0]:
        add(int(i), nl + 2 * c); add(int(i), nl + 2 * c + 1)
pairs = np.array(sorted(pairs), np.int32)
X = np.asarray(w["lm_xyz"], np.float64)
for tscale in (0.0, 0.25, 1.0, 4.0):
    ext = X.max(0) - X.min(0)
    pos = X.copy()
    if tscale > 0:
        pos[:, int(np.argmin(ext))] = kf * tscale * ext.max() / K                 # the flattest axis carries the keyframe index
    pos = np.vstack([pos, np.zeros((2 * K, 3))])
    last = np.zeros(nl + 2 * K, np.uint8); last[nl:] = 1
    n = len(pos)
    Dn = np.tile(np.eye(3) * 1e3, (n, 1, 1)); Vp = np.zeros((len(pairs), 3, 3)); bn = np.ones((n, 3))
    t0 = time.time()
    try:
        ok, x, st = CPU.nd_solve(pos, last, pairs, Dn, Vp, bn, 0.0)
    except RuntimeError as ex:
        print("tscale %.2f: %s" % (tscale, ex), flush=True); continue
    print("tscale %.2f: %d nodes, %d pairs -> %s  (%.1f GFLOP per factorisation; host plan + reference solve %.1f s)" % (tscale, n, len(pairs), st, st["flops"] / 1e9, time.time() - t0), flush=True)
