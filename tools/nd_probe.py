"""N1 where it could pay: a direct nested-dissection (multifrontal) solve of a2's single-frame system
(H + lambda I) x = b, OPT:148-557 -- pose (6) + N points (3 each), point-point coupling through the <= 11 graph neighbours of
a point (dampers +-s I3, springs q g g^T).  CPU prototype: builds the a2 block pattern of a synthetic frame the way the
driver does (GetEdges prefix walk with the reference's filters), orders it by recursive geometric bisection with vertex
separators (the pose last: an arrow), runs the symbolic factorisation on the separator tree and reports fill, flops, front
sizes per level and the critical path -- what a GPU implementation (one workgroup per front, levels in sequence, dense
fronts on v_mfma_f64_16x16x4) would have to execute per LM trial.  Also factorises numerically (NumPy dense fronts) and
checks the solve against a direct sparse solve.

    python tools/nd_probe.py [n_points] [leaf]
"""
import os, sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd", "py")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nrs_synth as S
import nrs_oracle as O

n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 32
tp = S.make_tracking_problem(n_points, 5)
g = tp["graph"]
ids = np.flatnonzero(tp["status"] == 0)
N = len(ids)
idx_of = -np.ones(n_points, int); idx_of[ids] = np.arange(N)
# OPT:224-337: per optimised point the first 11 accepted neighbours of its GetEdges list (duplicates skipped)
pairs = set()
reg = [set() for _ in range(N)]
for i, p in enumerate(ids):
    n_reg = 0
    for o, eid in O.graph_get_edges(g, int(p)):
        if n_reg > 10 or g["e_status"][eid] == 3:
            break
        j = idx_of[o]
        if j < 0 or j in reg[i]:
            continue
        reg[i].add(j); reg[j].add(i); pairs.add((min(i, j), max(i, j))); n_reg += 1
pairs = np.array(sorted(pairs))
E = len(pairs)
X = tp["X_prev"][ids].astype(np.float64)
print("frame: %d optimised points, %d point-point pairs (%.1f per point), unknowns %d" % (N, E, 2.0 * E / N, 6 + 3 * N))
adj = sp.coo_matrix((np.ones(2 * E), (np.r_[pairs[:, 0], pairs[:, 1]], np.r_[pairs[:, 1], pairs[:, 0]])), shape=(N, N)).tocsr()

# ---- nested dissection: recursive coordinate bisection, separator = vertices of the left half adjacent to the right half
order, tree = [], []          # tree nodes: (level, separator vertex list, children)


def dissect(v, level):
    if len(v) <= leaf:
        tree.append(dict(level=level, sep=v, kids=[])); return len(tree) - 1
    ext = X[v].max(0) - X[v].min(0)
    a = int(np.argmax(ext))
    o = v[np.argsort(X[v, a], kind="stable")]
    L, R = o[:len(o) // 2], o[len(o) // 2:]
    inR = np.zeros(N, bool); inR[R] = True
    sub = adj[L]
    touches = np.array([inR[sub.indices[sub.indptr[k]:sub.indptr[k + 1]]].any() for k in range(len(L))])
    sep, L2 = L[touches], L[~touches]
    kl = dissect(L2, level + 1) if len(L2) else None
    kr = dissect(R, level + 1) if len(R) else None
    tree.append(dict(level=level, sep=sep, kids=[k for k in (kl, kr) if k is not None]))
    return len(tree) - 1


sys.setrecursionlimit(10000)
root = dissect(np.arange(N), 0)
# elimination order: post-order of the tree (children before parents) = the order tree nodes were appended
perm = np.concatenate([t["sep"] for t in tree])
assert len(perm) == N and len(np.unique(perm)) == N
pos = np.empty(N, int); pos[perm] = np.arange(N)
# symbolic: boundary of a front = the not-yet-eliminated neighbours of its separator and of its children's boundaries
first = np.cumsum([0] + [len(t["sep"]) for t in tree])
for ti, t in enumerate(tree):
    own = set(t["sep"].tolist())
    bnd = set()
    for v in t["sep"]:
        bnd.update(int(u) for u in adj.indices[adj.indptr[v]:adj.indptr[v + 1]] if pos[u] >= first[ti + 1])
    for k in t["kids"]:
        bnd.update(u for u in tree[k]["bnd"] if u not in own)
    t["bnd"] = bnd
    s, b = 3 * len(own), 3 * len(bnd) + 6                         # + the pose block: every point couples to it
    t["s"], t["b"] = s, b
    t["flops"] = s ** 3 / 3.0 + s * s * b + s * b * b              # partial Cholesky of the front: factor, panel solve, Schur update
    t["lnz"] = s * (s + 1) / 2 + s * b
tot_flops = sum(t["flops"] for t in tree) + 6 ** 3 / 3.0
lnz = sum(t["lnz"] for t in tree) + 21
hnz = 3 * N * 2 + 9 * E + 18 * N + 21                              # lower triangle of H in scalars
print("leaf %d: %d fronts, nnz(L) %.2fM scalars (H: %.2fM, fill %.1fx), factorisation %.3f GFLOP" % (leaf, len(tree), lnz / 1e6, hnz / 1e6, lnz / hnz, tot_flops / 1e9))
depth = max(t["level"] for t in tree)
crit = 0.0
print("level: fronts, separator unknowns max / mean, boundary max, GFLOP of the level, largest front GFLOP")
for lv in range(depth + 1):
    fs = [t for t in tree if t["level"] == lv and t["s"] > 0]
    if not fs:
        continue
    big = max(fs, key=lambda t: t["flops"])
    crit += big["flops"]
    print("  %2d: %4d, %4d / %5.0f, %4d, %.4f, %.4f" % (lv, len(fs), max(t["s"] for t in fs), np.mean([t["s"] for t in fs]), max(t["b"] for t in fs),
                                                      sum(t["flops"] for t in fs) / 1e9, big["flops"] / 1e9))
print("critical path (largest front of every level, in sequence): %.3f GFLOP over %d levels" % (crit / 1e9, depth + 1))
# a GPU estimate: a front runs on ONE workgroup (4 waves, fp64 MFMA 16x16x4: 256 FLOP/cycle/CU peak = 0.61 TFLOP/s per CU at 2.4 GHz);
# dense partial Cholesky of 100-700 unknowns reaches a fraction of that (panel factorisations are latency-bound)
for eff in (0.1, 0.25):
    per_cu = 0.61e12 * eff
    print("  at %.0f %% of one CU's MFMA peak per front: critical path %.2f ms per factorisation; bulk (all CUs busy on the lower levels) %.2f ms"
          % (100 * eff, 1e3 * crit / per_cu, 1e3 * tot_flops / (256 * per_cu)))
# ---- numeric check of the ordering: a Laplacian-like SPD matrix with this block pattern, ND order vs natural order fill (SuperLU symmetric mode)
rng = np.random.default_rng(0)
rows, cols, vals = [], [], []
dg = np.full(3 * N, 1e-2)
for (i, j) in pairs:
    wv = rng.uniform(0.5, 2.0)
    for a in range(3):
        rows += [3 * i + a, 3 * j + a]; cols += [3 * j + a, 3 * i + a]; vals += [-wv, -wv]
        dg[3 * i + a] += wv; dg[3 * j + a] += wv
A = sp.coo_matrix((vals + dg.tolist(), (rows + list(range(3 * N)), cols + list(range(3 * N)))), shape=(3 * N, 3 * N)).tocsc()
p3 = (3 * perm[:, None] + np.arange(3)[None, :]).ravel()
for name, P in (("nested dissection", p3), ("natural (frame index)", np.arange(3 * N))):
    Ap = A[P][:, P]
    t0 = time.time()
    lu = spl.splu(Ap, permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    print("SuperLU %s order: nnz(L) %.2fM, factor %.0f ms on 1 core" % (name, lu.L.nnz / 1e6, 1e3 * (time.time() - t0)))
t0 = time.time()
lu = spl.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
print("SuperLU minimum degree: nnz(L) %.2fM, factor %.0f ms on 1 core" % (lu.L.nnz / 1e6, 1e3 * (time.time() - t0)))
