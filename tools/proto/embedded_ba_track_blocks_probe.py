"""Embedded C2: iteration counts per LM iteration for block-Jacobi (P0), per-node-TRACK blocks (all keyframe copies of a node, poses 6x6) (PT),
and symmetric multiplicative combos."""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import embedded_oracle as E, nrs_synth as S, nrs_oracle as O
p = S.make_dba_problem("C2"); k = 20; m = 500
flag, nb = S.embedded_problem(p, m)
e = E.dba_build_embedded(p["kf_points"], flag, nb["rowptr"], nb["col"], nb["w"], nb["d0"], nb["status"])
w = S.embedded_window(p, e)
G, skn = E.dba_graph_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"],
                              w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"])
G.initialize(0)
K6 = 6 * k
lm_pt = np.asarray(e["lm_pt"])
def inv_groups(A, groups, N):
    invs = [(g, np.linalg.inv(A[g][:, g].toarray())) for g in groups]
    def mv(v):
        o = np.zeros_like(v)
        for g, Bi in invs: o[g] = Bi @ v[g]
        return o
    return mv
def pcg(A, b, mv, N, maxiter=6000):
    it = [0]
    def cb(x): it[0] += 1
    x, info = spla.cg(A, b, rtol=1e-10, maxiter=maxiter, M=spla.LinearOperator((N, N), matvec=mv), callback=cb)
    return x, it[0]
lam, ni = -1, 2.0
for it in range(4):
    G.compute_active_errors(); chi = G.active_robust_chi2()
    H, b = G.build_system(); N = H.shape[0]
    if it == 0: lam = 1e-5 * float(np.max(np.abs(H.diagonal())))
    A = (H + lam * sp.identity(N)).tocsr()
    nn = (N - K6) // 3
    poses = [np.arange(6 * i, 6 * i + 6) for i in range(k)]
    bj = poses + [np.arange(K6 + 3 * i, K6 + 3 * i + 3) for i in range(nn)]
    tracks = poses + [np.concatenate([np.arange(K6 + 3 * i, K6 + 3 * i + 3) for i in np.where(lm_pt == q)[0]]) for q in np.unique(lm_pt)]
    t1 = time.time()
    m0 = inv_groups(A, bj, N); mt = inv_groups(A, tracks, N)
    x, n0 = pcg(A, b, m0, N)
    _, nt = pcg(A, b, mt, N)
    print("LM it %d lam %.4g: P0 %d | PT tracks %d   (%.0f s)" % (it, lam, n0, nt, time.time() - t1), flush=True)
    G.push(); G.update(x); G.compute_active_errors()
    temp = G.active_robust_chi2(); scale = float(np.dot(x, lam * x + b)) + 1e-3; rho = (chi - temp) / scale
    if rho > 0:
        alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0); lam *= max(1.0 / 3.0, alpha)
    else:
        lam *= ni; ni *= 2; G.pop()
