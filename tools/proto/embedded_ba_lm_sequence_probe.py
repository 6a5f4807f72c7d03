"""Full-size embedded C2: PCG iteration counts per LM iteration with block-Jacobi, fresh per-keyframe exact blocks, and stale ones."""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import embedded_oracle as E, nrs_synth as S, nrs_oracle as O
n, m, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (5000, 500, 20)
t0 = time.time()
p = S.make_dba_problem(n, k, 11) if n != 5000 else S.make_dba_problem("C2")
flag, nb = S.embedded_problem(p, m)
e = E.dba_build_embedded(p["kf_points"], flag, nb["rowptr"], nb["col"], nb["w"], nb["d0"], nb["status"])
w = S.embedded_window(p, e)
G, skn = E.dba_graph_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"],
                              w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"])
G.initialize(0)
print("built %.1f s" % (time.time() - t0), flush=True)
K6 = 6 * k
def groups_of(N):
    kf_of = np.concatenate([np.repeat(np.arange(k), 6), np.repeat(np.asarray(w["lm_kf"]), 3)])
    return [np.where(kf_of == c)[0] for c in range(k)]
def blockdiag_inv(A, groups, N, f32=False):
    invs = []
    for g in groups:
        Bi = np.linalg.inv(A[g][:, g].toarray())
        if f32: Bi = Bi.astype(np.float32).astype(np.float64)
        invs.append((g, Bi))
    def mv(v):
        o = np.zeros_like(v)
        for g, Bi in invs: o[g] = Bi @ v[g]
        return o
    return spla.LinearOperator((N, N), matvec=mv)
def pcg(A, b, M, N):
    it = [0]
    def cb(x): it[0] += 1
    x, info = spla.cg(A, b, rtol=1e-10, maxiter=8000, M=M, callback=cb)
    return x, it[0]
lam, ni = -1, 2.0
M_stale = None
for it in range(5):
    G.compute_active_errors()
    chi = G.active_robust_chi2()
    H, b = G.build_system()
    N = H.shape[0]
    if it == 0:
        lam = 1e-5 * float(np.max(np.abs(H.diagonal())))
        print("unknowns %d nnz %d; sizes of keyframe blocks %s" % (N, H.nnz, [len(g) for g in groups_of(N)][:5]))
    A = (H + lam * sp.identity(N)).tocsr()
    grp = groups_of(N)
    bj = [np.arange(6 * i, 6 * i + 6) for i in range(k)] + [np.arange(K6 + 3 * i, K6 + 3 * i + 3) for i in range((N - K6) // 3)]
    t1 = time.time()
    Mf = blockdiag_inv(A, grp, N)
    Mf32 = blockdiag_inv(A, grp, N, True)
    if M_stale is None: M_stale = Mf
    x, n_f = pcg(A, b, Mf, N)
    _, n_32 = pcg(A, b, Mf32, N)
    _, n_s = pcg(A, b, M_stale, N)
    _, n_0 = pcg(A, b, blockdiag_inv(A, bj, N), N) if it in (0, 4) else (None, -1)
    # cond of one block
    g0 = grp[0]; ev = np.linalg.eigvalsh(A[g0][:, g0].toarray())
    print("LM it %d lam %.4g chi %.6g: P0 %d | P1 fresh %d | P1 fresh fp32-stored %d | P1 stale(it0) %d  [block0 eig %.3g..%.3g] %.0f s" % (it, lam, chi, n_0, n_f, n_32, n_s, ev[0], ev[-1], time.time() - t1), flush=True)
    G.push(); G.update(x); G.compute_active_errors()
    temp = G.active_robust_chi2()
    scale = float(np.dot(x, lam * x + b)) + 1e-3
    rho = (chi - temp) / scale
    print("   rho %.4f chi_new %.6g" % (rho, temp), flush=True)
    if rho > 0:
        alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0); lam *= max(1.0 / 3.0, alpha); ni = 2.0
    else:
        lam *= ni; ni *= 2; G.pop()
