"""What are the slow modes of the keyframe-block preconditioned system?  Take the system at a small lambda (lam0/27) and test coarse spaces."""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import embedded_oracle as E, nrs_synth as S, nrs_oracle as O
n, m, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (5000, 500, 20)
p = S.make_dba_problem(n, k, 11) if n != 5000 else S.make_dba_problem("C2")
flag, nb = S.embedded_problem(p, m)
e = E.dba_build_embedded(p["kf_points"], flag, nb["rowptr"], nb["col"], nb["w"], nb["d0"], nb["status"])
w = S.embedded_window(p, e)
G, skn = E.dba_graph_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"],
                              w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"])
G.initialize(0); G.compute_active_errors()
H, b = G.build_system()
N = H.shape[0]; K6 = 6 * k
lam0 = 1e-5 * float(np.max(np.abs(H.diagonal())))
lm_kf = np.asarray(w["lm_kf"]); nn = len(lm_kf)
kf_of = np.concatenate([np.repeat(np.arange(k), 6), np.repeat(lm_kf, 3)])
grp = [np.where(kf_of == c)[0] for c in range(k)]
X = np.asarray(G.pts, np.float64)          # node copy positions
print("pose_fixed", getattr(G, "pose_fixed", None))
def pcg(A, b, M):
    it = [0]
    def cb(x): it[0] += 1
    x, info = spla.cg(A, b, rtol=1e-10, maxiter=8000, M=M, callback=cb)
    return x, it[0]
def rigid_modes(kinds):
    cols = []
    for c in range(k):
        idx = np.where(lm_kf == c)[0]
        cen = X[idx].mean(0)
        for kind in kinds:
            for a in range(3):
                v = np.zeros(N)
                if kind == 't':
                    v[K6 + 3 * idx + a] = 1.0
                elif kind == 'r':
                    ax = np.zeros(3); ax[a] = 1
                    d = np.cross(ax, X[idx] - cen)
                    for q in range(3): v[K6 + 3 * idx + q] = d[:, q]
                elif kind == 's' and a == 0:
                    d = X[idx] - cen
                    for q in range(3): v[K6 + 3 * idx + q] = d[:, q]
                else: continue
                cols.append(v)
        if 'p' in kinds:
            pass
    return cols
for fac in (1.0, 1 / 27.0):
    lam = lam0 * fac
    A = (H + lam * sp.identity(N)).tocsr()
    invs = [(g, np.linalg.inv(A[g][:, g].toarray())) for g in grp]
    def m1(v):
        o = np.zeros_like(v)
        for g, Bi in invs: o[g] = Bi @ v[g]
        return o
    M1 = spla.LinearOperator((N, N), matvec=m1)
    _, n1 = pcg(A, b, M1)
    print("lam x %.4f: P1 %d" % (fac, n1), flush=True)
    for name, kinds, with_pose in (("t", "t", False), ("t+r", "tr", False), ("t+r+pose", "tr", True), ("t+r+s+pose", "trs", True)):
        cols = rigid_modes(kinds)
        if with_pose:
            for i in range(K6):
                v = np.zeros(N); v[i] = 1; cols.append(v)
        Z = np.stack(cols, 1)
        AZ = A @ Z
        Ec = Z.T @ AZ
        Eci = np.linalg.pinv(Ec)
        def m2(v, Z=Z, Eci=Eci):
            return m1(v) + Z @ (Eci @ (Z.T @ v))
        _, n2 = pcg(A, b, spla.LinearOperator((N, N), matvec=m2))
        # deflated (multiplicative / balancing): M = P M1 P^T + Z E^-1 Z^T with P = I - AZ E^-1 Z^T ... use the ADEF-like symmetric form
        def m3(v, Z=Z, Eci=Eci, AZ=AZ):
            q = Z @ (Eci @ (Z.T @ v))
            r1 = v - A @ q
            y = m1(r1)
            y = y - Z @ (Eci @ (AZ.T @ y))
            return q + y
        _, n3 = pcg(A, b, spla.LinearOperator((N, N), matvec=m3))
        print("   coarse %-12s dim %d: additive %d | multiplicative-symmetric %d" % (name, Z.shape[1], n2, n3), flush=True)
