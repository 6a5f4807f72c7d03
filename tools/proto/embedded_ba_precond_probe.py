"""N2b: what preconditioner does the embedded BA window want?  CPU experiment on the oracle's system (H + lambda I) of the first LM trial:
PCG iterations to 1e-10 with (P0) 3x3 / 6x6 block-Jacobi, (P1) exact blocks per keyframe (its node copies + its pose; the dampers'
cross-keyframe blocks left out), (P2) exact node blocks per keyframe with block-Jacobi poses.
usage: python tools/proto/embedded_ba_precond_probe.py [n_points n_nodes n_kf]"""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import embedded_oracle as E, nrs_synth as S
n, m, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1500, 150, 6)
p = S.make_dba_problem(n, k, 7)
flag, nb = S.embedded_problem(p, m)
e = E.dba_build_embedded(p["kf_points"], flag, nb["rowptr"], nb["col"], nb["w"], nb["d0"], nb["status"])
w = S.embedded_window(p, e)
G, skn = E.dba_graph_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"],
                              w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"])
G.initialize(0); G.compute_active_errors()
H, b = G.build_system()
lam = 1e-5 * np.abs(H.diagonal()).max()
A = (H + lam * sp.identity(H.shape[0])).tocsr()
N = A.shape[0]; K6 = 6 * k
kf_of = np.concatenate([np.repeat(np.arange(k), 6), np.repeat(np.asarray(w["lm_kf"]), 3)])
print("unknowns %d (%d node copies), nnz %d, lambda %.3g" % (N, len(w["lm_kf"]), A.nnz, lam))

def run(name, M):
    it = [0]
    def cb(x): it[0] += 1
    t0 = time.time()
    x, info = spla.cg(A, b, rtol=1e-10, maxiter=5000, M=M, callback=cb)
    print("%-48s %5d iterations (info %d, %.1f s)" % (name, it[0], info, time.time() - t0), flush=True)

def blockdiag_inv(groups):
    invs = [(g, np.linalg.inv(A[g][:, g].toarray())) for g in groups]
    def mv(v):
        o = np.zeros_like(v)
        for g, Bi in invs: o[g] = Bi @ v[g]
        return o
    return spla.LinearOperator((N, N), matvec=mv)

bj = [np.arange(6 * i, 6 * i + 6) for i in range(k)] + [np.arange(K6 + 3 * i, K6 + 3 * i + 3) for i in range((N - K6) // 3)]
run("P0 block-Jacobi (3x3 rows, 6x6 poses)", blockdiag_inv(bj))
run("P1 exact per keyframe (nodes + pose)", blockdiag_inv([np.where(kf_of == c)[0] for c in range(k)]))
run("P2 exact node block per keyframe, 6x6 poses", blockdiag_inv([np.arange(6 * i, 6 * i + 6) for i in range(k)] + [K6 + np.where(kf_of[K6:] == c)[0] for c in range(k)]))
# P3: node blocks per keyframe from the SKINNED + reprojection part only would need the split of H; instead: per-keyframe blocks of pairs of keyframes
run("P3 exact per PAIR of keyframes (2c, 2c+1)", blockdiag_inv([np.where((kf_of // 2) == c)[0] for c in range((k + 1) // 2)]))

# ---- P4: the pose eliminated against a block-Jacobi node block (arrow structure per keyframe): S = Hpp - Hpn D^-1 Hnp (6 x 6 per keyframe)
def schur_pose(Amat):
    Ac = Amat.tocsc()
    pidx, nidx = np.arange(K6), np.arange(K6, N)
    Hpp, Hpn, Hnn = Ac[pidx][:, pidx].toarray(), Ac[pidx][:, nidx].tocsr(), Ac[nidx][:, nidx]
    nn = (N - K6) // 3
    Dinv = np.zeros((nn, 3, 3))
    blocks = np.zeros((nn, 3, 3))
    idx = 3 * np.arange(nn)
    Hd = Hnn.tocsr()
    for a in range(3):
        for c in range(3):
            blocks[:, a, c] = np.asarray(Hd[idx + a, idx + c]).ravel()
    Dinv = np.linalg.inv(blocks)
    def dinv(v): return np.einsum('nij,nj->ni', Dinv, v.reshape(-1, 3)).ravel()
    DH = np.stack([dinv(np.asarray(Hpn[i].todense()).ravel()) for i in range(K6)], 1)      # D^-1 Hnp  (nn3 x K6)
    Sm = Hpp - Hpn @ DH
    Sinv = np.zeros_like(Sm)
    for c in range(k): Sinv[6 * c:6 * c + 6, 6 * c:6 * c + 6] = np.linalg.inv(Sm[6 * c:6 * c + 6, 6 * c:6 * c + 6])
    def mv(v):
        yn = dinv(v[K6:])
        xp = Sinv @ (v[:K6] - Hpn @ yn)
        return np.concatenate([xp, yn - DH @ xp])
    return spla.LinearOperator((N, N), matvec=mv)

for scale_l in (1.0, 1.0 / 81):
    A = (H + lam * scale_l * sp.identity(N)).tocsr()
    d = A.diagonal()
    print("lambda x %.4f: median node diag %.3g, median pose diag %.3g" % (scale_l, np.median(d[K6:]), np.median(d[:K6])))
    run("  P0 block-Jacobi", blockdiag_inv(bj))
    run("  P4 pose Schur over block-Jacobi nodes", schur_pose(A))
    run("  P1 exact per keyframe", blockdiag_inv([np.where(kf_of == c)[0] for c in range(k)]))

# ---- P5: the POSES eliminated exactly (they are few: 6 x 6 blocks), PCG on the node Schur complement S = Hnn - Hnp Hpp^-1 Hpn with 3 x 3 block-Jacobi
for scale_l in (1.0, 1.0 / 81):
    A = (H + lam * scale_l * sp.identity(N)).tocsc()
    pidx, nidx = np.arange(K6), np.arange(K6, N)
    Hpp, Hpn, Hnn = A[pidx][:, pidx].toarray(), A[pidx][:, nidx].tocsr(), A[nidx][:, nidx].tocsr()
    Hppi = np.linalg.inv(Hpp)
    nn = (N - K6) // 3
    idx = 3 * np.arange(nn)
    def blocks_of(M):
        B = np.zeros((nn, 3, 3))
        for a in range(3):
            for c in range(3): B[:, a, c] = np.asarray(M[idx + a, idx + c]).ravel()
        return B
    Bn = blocks_of(Hnn)
    # diagonal blocks of S: Hnn_ii - Hnp_i Hpp^-1 Hpn_i
    Hpn_d = Hpn.toarray()
    Bs = Bn.copy()
    for i in range(nn):
        c = Hpn_d[:, 3 * i:3 * i + 3]
        Bs[i] -= c.T @ Hppi @ c
    Sop = spla.LinearOperator((N - K6, N - K6), matvec=lambda v: Hnn @ v - Hpn.T @ (Hppi @ (Hpn @ v)))
    bs = b[K6:] - Hpn.T @ (Hppi @ b[:K6])
    for name, Bd in (("Hnn blocks", Bn), ("S blocks", Bs)):
        Bi = np.linalg.inv(Bd)
        it = [0]
        def cb(x): it[0] += 1
        x, info = spla.cg(Sop, bs, rtol=1e-10, maxiter=5000, M=spla.LinearOperator((N - K6, N - K6), matvec=lambda v: np.einsum('nij,nj->ni', Bi, v.reshape(-1, 3)).ravel()), callback=cb)
        print("lambda x %.4f  P5 poses eliminated, block-Jacobi (%s): %d iterations (info %d)" % (scale_l, name, it[0], info), flush=True)
