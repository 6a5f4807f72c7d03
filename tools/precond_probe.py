"""Scratch probe (CPU): PCG iteration counts of candidate preconditioners on the pose-and-deformation systems (a2).
The systems are the (H + lambda I, b) pairs the oracle's LM loop hands to its solver; rows are ordered like the engine
orders them (Morton order of the image position, tiles of 32 points, groups of 8 tiles).
  python tools/precond_probe.py [n_points] [seed] [max_systems]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.sparse as sp, scipy.linalg as sla
import nrs_synth as S, nrs_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
max_sys = int(sys.argv[3]) if len(sys.argv) > 3 else 12
tp = S.make_tracking_problem(n, seed)
systems = []
def solver(A, b, x_prev):
    ok, x = O.solve_spd(A, b, x_prev)
    systems.append((sp.csr_matrix(A), b.copy(), x.copy()))
    return ok, x
fm = np.arange(n, dtype=np.int32)
O.track_deform_solve(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"],
                     tp["pose_q"], tp["pose_t"], tp["scale"], None, solver)
opt = np.where(tp["status"] == 0)[0]
N = len(opt)
uv = tp["uv"][opt]
def morton(ix, iy):
    k = np.zeros(len(ix), np.int64)
    for bit in range(12):
        k |= ((ix >> bit) & 1) << (2 * bit) | ((iy >> bit) & 1) << (2 * bit + 1)
    return k
order = np.argsort(morton((uv[:, 0] * 4).astype(np.int64), (uv[:, 1] * 4).astype(np.int64)), kind="stable")
TILE, GRP = 32, 256
print("systems %d, points %d" % (len(systems), N))

def pcg(A, b, Minv, rtol=1e-10, maxit=3000):
    x = np.zeros_like(b); r = b.copy(); u = Minv(r); p = u.copy(); g = r @ u; g0 = g
    for it in range(maxit):
        if g <= rtol * rtol * g0: return it
        w = A @ p; a = g / (p @ w); x += a * p; r -= a * w; u = Minv(r); gn = r @ u; p = u + (gn / g) * p; g = gn
    return maxit

rows = []
for (A, b, xs) in systems[:max_sys]:
    nd = A.shape[0]
    if nd != 6 + 3 * N: continue                                   # stage 1 systems only
    # permutation: pose, then points in Morton order
    perm = np.concatenate([np.arange(6), 6 + (3 * order[:, None] + np.arange(3)[None, :]).ravel()])
    Ap = A[perm][:, perm].tocsr(); bp = b[perm]
    Ad = Ap.toarray()
    # 3x3 block Jacobi (+ 6x6 pose)
    blk = [np.arange(6)] + [6 + 3 * i + np.arange(3) for i in range(N)]
    Dinv = sp.block_diag([np.linalg.inv(Ad[np.ix_(ix, ix)]) for ix in blk]).tocsr()
    # tile blocks (96 x 96) + pose
    tiles = [np.arange(6)] + [6 + np.arange(3 * t, min(3 * (t + TILE), 3 * N)) for t in range(0, N, TILE)]
    Tinv = sp.block_diag([np.linalg.inv(Ad[np.ix_(ix, ix)]) for ix in tiles]).tocsr()
    def coarse(size):                                              # piecewise-constant translations per `size` rows + the pose
        ng = (N + size - 1) // size
        Z = sp.lil_matrix((nd, 3 * ng + 6))
        for i in range(N):
            for c in range(3): Z[6 + 3 * i + c, 3 * (i // size) + c] = 1.0
        for c in range(6): Z[c, 3 * ng + c] = 1.0
        Z = Z.tocsr()
        Ac = (Z.T @ Ap @ Z).toarray()
        Aci = np.linalg.inv(Ac)
        return lambda r: Z @ (Aci @ (Z.T @ r))
    cg_, ct_ = coarse(GRP), coarse(TILE)
    # the engine's tile level: block-diagonal B_t^-1 of the tile translations (uncoupled)
    nt = (N + TILE - 1) // TILE
    Zt = sp.lil_matrix((nd, 3 * nt))
    for i in range(N):
        for c in range(3): Zt[6 + 3 * i + c, 3 * (i // TILE) + c] = 1.0
    Zt = Zt.tocsr(); Bt = (Zt.T @ Ap @ Zt).toarray()
    Bti = sp.block_diag([np.linalg.inv(Bt[3 * t:3 * t + 3, 3 * t:3 * t + 3]) for t in range(nt)]).tocsr()
    P = {
        "jac3": lambda r: Dinv @ r,
        "engine (jac3+grp+tile_diag)": lambda r: Dinv @ r + cg_(r) + Zt @ (Bti @ (Zt.T @ r)),
        "jac3+tilecoarse": lambda r: Dinv @ r + ct_(r),
        "tileblk": lambda r: Tinv @ r,
        "tileblk+grp": lambda r: Tinv @ r + cg_(r),
        "tileblk+tilecoarse": lambda r: Tinv @ r + ct_(r),
    }
    res = {k: pcg(Ap, bp, f) for k, f in P.items()}
    rows.append(res)
    print(" ".join("%s=%d" % kv for kv in res.items()), flush=True)
print("mean:", {k: float(np.mean([r[k] for r in rows])) for k in rows[0]})
