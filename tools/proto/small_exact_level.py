import pickle, numpy as np, scipy.sparse as sp, sys
sys.path.insert(0, "/tmp/proto")
from exp import pcg, block_jacobi, aggregates, caps
def level(A, Ad, size, exact, with_pose):
    n = A.shape[0]
    agg, na = aggregates(Ad, size)
    rows = np.arange(n - 6) + 6; cols = 3 * agg[(rows - 6) // 3] + (rows - 6) % 3
    nc = 3 * na + (6 if with_pose else 0)
    Z = sp.csr_matrix((np.ones(n - 6), (rows, cols)), shape=(n, nc))
    if with_pose: Z = Z + sp.csr_matrix((np.ones(6), (np.arange(6), 3 * na + np.arange(6))), shape=(n, nc))
    Ac = (Z.T @ A @ Z).toarray()
    if exact: Ai = np.linalg.inv(Ac)
    else:
        Ai = np.zeros_like(Ac)
        for i in range(0, 3 * na, 3): Ai[i:i+3, i:i+3] = np.linalg.inv(Ac[i:i+3, i:i+3])
    return (lambda r: Z @ (Ai @ (Z.T @ r))), nc
for ci in [0, 1, 3, 8]:
    A, b = caps[ci]
    Mj, D, Ad = block_jacobi(A)
    base = pcg(A, b, Mj); res = {}
    L32, _ = level(A, Ad, 32, False, False)
    for size in (128, 256, 512):
        for wp in (False, True):
            Lx, nc = level(A, Ad, size, True, wp)
            res["exact%d%s(nc=%d)" % (size, "+pose" if wp else "", nc)] = pcg(A, b, lambda r: Mj(r) + Lx(r))
            res["32diag+exact%d%s" % (size, "+pose" if wp else "")] = pcg(A, b, lambda r: Mj(r) + L32(r) + Lx(r))
    print("sys", ci, "base", base, res, flush=True)
