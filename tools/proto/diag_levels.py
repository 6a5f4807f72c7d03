import pickle, numpy as np, scipy.sparse as sp, sys
sys.path.insert(0, "/tmp/proto")
from exp import pcg, block_jacobi, aggregates, caps
def diag_level(A, Ad, size):
    n = A.shape[0]
    agg, na = aggregates(Ad, size)
    rows = np.arange(n - 6) + 6; cols = 3 * agg[(rows - 6) // 3] + (rows - 6) % 3
    Z = sp.csr_matrix((np.ones(n - 6), (rows, cols)), shape=(n, 3 * na))
    Ac = (Z.T @ A @ Z).toarray()
    Dc = np.zeros_like(Ac)
    for i in range(0, 3 * na, 3): Dc[i:i+3, i:i+3] = np.linalg.inv(Ac[i:i+3, i:i+3])
    return lambda r: Z @ (Dc @ (Z.T @ r))
for ci in [0, 1, 3, 8]:
    A, b = caps[ci]
    Mj, D, Ad = block_jacobi(A)
    base = pcg(A, b, Mj)
    res = {}
    L = {s: diag_level(A, Ad, s) for s in (8, 16, 32, 64, 128, 256)}
    for s in L: res[s] = pcg(A, b, lambda r: Mj(r) + L[s](r))
    res["16+64"] = pcg(A, b, lambda r: Mj(r) + L[16](r) + L[64](r))
    res["8+32+128"] = pcg(A, b, lambda r: Mj(r) + L[8](r) + L[32](r) + L[128](r))
    res["32+128"] = pcg(A, b, lambda r: Mj(r) + L[32](r) + L[128](r))
    res["32+256"] = pcg(A, b, lambda r: Mj(r) + L[32](r) + L[256](r))
    print("sys", ci, "base", base, res, flush=True)
