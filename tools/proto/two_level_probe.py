import pickle, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, time
d = pickle.load(open("/tmp/proto/caps.pkl", "rb"))
caps = d["caps"]
def pcg(A, b, Minv, rtol=1e-10, maxit=3000):
    x = np.zeros_like(b); r = b.copy(); u = Minv(r); p = u.copy(); g = r @ u; g0 = g
    for it in range(maxit):
        w = A @ p; a = g / (p @ w); x += a * p; r -= a * w; u = Minv(r); gn = r @ u
        if gn <= rtol * rtol * g0: return it + 1
        p = u + (gn / g) * p; g = gn
    return maxit
def block_jacobi(A, npose=6):
    n = A.shape[0]; A = sp.csr_matrix(A)
    Pi = np.linalg.inv(A[:npose, :npose].toarray())
    nb = (n - npose) // 3
    Ad = sp.bsr_matrix(A[npose:, npose:], blocksize=(3, 3))
    # diagonal blocks
    D = np.zeros((nb, 3, 3))
    indptr, indices, data = Ad.indptr, Ad.indices, Ad.data
    for i in range(nb):
        for k in range(indptr[i], indptr[i + 1]):
            if indices[k] == i: D[i] = data[k]
    Di = np.linalg.inv(D)
    def M(r):
        out = np.empty_like(r); out[:npose] = Pi @ r[:npose]
        out[npose:] = np.einsum('nij,nj->ni', Di, r[npose:].reshape(nb, 3)).ravel(); return out
    return M, D, Ad
def aggregates(Ad, size):
    nb = Ad.shape[0] // 3
    G = sp.csr_matrix((np.ones(len(Ad.indices)), Ad.indices, Ad.indptr), shape=(nb, nb))
    agg = -np.ones(nb, int); na = 0
    for s in range(nb):
        if agg[s] >= 0: continue
        # BFS up to size
        q = [s]; agg[s] = na; cnt = 1; h = 0
        while h < len(q) and cnt < size:
            v = q[h]; h += 1
            for w in G.indices[G.indptr[v]:G.indptr[v + 1]]:
                if agg[w] < 0 and cnt < size: agg[w] = na; q.append(w); cnt += 1
        na += 1
    return agg, na
def two_level(A, Mj, Z):
    Ac = (Z.T @ A @ Z).toarray(); Aci = np.linalg.inv(Ac)
    return lambda r: Mj(r) + Z @ (Aci @ (Z.T @ r))
if __name__ == "__main__":
  for ci in [0, 1, 2, 3, 4, 8]:
      A, b = caps[ci]; n = A.shape[0]; nb = (n - 6) // 3
      Mj, D, Ad = block_jacobi(A)
      base = pcg(A, b, Mj)
      res = [base]
      for size in (16, 32, 64):
          agg, na = aggregates(Ad, size)
          # 3-dof piecewise constant + pose identity
          rows = np.arange(n - 6) + 6; cols = 6 + 3 * agg[(rows - 6) // 3] + (rows - 6) % 3
          Z = sp.csr_matrix((np.ones(n - 6), (rows, cols)), shape=(n, 6 + 3 * na))
          Z = Z + sp.csr_matrix((np.ones(6), (np.arange(6), np.arange(6))), shape=(n, 6 + 3 * na))
          r3 = pcg(A, b, two_level(A, Mj, Z))
          # depth-only: weakest eigenvector of the diagonal block
          w, V = np.linalg.eigh(D); e = V[:, :, 0]       # smallest eigenvalue
          # orient consistently
          e *= np.sign(e @ e[0])[:, None] + (e @ e[0] == 0)[:, None]
          Z1 = sp.csr_matrix((e.ravel(), (rows, 6 + agg[(rows - 6) // 3])), shape=(n, 6 + na))
          Z1 = Z1 + sp.csr_matrix((np.ones(6), (np.arange(6), np.arange(6))), shape=(n, 6 + na))
          r1 = pcg(A, b, two_level(A, Mj, Z1))
          res.append((size, na, r3, r1))
      print("system", ci, "n", n, "block-jacobi", base, "| (size, n_agg, 3dof, depth-only):", res[1:], flush=True)
