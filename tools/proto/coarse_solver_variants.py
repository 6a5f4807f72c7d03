import pickle, numpy as np, scipy.sparse as sp, sys
sys.path.insert(0, "/tmp/proto")
from exp import pcg, block_jacobi, aggregates, caps
def cheb_solver(Ac, m):
    # m steps of Chebyshev on Jacobi-scaled A_c: fixed polynomial => fixed SPD operator
    n = Ac.shape[0]
    # block-diagonal (3x3) scaling
    Dc = np.zeros_like(Ac)
    for i in range(0, n - 6, 3): Dc[i:i+3, i:i+3] = np.linalg.inv(Ac[i:i+3, i:i+3])
    Dc[n-6:, n-6:] = np.linalg.inv(Ac[n-6:, n-6:])
    B = Dc @ Ac
    ev = np.linalg.eigvals(B).real; lmax = ev.max() * 1.05; lmin = lmax / 30.0
    theta = (lmax + lmin) / 2; delta = (lmax - lmin) / 2
    def solve(r):
        x = np.zeros_like(r); res = r.copy(); sigma = theta / delta; rho = 1 / sigma
        d = (Dc @ res) / theta
        for k in range(m):
            x = x + d; res = r - Ac @ x
            rho_new = 1 / (2 * sigma - rho)
            d = rho_new * rho * d + (2 * rho_new / delta) * (Dc @ res); rho = rho_new
        return x
    return solve
for ci in [1, 3]:
    A, b = caps[ci]; n = A.shape[0]
    Mj, D, Ad = block_jacobi(A)
    base = pcg(A, b, Mj)
    agg, na = aggregates(Ad, 32)
    rows = np.arange(n - 6) + 6; cols = 3 * agg[(rows - 6) // 3] + (rows - 6) % 3
    Z = sp.csr_matrix((np.ones(n - 6), (rows, cols)), shape=(n, 3 * na + 6))
    Z = Z + sp.csr_matrix((np.ones(6), (np.arange(6), 3 * na + np.arange(6))), shape=(n, 3 * na + 6))
    Ac = (Z.T @ A @ Z).toarray()
    exact = np.linalg.inv(Ac)
    r_exact = pcg(A, b, lambda r: Mj(r) + Z @ (exact @ (Z.T @ r)))
    # V1: block-diagonal of A_c only
    Dc = np.zeros_like(Ac)
    for i in range(0, 3 * na, 3): Dc[i:i+3, i:i+3] = np.linalg.inv(Ac[i:i+3, i:i+3])
    Dc[3*na:, 3*na:] = np.linalg.inv(Ac[3*na:, 3*na:])
    r_v1 = pcg(A, b, lambda r: Mj(r) + Z @ (Dc @ (Z.T @ r)))
    out = []
    for m in (4, 8, 16):
        sol = cheb_solver(Ac, m)
        out.append((m, pcg(A, b, lambda r: Mj(r) + Z @ sol(Z.T @ r))))
    # fp32 coarse inverse
    e32 = exact.astype(np.float32).astype(np.float64)
    r_f32 = pcg(A, b, lambda r: Mj(r) + Z @ (e32 @ (Z.T @ r)))
    # stale lambda: inverse built with lambda/8 and lambda*8 (A includes lambda already; emulate by shifting)
    lam_est = 1e-5 * abs(A.diagonal()).max()
    print("sys", ci, "base", base, "exact", r_exact, "diag-only", r_v1, "cheb", out, "fp32 inverse", r_f32, flush=True)
