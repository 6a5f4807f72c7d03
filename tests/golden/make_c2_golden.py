"""Full-size golden for BASELINE configs[1] (C2: 5k points x 20 keyframes): runs the oracle's LM
(oracle/nrs_oracle.py) on the complete problem with an iterative linear solver (block-Jacobi PCG to
1e-12, NumPy/SciPy) in place of the dense/LU solve that does not finish at this size, and stores the
LM trace, the final poses and a few landmark checksums.  Takes ~10-20 minutes on one core.
    python tests/golden/make_c2_golden.py
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd", "py"))
import nrs_oracle as O  # noqa: E402
import nrs_synth as S  # noqa: E402


def make_pcg_solver(K6, rtol=1e-12):
    def solver(A, b, x_prev):
        A = sp.csr_matrix(A)
        n = A.shape[0]
        m = (n - K6) // 3
        App = A[:K6, :K6].toarray()
        invp = np.zeros((K6, K6))
        for k in range(K6 // 6):
            invp[6 * k:6 * k + 6, 6 * k:6 * k + 6] = np.linalg.inv(App[6 * k:6 * k + 6, 6 * k:6 * k + 6])
        idx = K6 + 3 * np.arange(m)
        blocks = np.zeros((m, 3, 3))
        for i in range(3):
            for j in range(3):
                blocks[:, i, j] = np.asarray(A[idx + i, idx + j]).ravel()
        invl = np.linalg.inv(blocks)

        def M(v):
            out = np.empty_like(v)
            out[:K6] = invp @ v[:K6]
            out[K6:] = np.einsum('nij,nj->ni', invl, v[K6:].reshape(-1, 3)).ravel()
            return out

        x, info = spla.cg(A, b, rtol=rtol, maxiter=20000, M=spla.LinearOperator((n, n), matvec=M))
        return (info == 0 and np.all(np.isfinite(x))), x
    return solver


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    p = S.make_dba_problem(name)
    nb = p["nbr"]
    e = O.dba_build(p["kf_points"], nb["rowptr"], nb["col"], nb["w"], nb["d0"], nb["status"])   # the oracle's own edge list
    out = os.path.join(HERE, "dba_%s_trace.npz" % name)
    stamp = dict(n_sp=len(e["sp_ij"]), n_dm=len(e["dm_idx"]), edge_checksum=S.edge_checksum(e))
    if "--stamp-edges" in sys.argv:
        # the solve below takes ~7 minutes; when only the provenance of the edge list changed (it used to come from
        # the product's host builder, which tests/test_host_cpu.py holds equal to the oracle's at this size) the
        # existing trace stays valid and just receives the oracle edge list's size and checksum
        old = dict(np.load(out))
        old.update(stamp)
        np.savez_compressed(out, **old)
        print("stamped", out, stamp)
        return
    t0 = time.time()
    tr = []
    q, t, pts, nit = O.dba_solve(p["model"], p["prm"], p["poses_q"], p["poses_t"], p["lm_xyz"], p["lm_kf"], p["lm_uv"],
                                 e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"], p["scale"], 5, tr,
                                 solver=make_pcg_solver(6 * p["n_kf"]))
    print("oracle %s: %d LM iterations, %d trials, %.0f s" % (name, nit, len(tr), time.time() - t0))
    sel = np.linspace(0, len(pts) - 1, 2000).astype(np.int64)
    np.savez_compressed(out, out_q=q, out_t=t, out_iters=nit,
                        out_accepted=np.array([x["accepted"] for x in tr]), out_chi=np.array([x["chi"] for x in tr]),
                        out_chi_new=np.array([x["chi_new"] for x in tr]), out_lam=np.array([x["lam"] for x in tr]),
                        sel=sel, out_pts_sel=pts[sel], out_pts_sum=pts.sum(0), n_lm=len(pts), **stamp)
    for x in tr:
        print(x["iter"], x["trial"], x["accepted"], "%.6e %.6e" % (x["chi"], x["chi_new"]))


if __name__ == "__main__":
    main()
