"""Full-size golden for BASELINE configs[1] AS WRITTEN -- 5k map points x 500 deformation-graph nodes x 20 keyframes, the embedded
form of the BA window (N2b): the oracle's LM (oracle/embedded_oracle.py dba_solve_embedded, its own edge / skinning lists from
dba_build_embedded) on the complete window, linear solves by the oracle's sparse LU (nrs_oracle.solve_spd).  Stores the LM trace,
the final poses, samples and sums of the node copies and of the skinned points, and the checksums of the lists it was computed on.
    python tests/golden/make_embedded_ba_golden.py [n_nodes, default 500]
"""
import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd", "py"))
import embedded_oracle as E  # noqa: E402
import nrs_synth as S  # noqa: E402


def skin_checksum(e):
    h = 0
    for key, dt in (("lm_obs", np.int32), ("sk_obs", np.int32), ("sk_node", np.int32), ("sk_omega", np.float64)):
        h = (h * 1000003 + zlib.crc32(np.ascontiguousarray(e[key], dt).tobytes())) & 0xFFFFFFFFFFFF
    return int(h)


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    p = S.make_dba_problem("C2")
    flag, nb = S.embedded_problem(p, m)
    e = E.dba_build_embedded(p["kf_points"], flag, nb["rowptr"], nb["col"], nb["w"], nb["d0"], nb["status"])
    w = S.embedded_window(p, e)
    print("C2 embedded: %d node copies, %d springs, %d dampers, %d skinned observations" % (len(e["lm_obs"]), len(e["sp_ij"]), len(e["dm_idx"]), len(e["sk_obs"])), flush=True)
    t0 = time.time()
    tr = []
    q, t, pts, sk, nit = E.dba_solve_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"],
                                              e["dm_idx"], e["dm_w"], w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"], 5, tr)
    print("oracle: %d LM iterations, %d trials, %.0f s" % (nit, len(tr), time.time() - t0))
    sel = np.linspace(0, len(pts) - 1, 1000).astype(np.int64)
    ssel = np.linspace(0, len(sk) - 1, 2000).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "dba_C2_embedded%d_trace.npz" % m), n_nodes=m, out_q=q, out_t=t, out_iters=nit,
                        out_accepted=np.array([x["accepted"] for x in tr]), out_chi=np.array([x["chi"] for x in tr]),
                        out_chi_new=np.array([x["chi_new"] for x in tr]), out_lam=np.array([x["lam"] for x in tr]),
                        sel=sel, out_pts_sel=pts[sel], out_pts_sum=pts.sum(0), n_lm=len(pts), ssel=ssel, out_sk_sel=sk[ssel], out_sk_sum=sk.sum(0),
                        n_skin=len(sk), n_sp=len(e["sp_ij"]), n_dm=len(e["dm_idx"]), edge_checksum=S.edge_checksum(e), skin_checksum=skin_checksum(e))
    for x in tr:
        print(x["iter"], x["trial"], x["accepted"], "%.6e %.6e" % (x["chi"], x["chi_new"]))


if __name__ == "__main__":
    main()
