"""Golden for the EMBEDDED-DEFORMATION mode of a2 (include/nrs.h nrs_track_deform_solve_embedded; N2 as SURVEY.md 8d words it) at the
size bench.py measures it on: 5000 points x 500 nodes, the map's graph at the reference's all-pairs density, one pinhole and one
KannalaBrandt8 frame.  oracle/embedded_oracle.track_deform_solve_embedded on oracle/rgraph_oracle.DenseGraph runs ONCE here, in
the build container (minutes per frame: a GetEdges list of ~2000 entries per point); tests/test_gpu_embedded5k.py regenerates the
seeded inputs on the GPU box and holds the device result to what is stored: pose, statuses, lost ids, positions, median deformation,
every LM trial, graph statuses of probe rows.  The mode is "parity unpinned" beyond every-point-a-node: this
golden pins the product to ITS oracle at full size, not to reference output.

    python tests/golden/make_embedded5k_golden.py      ->  tests/golden/embedded5k_<model>.npz
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "nr-slam_amd", "py"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import embedded_oracle as E
import nrs_synth as S
import rgraph_oracle as RG
import skin_oracle as K

N_POINTS, N_NODES = 5000, 500
CASES = (("pinhole", S.PINHOLE, 4501), ("kb8", S.KB8, 4502))


def make_inputs(seed, model, n=N_POINTS, m=N_NODES):
    tp = S.make_tracking_problem(n, seed, model)
    nodes = K.select_nodes(tp["X_prev"], m, tp["status"] == 0)       # (bit-exact twin of nrs_skin_select_nodes: tests/test_gpu_skin.py)
    node = np.zeros(n, np.uint8)
    node[nodes] = 1
    probe = np.sort(np.random.default_rng(seed).choice(n, 40, replace=False)).astype(np.int32)
    return tp, node, probe


def main():
    for name, model, seed in CASES:
        dst = os.path.join(ROOT, "tests", "golden", "embedded5k_%s.npz" % name)
        if os.path.exists(dst):
            print("kept", dst)
            continue
        t0 = time.time()
        tp, node, probe = make_inputs(seed, model)
        n = N_POINTS
        ids = np.arange(n, dtype=np.int32)
        D = RG.DenseGraph(n, tp["graph"]["sigma"], tp["graph"]["stretch_th"])
        D.add_edges(tp["X_prev"], ids, ids)
        print("%s: graph ready (%.0f s)" % (name, time.time() - t0), flush=True)
        otr = []
        o = E.track_deform_solve_embedded(tp["model"], tp["prm"], D, tp["X_prev"], ids, tp["status"], tp["uv"], tp["X_prev"], node, tp["pose_q"], tp["pose_t"],
                                          tp["scale"], otr)
        tr = [(r, t["iter"], t["trial"], t["lam"], t["chi"], t["chi_new"], t["rho"], float(t["accepted"])) for r, lst in enumerate(otr) for t in lst]
        np.savez_compressed(dst, n=n, m=N_NODES, seed=seed, model=model, pose_q=o["pose_q"], pose_t=o["pose_t"], f_status=o["f_status"].astype(np.int8),
                            lost=np.array(o["lost"], np.int32), f_pos=o["f_pos"].astype(np.float32), map_pos=o["map_pos"].astype(np.float32),
                            median=np.float64(o["median"]), trials=np.array(tr, np.float64), probe=probe, probe_status=D.st[probe].astype(np.int8),
                            n_skinned=o["n_skinned"], n_edges=o["n_edges"],
                            in_sum=np.float64(tp["uv"].astype(np.float64).sum() + tp["X_prev"].astype(np.float64).sum() + tp["status"].sum() + node.sum()))
        print("%s: wrote %s (%d bytes), %d trials, %d skinned, %d lost, %.0f s" % (name, dst, os.path.getsize(dst), len(tr), o["n_skinned"], len(o["lost"]), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
