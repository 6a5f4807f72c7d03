"""Golden for a2 (CameraPoseAndDeformationOptimization, reference modules/optimization/g2o_optimization.cc:148-557) at
the size bench.py's `tracked_fps` measures it on: ~4.4k points of one frame, the map's graph at the reference's all-pairs
density (modules/map/map.cc:148-166), one pinhole and one KannalaBrandt8 frame.

The oracle (oracle/nrs_oracle.py: NumPy restatement, sparse direct solve per LM trial; oracle/rgraph_oracle.DenseGraph:
the graph) takes minutes per frame at this size, which is why its output is committed: this script runs it ONCE, in the
build container, on seeded inputs that tests/test_gpu_track5k.py regenerates on the GPU box (nrs_synth is deterministic),
and stores what the device result is held to: pose, statuses, lost ids, positions, median deformation, every LM trial
(round, lambda, chi2, chi2 after, decision), the graph statuses of probe rows after OPT:457-474.

    python tests/golden/make_track5k_golden.py [n_points]      ->  tests/golden/track5k_<model>.npz
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "nr-slam_amd", "py"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import nrs_oracle as O
import nrs_synth as S
import rgraph_oracle as RG

N_POINTS = 4446                 # what make_frame_sequence(5000, ...) keeps inside the image: bench.py's frame
CASES = (("pinhole", S.PINHOLE, 4401), ("kb8", S.KB8, 4402))


def make_inputs(n, seed, model):
    """The frame, the all-pairs graph's construction inputs and a history frame that stretched a patch (so BAD connections
    exist when a2 runs).  Shared with tests/test_gpu_track5k.py: both sides build their graph from these arrays."""
    tp = S.make_tracking_problem(n, seed, model)
    rng = np.random.default_rng(seed)
    hist = tp["X_prev"].copy()
    c0 = hist[3]
    patch = np.linalg.norm(hist - c0, axis=1) < 2.5 * tp["graph"]["sigma"]
    hist[patch] = c0 + (hist[patch] - c0) * np.float32(2.6)
    upd = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.int32)
    probe = np.sort(rng.choice(n, 40, replace=False)).astype(np.int32)
    return tp, hist, upd, probe


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else N_POINTS
    for name, model, seed in CASES:
        dst = os.path.join(ROOT, "tests", "golden", "track5k_%s.npz" % name)
        if os.path.exists(dst) and n == N_POINTS:
            print("kept", dst)
            continue
        t0 = time.time()
        tp, hist, upd, probe = make_inputs(n, seed, model)
        ids = np.arange(n, dtype=np.int32)
        D = RG.DenseGraph(n, tp["graph"]["sigma"], tp["graph"]["stretch_th"])
        D.add_edges(tp["X_prev"], ids, ids)
        good = np.array([D.update_vertex(hist, int(i)) for i in upd], np.int32)
        print("%s: graph ready (%.0f s)" % (name, time.time() - t0), flush=True)
        otr = []
        o = O.track_deform_solve(tp["model"], tp["prm"], D, tp["X_prev"], ids, tp["status"], tp["uv"], tp["X_prev"],
                                 tp["pose_q"], tp["pose_t"], tp["scale"], otr)
        rounds = otr if otr and isinstance(otr[0], list) else [[t for t in otr if t.get("round") == r] for r in sorted({t.get("round") for t in otr})]
        tr = [(r, t["iter"], t["trial"], t["lam"], t["chi"], t["chi_new"], t["rho"], float(t["accepted"])) for r, lst in enumerate(rounds) for t in lst]
        out = dict(n=n, seed=seed, model=model, pose_q=o["pose_q"], pose_t=o["pose_t"], f_status=o["f_status"].astype(np.int8),
                   lost=np.array(o["lost"], np.int32), f_pos=o["f_pos"].astype(np.float32), map_pos=o["map_pos"].astype(np.float32),
                   median=np.float64(o["median"]), trials=np.array(tr, np.float64), good=good, probe=probe, probe_status=D.st[probe].astype(np.int8),
                   # checksum of the regenerated inputs: the test refuses to compare if numpy's generators ever change
                   in_sum=np.float64(tp["uv"].astype(np.float64).sum() + tp["X_prev"].astype(np.float64).sum() + tp["status"].sum()))
        if n != N_POINTS:
            dst = dst.replace(".npz", "_%d.npz" % n)
        np.savez_compressed(dst, **out)
        print("%s: wrote %s (%d bytes), %d trials, %d lost, %.0f s" % (name, dst, os.path.getsize(dst), len(tr), len(o["lost"]), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
