"""The embedded-deformation oracle (oracle/embedded_oracle.py, N2 as SURVEY.md 8d words it) against the reference restatement it
generalises: with every optimised point a node it must BE nrs_oracle.track_deform_solve (OPT:148-557) to the last bit -- the only
pin this mode has; with fewer nodes it must still lower the reprojection error of the skinned points through their nodes."""
import numpy as np

import embedded_oracle as E
import nrs_oracle as O
import nrs_synth as S
import skin_oracle as K


def _solve(tp, node, trace=None):
    n = len(tp["status"])
    return E.track_deform_solve_embedded(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], np.arange(n), tp["status"], tp["uv"], tp["X_prev"],
                                         node, tp["pose_q"], tp["pose_t"], tp["scale"], trace)


def test_all_points_nodes_is_the_reference_function_bit_for_bit():
    for model, n, seed in ((S.PINHOLE, 150, 11), (S.KB8, 120, 12)):
        tp = S.make_tracking_problem(n, seed, model)
        ta, tb = [], []
        a = _solve(tp, np.ones(n, np.uint8), ta)
        b = O.track_deform_solve(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], np.arange(n), tp["status"], tp["uv"], tp["X_prev"],
                                 tp["pose_q"], tp["pose_t"], tp["scale"], tb)
        for k in ("pose_q", "pose_t", "f_pos", "f_status", "map_pos"):
            assert np.array_equal(a[k], b[k]), k
        assert a["lost"] == b["lost"] and a["median"] == b["median"] and a["n_skinned"] == 0
        assert np.array_equal(a["graph"]["e_status"], b["graph"]["e_status"]) and np.array_equal(a["graph"]["e_w"], b["graph"]["e_w"])
        assert len(ta) == len(tb) and all(x == y for ra, rb in zip(ta, tb) for x, y in zip(ra, rb))


def test_skinned_points_follow_and_constrain_their_nodes():
    n, m = 400, 60
    tp = S.make_tracking_problem(n, 21)
    nodes = K.select_nodes(tp["X_prev"], m, tp["status"] == 0)
    node = np.zeros(n, np.uint8)
    node[nodes] = 1
    tr = []
    r = _solve(tp, node, tr)
    assert r["n_nodes"] == m and r["n_skinned"] > 200
    assert tr[0][0]["chi"] > 3 * tr[1][-1]["chi_new"]                      # the LM lowers chi2 (mostly the skinned points' reprojection error)
    opt = tp["status"] == 0
    moved = np.linalg.norm(r["f_pos"] - tp["X_prev"], axis=1)
    assert (moved[opt & (node == 0)] > 0).mean() > 0.9                      # skinned points moved with their nodes
    # nothing but the node set differs from the parity solve in what is classified: the pose stays close to the parity pose
    p = O.track_deform_solve(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], np.arange(n), tp["status"], tp["uv"], tp["X_prev"],
                             tp["pose_q"], tp["pose_t"], tp["scale"])
    assert np.allclose(r["pose_t"], p["pose_t"], atol=0.5) and np.allclose(r["pose_q"], p["pose_q"], atol=2e-2)
