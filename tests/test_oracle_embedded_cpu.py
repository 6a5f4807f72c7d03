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


# ---- N2b: the embedded form of the BA window (LocalDeformableBundleAdjustment with skinned observations)
def _ba(p, flag):
    nb = S.node_lists(p["scene"]["X0"], p["scene"]["sigma"], flag)
    e = E.dba_build_embedded(p["kf_points"], flag, nb["rowptr"], nb["col"], nb["w"], nb["d0"], nb["status"])
    return e, S.embedded_window(p, e)


def test_ba_all_points_nodes_is_the_reference_window_bit_for_bit():
    for model, n, k, seed in ((S.PINHOLE, 90, 3, 5), (S.KB8, 70, 4, 6)):
        p = S.make_dba_problem(n, k, seed, model)
        nb = S.node_lists(p["scene"]["X0"], p["scene"]["sigma"], np.ones(p["n_points"], np.uint8))
        e, w = _ba(p, np.ones(p["n_points"], np.uint8))
        ref = O.dba_build(p["kf_points"], nb["rowptr"], nb["col"], nb["w"], nb["d0"], nb["status"])
        assert len(e["sk_obs"]) == 0 and np.array_equal(e["lm_obs"], np.arange(len(p["lm_kf"])))
        for key in ("sp_ij", "sp_d0", "dm_idx", "dm_w", "lm_kf", "lm_pt"):
            assert np.array_equal(e[key], ref[key]), key
        ta, tb = [], []
        qa, ta_, xa, sk, na = E.dba_solve_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"],
                                                   e["dm_idx"], e["dm_w"], w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"], 5, ta)
        qb, tb_, xb, nb_ = O.dba_solve(p["model"], p["prm"], p["poses_q"], p["poses_t"], p["lm_xyz"], p["lm_kf"], p["lm_uv"], ref["sp_ij"], ref["sp_d0"],
                                       ref["dm_idx"], ref["dm_w"], p["scale"], 5, tb)
        assert np.array_equal(qa, qb) and np.array_equal(ta_, tb_) and np.array_equal(xa, xb) and na == nb_ and ta == tb


def test_ba_skinned_observations_constrain_their_nodes():
    p = S.make_dba_problem(300, 4, 9)
    flag = S.pick_nodes(p["scene"]["X0"], 40)
    e, w = _ba(p, flag)
    n_obs = len(p["lm_kf"])
    assert len(e["lm_obs"]) + len(e["sk_obs"]) <= n_obs and len(e["sk_obs"]) > 0.7 * n_obs
    assert np.allclose(e["sk_omega"].sum(1), 1.0, atol=1e-12) and np.median((e["sk_node"] >= 0).sum(1)) == 11
    # a skinned observation's nodes live in its own keyframe
    for i in range(0, len(e["sk_obs"]), 37):
        nk = e["sk_node"][i][e["sk_node"][i] >= 0]
        assert (e["lm_kf"][nk] == w["sk_kf"][i]).all()
    tr = []
    q, t, x, sk, nit = E.dba_solve_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"],
                                            e["dm_idx"], e["dm_w"], w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"], 5, tr)
    acc = [s for s in tr if s["accepted"]]
    assert nit == 5 and len(acc) >= 3 and acc[-1]["chi_new"] < 0.85 * tr[0]["chi"]      # (the skinned points keep their own measurement noise: 40 nodes cannot absorb it)
    moved = np.linalg.norm(sk - w["sk_xyz"].astype(np.float64), axis=1)
    assert (moved > 0).mean() > 0.95 and np.isfinite(sk).all()
