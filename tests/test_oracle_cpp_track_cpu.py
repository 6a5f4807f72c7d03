"""The C++ restatement of the per-frame solves (oracle/nrs_cpu_track.hpp: the compiled CPU baseline of the tracked-fps half of
the metric and a second checker) held to the NumPy oracle: a1 CameraPoseOptimization and a2
CameraPoseAndDeformationOptimization on the committed goldens and on fresh synthetic frames -- identical LM accept / reject
sequences, lambda / chi2 of the leading trials, inlier masks, statuses, lost sets and graph state exact, poses 1e-9,
positions 1e-6 (the two differ in their linear solve only: dense / SuperLU there, AMD-ordered block Cholesky here)."""
import os

import numpy as np
import pytest

import nrs_cpu as CPU
import nrs_oracle as O
import nrs_synth as S

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def cpu_lib():
    CPU.build()
    return CPU.load()


def _same_leading_trials(a, b, rounds, rtol=1e-7, noise=3e-7):
    n = 0
    for rnd in range(rounds):
        x = [t for t in a if t["round"] == rnd]
        y = b[rnd]
        for i, t in enumerate(y):
            if abs(t["chi"] - t["chi_new"]) <= noise * abs(t["chi"]):
                break                                    # from here on decisions sit on the fp32-projection noise floor
            assert i < len(x) and x[i]["accepted"] == t["accepted"], (rnd, i)
            assert abs(x[i]["chi"] - t["chi"]) <= rtol * abs(t["chi"]) + 1e-9 and abs(x[i]["lam"] - t["lam"]) <= rtol * abs(t["lam"])
            n += 1
    return n


def test_pose_only_golden_and_fresh(cpu_lib):
    d = np.load(os.path.join(G, "pose_only_120.npz"))
    q, t, inl, tr, st = CPU.pose_only_solve(int(d["model"]), d["prm"], d["uv"], d["X"], d["pose_q"], d["pose_t"], cpu_lib)
    assert np.allclose(q, d["out_q"], atol=1e-9) and np.allclose(t, d["out_t"], atol=1e-8) and np.array_equal(inl, d["out_inlier"])
    for n, seed, model in ((400, 3, S.PINHOLE), (300, 4, S.KB8)):
        tp = S.make_tracking_problem(n, seed, model)
        m = tp["status"] == 0
        otr = []
        oq, ot, oinl = O.pose_only_solve(tp["model"], tp["prm"], tp["uv"][m], tp["X_prev"][m], tp["pose_q"], tp["pose_t"], otr)
        q, t, inl, tr, st = CPU.pose_only_solve(tp["model"], tp["prm"], tp["uv"][m], tp["X_prev"][m], tp["pose_q"], tp["pose_t"], cpu_lib)
        assert np.allclose(q, oq, atol=1e-9) and np.allclose(t, ot, atol=1e-8) and np.array_equal(inl, oinl)
        assert _same_leading_trials(tr, otr, 3) > 6


@pytest.mark.parametrize("n,seed,model", [(150, 5, S.PINHOLE), (400, 6, S.PINHOLE), (350, 7, S.KB8)])
def test_track_deform_matches_numpy_oracle(cpu_lib, n, seed, model):
    tp = S.make_tracking_problem(n, seed, model)
    fm = np.arange(n)
    otr = []
    o = O.track_deform_solve(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"],
                             tp["pose_q"], tp["pose_t"], tp["scale"], otr)
    r = CPU.track_deform_solve(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"],
                               tp["pose_q"], tp["pose_t"], tp["scale"], cpu_lib)
    assert np.array_equal(r["f_status"], o["f_status"]) and r["lost"] == list(o["lost"])
    assert np.allclose(r["pose_q"], o["pose_q"], atol=1e-9) and np.allclose(r["pose_t"], o["pose_t"], atol=1e-8)
    assert np.allclose(r["f_pos"], o["f_pos"], atol=1e-6) and np.allclose(r["map_pos"], o["map_pos"], atol=1e-6)
    for k in ("e_status",):
        assert np.array_equal(r["graph"][k], o["graph"][k])
    for k in ("e_w", "e_max", "e_min"):
        assert np.allclose(r["graph"][k], o["graph"][k], atol=1e-6)
    assert abs(r["median"] - o["median"]) <= 1e-6
    assert _same_leading_trials(r["trace"], otr, len(otr)) > 10
    assert r["stats"]["n_factor"] == r["stats"]["n_trials"] > 0


def test_lk_cpp_is_the_numpy_lk_bit_for_bit(cpu_lib):
    """LucasKanadeTracker::SetReferenceImage + Track: the committed golden (240 x 180, 2 levels) and a fresh 640 x 480 pair with
    five levels: positions, statuses, good count and SSIM values identical"""
    import lk_oracle as LK
    d = np.load(os.path.join(G, "lk_240x180.npz"))
    lk = CPU.LucasKanadeCpp(max_level=2, lib=cpu_lib)
    lk.set_reference(d["im0"], d["pts"])
    xy, st, good, _ = lk.track(d["im1"], d["pts"] + np.float32(0.5), np.zeros(len(d["pts"]), np.int32))
    lk.close()
    assert np.array_equal(xy, d["out_xy"]) and np.array_equal(st, d["out_status"]) and good == int(d["out_good"])
    sq = S.make_lk_sequence(120, 9)
    o = LK.LucasKanadeOracle()
    o.set_reference(sq["im0"], sq["pts"])
    st0 = np.zeros(len(sq["pts"]), np.int32)
    oxy, ost, ogood, ossim = o.track(sq["im1"], sq["pts"], st0)
    lk = CPU.LucasKanadeCpp(lib=cpu_lib)
    lk.set_reference(sq["im0"], sq["pts"])
    xy, st, good, ssim = lk.track(sq["im1"], sq["pts"], st0)
    lk.close()
    assert np.array_equal(xy, oxy) and np.array_equal(st, ost) and good == ogood
    m = ~np.isnan(ossim)
    assert np.array_equal(np.isnan(ssim), np.isnan(ossim)) and np.array_equal(ssim[m], ossim[m])


# ---- N2a: the embedded-deformation form (oracle/nrs_cpu.cpp nrs_cpu_track_deform_solve_embedded) against oracle/embedded_oracle.py
def test_track_deform_embedded_with_every_point_a_node_is_the_parity_solve(cpu_lib):
    tp = S.make_tracking_problem(300, 8)
    fm = np.arange(300)
    a = CPU.track_deform_solve_embedded(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], np.ones(300, np.uint8),
                                        tp["pose_q"], tp["pose_t"], tp["scale"], cpu_lib)
    b = CPU.track_deform_solve(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], cpu_lib)
    assert a["n_skinned"] == 0 and a["lost"] == b["lost"] and a["median"] == b["median"]
    for k in ("pose_q", "pose_t", "f_pos", "f_status", "map_pos"):
        assert np.array_equal(a[k], b[k]), k
    assert [(t["accepted"], t["lam"], t["chi"], t["chi_new"]) for t in a["trace"]] == [(t["accepted"], t["lam"], t["chi"], t["chi_new"]) for t in b["trace"]]


@pytest.mark.parametrize("n,m,seed,model", [(400, 60, 21, S.PINHOLE), (600, 150, 22, S.KB8), (500, 100, 23, S.PINHOLE)])
def test_track_deform_embedded_matches_numpy_oracle(cpu_lib, n, m, seed, model):
    import embedded_oracle as E
    import skin_oracle as K
    tp = S.make_tracking_problem(n, seed, model)
    fm = np.arange(n)
    node = np.zeros(n, np.uint8)
    node[K.select_nodes(tp["X_prev"], m, tp["status"] == 0)] = 1
    otr = []
    o = E.track_deform_solve_embedded(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], node,
                                      tp["pose_q"], tp["pose_t"], tp["scale"], otr)
    r = CPU.track_deform_solve_embedded(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], node,
                                        tp["pose_q"], tp["pose_t"], tp["scale"], cpu_lib)
    assert (r["n_nodes"], r["n_skinned"]) == (o["n_nodes"], o["n_skinned"]) and r["n_skinned"] > 0.3 * n
    assert np.array_equal(r["f_status"], o["f_status"]) and r["lost"] == list(o["lost"])
    assert np.allclose(r["pose_q"], o["pose_q"], atol=1e-9) and np.allclose(r["pose_t"], o["pose_t"], atol=1e-8)
    assert np.allclose(r["f_pos"], o["f_pos"], atol=1e-6) and np.allclose(r["map_pos"], o["map_pos"], atol=1e-6)
    assert np.array_equal(r["graph"]["e_status"], o["graph"]["e_status"])
    for k in ("e_w", "e_max", "e_min"):
        assert np.allclose(r["graph"][k], o["graph"][k], atol=1e-6)
    assert abs(r["median"] - o["median"]) <= 1e-6
    assert _same_leading_trials(r["trace"], otr, len(otr)) > 8
