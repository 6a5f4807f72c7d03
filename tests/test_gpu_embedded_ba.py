"""GPU parity: N2b, the EMBEDDED form of LocalDeformableBundleAdjustment (BASELINE configs[1] as written: points x graph nodes x
keyframes) through the C ABI (nrs_dba_*_embedded) against oracle/embedded_oracle.py dba_solve_embedded.

The reference has no such estimator; the mode has one pin: with every point a node the window IS the plain one (same lists --
tests/test_host_cpu.py -- and the same bits out of the solve, here), and the oracle is nrs_oracle.dba_solve bit for bit
(tests/test_oracle_embedded_cpu.py).  Beyond M = N ("parity unpinned") the GPU is held to that oracle at the a3 tolerances
(tests/test_gpu_dba.py): gradient / Hessian diagonal 1e-6 relative, per-trial chi2 1e-6 relative, identical accept / reject
sequence, rotation 1e-6, translation 1e-5, node copies and skinned points 1e-4 map units."""
import numpy as np
import pytest

import embedded_oracle as E
import nrs
import nrs_synth as S

pytestmark = pytest.mark.gpu


def _setup(n, k, m, seed, model=S.PINHOLE):
    p = S.make_dba_problem(n, k, seed, model)
    flag, nb = S.embedded_problem(p, m)
    p["nbr_nodes"] = nb
    e = nrs.dba_build_edges_embedded(p["kf_points"], flag, nb)
    w = S.embedded_window(p, e)
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    return p, e, w, cam, qt


def _oracle(p, e, w, iters=5, trace=None):
    return E.dba_solve_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"],
                                e["dm_idx"], e["dm_w"], w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"], iters, trace)


def test_every_point_a_node_is_the_plain_window_bit_for_bit(ctx):
    p, e, w, cam, qt = _setup(300, 4, 300, 51)
    assert len(e["sk_obs"]) == 0
    ta, tb = nrs.Trace(), nrs.Trace()
    pa, xa, _ = ctx.dba_solve_embedded(cam, qt, w, e, p["scale"], 5, ta)
    pb, xb = ctx.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], nrs.dba_build_edges(p["kf_points"], p["nbr_nodes"]), p["scale"], 5, tb)
    assert np.array_equal(pa, pb) and np.array_equal(xa, xb)
    assert [(t["accepted"], t["chi"], t["chi_new"], t["lam"]) for t in ta.trials] == [(t["accepted"], t["chi"], t["chi_new"], t["lam"]) for t in tb.trials]


@pytest.mark.parametrize("model", [S.PINHOLE, S.KB8])
def test_gradient_and_diagonal_include_the_skinned_observations(ctx, model):
    p, e, w, cam, qt = _setup(300, 4, 40, 52, model)
    ctx.dba_upload_embedded(cam, qt, w, e, p["scale"])
    b, d = ctx.dba_gradient()
    G, skn = E.dba_graph_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"],
                                  e["dm_idx"], e["dm_w"], w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"])
    G.initialize(0)
    G.compute_active_errors()
    H, bo = G.build_system()
    assert np.max(np.abs(b - bo)) <= 1e-6 * np.max(np.abs(bo))
    assert np.max(np.abs(d - H.diagonal())) <= 1e-6 * np.max(np.abs(H.diagonal()))
    # (and they matter: without the skinned observations the gradient is another one)
    G0, _ = E.dba_graph_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"],
                                 e["dm_idx"], e["dm_w"], w["sk_kf"][:0], w["sk_uv"][:0], w["sk_xyz"][:0], e["sk_node"][:0], e["sk_omega"][:0], p["scale"])
    G0.initialize(0)
    G0.compute_active_errors()
    assert np.max(np.abs(G0.build_system()[1] - bo)) > 0.1 * np.max(np.abs(bo))


@pytest.mark.parametrize("solver", ["factorisation", "pcg"])
@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("n,k,m,seed,model", [(300, 4, 40, 53, S.PINHOLE), (600, 6, 80, 54, S.PINHOLE), (400, 5, 60, 55, S.KB8)])
def test_solve_matches_oracle(ctx, ctx_exact, ctx_emb_pcg, ctx_emb_pcg_exact, n, k, m, seed, model, exact, solver):
    """both linear solvers of the embedded window (nrs_options.embedded_solver): the keyframe-block factorisation (the default at these
    sizes: one or two PCG iterations per trial) and the block-Jacobi PCG (hundreds)"""
    ctx = (ctx_exact if exact else ctx) if solver == "factorisation" else (ctx_emb_pcg_exact if exact else ctx_emb_pcg)
    p, e, w, cam, qt = _setup(n, k, m, seed, model)
    assert len(e["sk_obs"]) > 0.7 * len(p["lm_kf"])
    tr = nrs.Trace()
    pq, xyz, sk = ctx.dba_solve_embedded(cam, qt, w, e, p["scale"], 5, tr)
    otr = []
    oq, ot, opts, osk, nit = _oracle(p, e, w, 5, otr)
    assert tr.iterations == nit
    assert [t["accepted"] for t in tr.trials] == [t["accepted"] for t in otr]
    for a, b in zip(tr.trials, otr):
        assert (a["iter"], a["trial"]) == (b["iter"], b["trial"])
        assert abs(a["lam"] - b["lam"]) <= 1e-6 * b["lam"]
        assert abs(a["chi"] - b["chi"]) <= 1e-6 * b["chi"]
        if a["early"]:
            assert not exact and not a["accepted"] and not b["accepted"] and b["rho"] < -0.02
        else:
            assert abs(a["chi_new"] - b["chi_new"]) <= 1e-6 * b["chi_new"]
    assert np.allclose(pq[:, :4], oq, atol=1e-6, rtol=0)
    assert np.allclose(pq[:, 4:], ot, atol=1e-5, rtol=0)
    assert np.allclose(xyz, opts, atol=1e-4, rtol=0)
    assert np.allclose(sk, osk, atol=1e-4, rtol=0)
    inner = sum(t["inner"] for t in tr.trials if not t["early"])
    assert inner <= 3 * len(tr.trials) if solver == "factorisation" else inner > 20 * len(tr.trials)


@pytest.mark.parametrize("which", ["factorisation", "pcg"])
def test_resident_form_reset_and_determinism(ctx, ctx_emb_pcg, which):
    ctx = ctx if which == "factorisation" else ctx_emb_pcg
    p, e, w, cam, qt = _setup(500, 5, 70, 56)
    ctx.dba_upload_embedded(cam, qt, w, e, p["scale"])
    out = []
    for _ in range(3):
        ctx.dba_reset()
        tr = nrs.Trace()
        ctx.dba_optimize(5, tr)
        pq, xyz = ctx.dba_download()
        out.append((pq, xyz, ctx.dba_download_skinned(), [(t["accepted"], t["chi"], t["chi_new"], t["inner"]) for t in tr.trials]))
    for o in out[1:]:
        assert np.array_equal(o[0], out[0][0]) and np.array_equal(o[1], out[0][1]) and np.array_equal(o[2], out[0][2]) and o[3] == out[0][3]
    osk = _oracle(p, e, w)[3]
    assert np.allclose(out[0][2], osk, atol=1e-4, rtol=0)


@pytest.mark.parametrize("switch", ["NRS_SKIN_OP_OWN_LAUNCH", "NRS_SKIN_ROWS_OWN_LAUNCH"])
def test_two_launches_per_iteration_give_the_bits_of_three_and_four(ctx_emb_pcg, monkeypatch, switch):
    ctx = ctx_emb_pcg
    """A PCG iteration of the embedded window is two launches: k_spmv_f_skin (the regularisers' operator and k_skin_op) and
    k_pcg_update<true> (the observations' row pass and the vector update).  NRS_SKIN_OP_OWN_LAUNCH=1 / NRS_SKIN_ROWS_OWN_LAUNCH=1 give
    k_skin_op / the row pass (k_skin_op_rows, then the generic update) launches of their own (the env is read per launch): the same
    arithmetic on the same data -- every trial and every output bit for bit."""
    p, e, w, cam, qt = _setup(500, 5, 70, 58)
    out = []
    for own in (False, True):
        if own:
            nrs.debug_set(switch, "1")
        tr = nrs.Trace()
        pq, xyz, sk = ctx.dba_solve_embedded(cam, qt, w, e, p["scale"], 5, tr)
        out.append((pq, xyz, sk, [(t["accepted"], t["chi"], t["chi_new"], t["lam"], t["inner"]) for t in tr.trials]))
    nrs.debug_set(switch, None)
    assert sum(t[4] for t in out[0][3]) > 50
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert out[0][3] == out[1][3]


def test_hierarchical_reduction_path_matches(ctx_emb_pcg, monkeypatch):
    ctx = ctx_emb_pcg
    """Large windows reduce the partials in a kernel of their own (k_reduce_partials, from 4096 tiles on); NRS_HIER=1 forces that form
    on a small window: the row pass is then k_skin_op_rows and the update the generic kernel.  Same problem, same LM decisions, results
    within the oracle tolerances of the default form (the sums are grouped differently)."""
    p, e, w, cam, qt = _setup(400, 4, 50, 59)
    ta = nrs.Trace()
    pa, xa, ska = ctx.dba_solve_embedded(cam, qt, w, e, p["scale"], 5, ta)
    nrs.debug_set("NRS_HIER", "1")
    tb = nrs.Trace()
    pb, xb, skb = ctx.dba_solve_embedded(cam, qt, w, e, p["scale"], 5, tb)
    nrs.debug_set("NRS_HIER", None)
    assert [t["accepted"] for t in ta.trials] == [t["accepted"] for t in tb.trials]
    for a, b in zip(ta.trials, tb.trials):
        assert abs(a["chi"] - b["chi"]) <= 1e-6 * b["chi"]
    assert np.allclose(pa[:, :4], pb[:, :4], atol=1e-6, rtol=0) and np.allclose(pa[:, 4:], pb[:, 4:], atol=1e-5, rtol=0)
    assert np.allclose(xa, xb, atol=1e-4, rtol=0) and np.allclose(ska, skb, atol=1e-4, rtol=0)


def test_bad_inputs_are_rejected(ctx):
    p, e, w, cam, qt = _setup(200, 3, 30, 57)
    bad = dict(e, sk_node=e["sk_node"].copy())
    bad["sk_node"][0, 0] = len(w["lm_kf"])                        # node copy out of range
    with pytest.raises(nrs.NrsError):
        ctx.dba_upload_embedded(cam, qt, w, bad, p["scale"])
    bad = dict(e, sk_node=e["sk_node"].copy())
    other = np.where(w["lm_kf"] != w["sk_kf"][0])[0][0]           # a node copy of another keyframe
    bad["sk_node"][0, 0] = other
    with pytest.raises(nrs.NrsError):
        ctx.dba_upload_embedded(cam, qt, w, bad, p["scale"])


@pytest.mark.parametrize("mode", ["pcg-exact", "pcg-early-rejection", "factorisation"])
def test_full_size_c2_with_500_nodes_matches_the_oracle_golden(ctx, ctx_emb_pcg, ctx_emb_pcg_exact, mode):
    """(mode: the block-Jacobi PCG, nrs_options.embedded_solver = 2, with every trial solved to pcg_rtol / with early trial rejection, and the
    default context, whose cost model takes the keyframe-block factorisation at this size: 20 blocks of ~1.4k x 1.4k)
    BASELINE configs[1] as written -- 5k points x 500 nodes x 20 keyframes -- against tests/golden/dba_C2_embedded500_trace.npz: the
    oracle's LM on the complete embedded window (tests/golden/make_embedded_ba_golden.py: 443 s in the build container, sparse LU per
    trial), on the oracle's own lists (checksums in the fixture; the product's host builder must reproduce them).  Tolerances of the
    small cases."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    from make_embedded_ba_golden import skin_checksum
    g = np.load(os.path.join(here, "golden", "dba_C2_embedded500_trace.npz"))
    c = {"pcg-exact": ctx_emb_pcg_exact, "pcg-early-rejection": ctx_emb_pcg, "factorisation": ctx}[mode]
    p = S.make_dba_problem("C2")
    flag, nb = S.embedded_problem(p, int(g["n_nodes"]))
    e = nrs.dba_build_edges_embedded(p["kf_points"], flag, nb)
    assert (int(g["n_lm"]), int(g["n_skin"]), int(g["n_sp"]), int(g["n_dm"])) == (len(e["lm_obs"]), len(e["sk_obs"]), len(e["sp_ij"]), len(e["dm_idx"]))
    assert int(g["edge_checksum"]) == S.edge_checksum(e) and int(g["skin_checksum"]) == skin_checksum(e)
    w = S.embedded_window(p, e)
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    c.dba_upload_embedded(cam, qt, w, e, p["scale"])
    assert c.debug_kft_info()["on"] == (mode == "factorisation")
    tr = nrs.Trace()
    c.dba_optimize(5, tr)
    pq, xyz = c.dba_download()
    sk = c.dba_download_skinned()
    t = tr.trials
    if mode == "factorisation":
        assert sum(x["inner"] for x in t) <= 3 * len(t)
    assert tr.iterations == int(g["out_iters"])
    assert [x["accepted"] for x in t] == g["out_accepted"].tolist()
    for x, chi, chi_new, lam in zip(t, g["out_chi"], g["out_chi_new"], g["out_lam"]):
        assert abs(x["lam"] - lam) <= 1e-6 * lam and abs(x["chi"] - chi) <= 1e-6 * chi
        if not x["early"]:
            assert abs(x["chi_new"] - chi_new) <= 1e-6 * chi_new
    assert np.allclose(pq[:, :4], g["out_q"], atol=1e-6, rtol=0) and np.allclose(pq[:, 4:], g["out_t"], atol=1e-5, rtol=0)
    assert np.allclose(xyz[g["sel"]], g["out_pts_sel"], atol=1e-4, rtol=0) and np.allclose(sk[g["ssel"]], g["out_sk_sel"], atol=1e-4, rtol=0)
    assert np.allclose(xyz.sum(0), g["out_pts_sum"], atol=1e-4 * np.sqrt(len(xyz)), rtol=0)
    assert np.allclose(sk.sum(0), g["out_sk_sum"], atol=1e-4 * np.sqrt(len(sk)), rtol=0)


def test_gather_path_applies_the_observations_diagonal_blocks_once(ctx_exact, monkeypatch):
    """on the stored-block operator (use_lds = 0: NRS_NO_LDS, an irregular graph whose halo does not fit the LDS) the
    skinned observations' diagonal blocks sat in D (for the preconditioner) AND were applied by k_skin_op / the row pass -- H u counted
    them twice and the PCG converged to another system's solution.  The operator now reads the lineariser's own D (Dev::D_op): the
    gather path is held to the oracle like the LDS path, and to the LDS path's own result."""
    p, e, w, cam, qt = _setup(300, 4, 40, 57)
    tl = nrs.Trace()
    pl, xl, sl = ctx_exact.dba_solve_embedded(cam, qt, w, e, p["scale"], 5, tl)
    nrs.debug_set("NRS_NO_LDS", "1")
    tg = nrs.Trace()
    pg, xg, sg = ctx_exact.dba_solve_embedded(cam, qt, w, e, p["scale"], 5, tg)
    nrs.debug_set("NRS_NO_LDS", None)
    otr = []
    oq, ot, opts, osk, nit = _oracle(p, e, w, 5, otr)
    assert tg.iterations == nit and [t["accepted"] for t in tg.trials] == [t["accepted"] for t in otr]
    for a, b in zip(tg.trials, otr):
        assert abs(a["lam"] - b["lam"]) <= 1e-6 * b["lam"] and abs(a["chi"] - b["chi"]) <= 1e-6 * b["chi"] and abs(a["chi_new"] - b["chi_new"]) <= 1e-6 * b["chi_new"]
    assert np.allclose(pg[:, :4], oq, atol=1e-6, rtol=0) and np.allclose(pg[:, 4:], ot, atol=1e-5, rtol=0)
    assert np.allclose(xg, opts, atol=1e-4, rtol=0) and np.allclose(sg, osk, atol=1e-4, rtol=0)
    assert np.allclose(xg, xl, atol=1e-6, rtol=0) and np.allclose(sg, sl, atol=1e-6, rtol=0)
