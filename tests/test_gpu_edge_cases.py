"""Edge cases of the engine through the C ABI (empty / ragged inputs, degenerate graphs), each
checked against the oracle where the oracle defines the answer."""
import numpy as np
import pytest

import nrs
import nrs_oracle as O
import nrs_synth as S

pytestmark = pytest.mark.gpu


def _ba(ctx, p, e, iters=5):
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    tr = nrs.Trace()
    pq, xyz = ctx.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], iters, tr)
    otr = []
    oq, ot, opts, nit = O.dba_solve(p["model"], p["prm"], p["poses_q"], p["poses_t"], p["lm_xyz"], p["lm_kf"], p["lm_uv"],
                                    e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"], p["scale"], iters, otr)
    assert [t["accepted"] for t in tr.trials] == [t["accepted"] for t in otr]
    assert np.allclose(pq[:, :4], oq, atol=1e-6, rtol=0) and np.allclose(pq[:, 4:], ot, atol=1e-5, rtol=0)
    assert np.allclose(xyz, opts, atol=1e-4, rtol=0)
    return tr


def test_ba_without_regularisers(ctx):
    """no springs, no dampers: plain reprojection BA (every landmark sees one pose)."""
    p = S.make_dba_problem(80, 3, 71)
    e = dict(sp_ij=np.zeros((0, 2), np.int32), sp_d0=np.zeros(0, np.float32),
             dm_idx=np.zeros((0, 4), np.int32), dm_w=np.zeros(0, np.float32))
    _ba(ctx, p, e)


def test_ba_single_keyframe_and_ragged_keyframes(ctx):
    p = S.make_dba_problem(70, 1, 72)                       # one keyframe: springs only
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    assert len(e["dm_idx"]) == 0 and len(e["sp_ij"]) > 0
    _ba(ctx, p, e)
    p = S.make_dba_problem(300, 4, 73)                      # very different keyframe sizes
    keep = np.ones(len(p["lm_kf"]), bool)
    idx1 = np.where(p["lm_kf"] == 1)[0]
    keep[idx1[7:]] = False                                  # keyframe 1 keeps 7 observations
    idx3 = np.where(p["lm_kf"] == 3)[0]
    keep[idx3[::2]] = False
    for k in ("lm_xyz", "lm_kf", "lm_pt", "lm_uv"):
        p[k] = p[k][keep]
    p["kf_points"] = [p["lm_pt"][p["lm_kf"] == k] for k in range(4)]
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    sizes = [len(k) for k in p["kf_points"]]
    assert sizes[1] == 7 and max(sizes) > 10 * min(sizes)
    _ba(ctx, p, e)


def test_ba_more_rows_than_one_group(ctx):
    """a keyframe with more than ROW_ALIGN (256) landmarks spans several row groups."""
    p = S.make_dba_problem(700, 2, 74, dropout=0.0)
    assert min(len(k) for k in p["kf_points"]) > 512
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    _ba(ctx, p, e)


def test_ba_zero_iterations_and_reupload(ctx):
    p = S.make_dba_problem(60, 3, 75)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    tr = nrs.Trace()
    ctx.dba_optimize(0, tr)
    pq, xyz = ctx.dba_download()
    assert tr.iterations == 0 and len(tr.trials) == 0
    assert np.allclose(xyz, p["lm_xyz"].astype(np.float64)) and np.allclose(pq[:, 4:], p["poses_t"])
    # a larger problem after a smaller one re-uses / grows the arena
    p2 = S.make_dba_problem(400, 4, 76)
    e2 = nrs.dba_build_edges(p2["kf_points"], p2["nbr"])
    _ba(ctx, p2, e2)
    _ba(ctx, p, e)


def test_calls_before_upload_fail(lib_built):
    c = nrs.Context()
    with pytest.raises(nrs.NrsError) as ei:
        c.dba_optimize(5)
    assert ei.value.code == -5
    with pytest.raises(nrs.NrsError):
        c.dba_reset()
    with pytest.raises(nrs.NrsError) as ei:
        c.dba_stats()
    assert ei.value.code == -5
    c.close()


def test_track_all_neighbours_bad_and_isolated_points(ctx):
    """every graph edge BAD: no regularisers at all, points only see their reprojection edge."""
    tp = S.make_tracking_problem(200, 77)
    g = dict(tp["graph"])
    g["e_status"] = np.full_like(g["e_status"], S.GRAPH_BAD)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    fm = np.arange(200, dtype=np.int32)
    r = ctx.track_deform_solve(cam, g, tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"])
    o = O.track_deform_solve(tp["model"], tp["prm"], g, tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"])
    assert np.allclose(r["pose_q"], o["pose_q"], atol=1e-6, rtol=0) and np.allclose(r["pose_t"], o["pose_t"], atol=1e-5, rtol=0)
    assert np.array_equal(r["f_status"], o["f_status"]) and r["lost"] == o["lost"]
    assert np.allclose(r["f_pos"], o["f_pos"], atol=1e-4, rtol=0)


def test_track_frame_with_unmapped_slots(ctx):
    """frame slots without a map point (f_map = -1) and map points that are not in the frame."""
    tp = S.make_tracking_problem(260, 78)
    n = 260
    rng = np.random.default_rng(1)
    in_frame = np.sort(rng.choice(n, 200, replace=False))
    f_map = np.concatenate([in_frame, -np.ones(15, np.int64)]).astype(np.int32)
    st = np.concatenate([tp["status"][in_frame], np.full(15, 1, np.int32)])
    uv = np.concatenate([tp["uv"][in_frame], np.zeros((15, 2), np.float32)])
    pos = np.concatenate([tp["X_prev"][in_frame], np.zeros((15, 3), np.float32)])
    cam = nrs.make_camera(tp["model"], tp["prm"])
    r = ctx.track_deform_solve(cam, tp["graph"], tp["X_prev"], f_map, st, uv, pos, tp["pose_q"], tp["pose_t"], tp["scale"])
    o = O.track_deform_solve(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], f_map, st, uv, pos, tp["pose_q"], tp["pose_t"], tp["scale"])
    assert np.allclose(r["pose_q"], o["pose_q"], atol=1e-6, rtol=0)
    assert np.array_equal(r["f_status"], o["f_status"]) and r["lost"] == o["lost"]
    assert np.allclose(r["map_pos"], o["map_pos"], atol=1e-4, rtol=0)


def test_ba_two_tile_classes_bit_identical(ctx, monkeypatch):
    """Tiles with outlier halos run in their own launch (own LDS size).  Forcing the split at the
    median halo size must not change a single bit: per-tile arithmetic and the order of the partial
    sums are the same, only the launch a tile runs in differs."""
    p = S.make_dba_problem(1500, 8, 75)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    nrs.debug_set("NRS_FUSED_MAX_ROWS", "0")             # the two-kernel PCG path (as for large windows)
    res = []
    for cut in (None, "50"):
        if cut:
            nrs.debug_set("NRS_TILE_CUT_PCT", cut)
        tr = nrs.Trace()
        pq, xyz = ctx.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], 3, tr)
        res.append((pq, xyz, [(t["accepted"], t["inner"], t["chi"], t["chi_new"]) for t in tr.trials]))
    assert res[0][2] == res[1][2]
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("exact", [0, 1])
def test_ba_trial_chi2_paths_agree(exact, monkeypatch):
    """Trial states of a BA window are evaluated edge-parallel (one thread per spring / damper); the
    switch NRS_NO_EDGE_CHI=1 evaluates them from the incidence records of the counting rows instead.
    Same residuals and Huber, another order of the sum: identical decisions, chi2 to 1e-11 relative,
    same result.  (A spring residual k (d - d0) / d0 amplifies the last bit of d by d / (d - d0): the two kernels form d in
    differently contracted expressions, so the sums agree to ~1e-12, not to the 1e-14 of a pure reordering.)"""
    p = S.make_dba_problem(900, 6, 76)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    res = []
    for sw in (None, "1"):
        if sw:
            nrs.debug_set("NRS_NO_EDGE_CHI", sw)
        c = nrs.Context(exact_trials=exact)
        tr = nrs.Trace()
        pq, xyz = c.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], 5, tr)
        c.close()
        res.append((pq, xyz, tr.trials))
    a, b = res
    assert [(t["accepted"], t["early"], t["inner"]) for t in a[2]] == [(t["accepted"], t["early"], t["inner"]) for t in b[2]]
    for s, t in zip(a[2], b[2]):
        assert abs(s["chi_new"] - t["chi_new"]) <= 1e-11 * abs(t["chi_new"]) and abs(s["chi"] - t["chi"]) <= 1e-11 * abs(t["chi"])
    assert np.allclose(a[0], b[0], atol=1e-9, rtol=0) and np.allclose(a[1], b[1], atol=1e-6, rtol=0)


def test_contexts_and_engines_do_not_leak_device_memory(lib_built):
    """create / solve / destroy many times: device memory in use must not grow (arena reuse inside a
    context, full release on nrs_destroy)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")          # the runtime instance libnrs_hip.so itself is linked to

    def free_bytes():
        assert hip.hipDeviceSynchronize() == 0
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    p = S.make_dba_problem(400, 3, 77)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    tp = S.make_tracking_problem(300, 78)
    camt = nrs.make_camera(tp["model"], tp["prm"])
    fm = np.arange(300, dtype=np.int32)

    def cycle():
        c = nrs.Context()
        for _ in range(3):
            c.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], 2)
            c.track_deform_solve(camt, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"],
                                 tp["pose_q"], tp["pose_t"], tp["scale"])
        c.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
        c.dba_residuals()                         # the lazily staged tap buffer of the context
        g = nrs.RGraph(c, 300, tp["graph"]["sigma"], 1.1)   # dense graph: device state + pinned staging area
        g.add_edges(tp["X_prev"], fm, fm)
        g.get_edges(fm, 64)
        nodes = c.skin_select_nodes(tp["X_prev"], 40, tp["status"] == 0)
        c.track_deform_solve_rg(camt, g, tp["X_prev"], fm, nrs.skinned_status(tp["status"], fm, nodes), tp["uv"], tp["X_prev"],
                                tp["pose_q"], tp["pose_t"], tp["scale"], None, 256)
        g.close()
        c.close()

    cycle()                                   # warm-up: runtime / code-object allocations happen once
    free0 = free_bytes()
    for _ in range(25):
        cycle()
    free1 = free_bytes()
    assert free0 - free1 < 8 << 20, "device memory in use grew by %d bytes over 25 create/solve/destroy cycles" % (free0 - free1)


def test_non_finite_inputs_fail_fast_with_a_numeric_error(ctx):
    """NaN / inf in observations, positions or poses: the solves return NRS_ERR_NUMERIC at once
    (no hang, no non-finite output); the context stays usable."""
    p = S.make_dba_problem(200, 3, 79)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    for what in ("uv", "xyz", "pose"):
        uv, xyz, q = p["lm_uv"].copy(), p["lm_xyz"].copy(), qt.copy()
        if what == "uv":
            uv[5, 0] = np.nan
        elif what == "xyz":
            xyz[7, 2] = np.nan
        else:
            q[1, 4] = np.inf
        with pytest.raises(nrs.NrsError, match="NRS_ERR_NUMERIC"):
            ctx.dba_solve(cam, q, xyz, p["lm_kf"], uv, e, p["scale"], 5)
    tp = S.make_tracking_problem(200, 80)
    camt = nrs.make_camera(tp["model"], tp["prm"])
    fm = np.arange(200, dtype=np.int32)
    uv = tp["uv"].copy()
    uv[np.where(tp["status"] == 0)[0][3], 1] = np.nan
    with pytest.raises(nrs.NrsError, match="NRS_ERR_NUMERIC"):
        ctx.track_deform_solve(camt, tp["graph"], tp["X_prev"], fm, tp["status"], uv, tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"])
    pq, xyz = ctx.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], 2)      # still works
    assert np.isfinite(pq).all() and np.isfinite(xyz).all()
