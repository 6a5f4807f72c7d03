"""Device-side problem construction (csrc/nrs_engine_devpack.hpp) against the host construction it replaces
(csrc/nrs_engine_setup.hpp): every packed array -- vertex -> row map, sliced-ELL offsets, incidence headers and static data,
halo lists, tile classes, chi2 edge lists, row data -- bit for bit (FNV checksums through nrs_dba_pack_hash), and the solves on
top of the two identical to the last bit.  Windows: C2 (BASELINE configs[1]), a ragged window (few keyframes, many points, a
large share of unobserved points), and C3."""
import numpy as np
import pytest

import nrs
import nrs_synth as S

pytestmark = pytest.mark.gpu


def _solve(monkeypatch, host, p, e, cam, qt, iters=3):
    if host:
        nrs.debug_set("NRS_HOST_PACK", "1")
    else:
        nrs.debug_set("NRS_HOST_PACK", None)
    c = nrs.Context()
    c.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    h = c.dba_pack_hash()
    tr = nrs.Trace(64)
    c.dba_optimize(iters, tr)
    pq, xyz = c.dba_download()
    rr, rs, rd = c.dba_residuals()
    c.close()
    return h, tr.trials, pq, xyz, (rr, rs, rd)


@pytest.mark.parametrize("name,n,k,seed", [("C2", 5000, 20, 1), ("ragged", 16000, 5, 17), ("C3", 10000, 50, 2),
                                            ("reference window (fused path, T = 8)", 5000, 5, 1), ("small fused", 300, 4, 3)])
def test_device_pack_is_the_host_pack(monkeypatch, name, n, k, seed):
    p = S.make_dba_problem(n, k, seed) if name != "ragged" else S.make_dba_problem(n, k, seed, dropout=0.3)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    hd, td, qd, xd, rd = _solve(monkeypatch, False, p, e, cam, qt)
    hh, th, qh, xh, rh = _solve(monkeypatch, True, p, e, cam, qt)
    assert hd[21] == 1 and hh[21] == 0, "the two runs must take the two constructions"
    names = ["vrow", "ss_ptr", "sd_ptr", "s_om", "s_d0", "d_hdr", "d_w", "halo_ptr", "halo_rows", "halo_ns", "tile_list", "ec_sp", "ec_dm", "ec_w",
             "rflag", "uv", "xl_init", "pose_init", "grp_pose", "pose_grp_ptr", "scalars", "(path)", "tile_desc", "halo_fix"]
    bad = [nm for i, nm in enumerate(names) if hd[i] != hh[i] and i != 21]
    assert not bad, bad
    assert [(t["accepted"], t["inner"], t["lam"], t["chi"], t["chi_new"]) for t in td] == [(t["accepted"], t["inner"], t["lam"], t["chi"], t["chi_new"]) for t in th]
    assert np.array_equal(qd, qh) and np.array_equal(xd, xh)
    assert all(np.array_equal(a, b) for a, b in zip(rd, rh))


@pytest.mark.parametrize("model", [S.PINHOLE, S.KB8])
def test_compact_damper_headers_change_nothing(monkeypatch, model):
    """Two-kernel windows whose rows all know their temporal partners read 4-byte damper headers {o0:12, o2:12, meta:8} derived
    from the 8-byte ones (nrs_engine_setup.hpp engine_compact_headers).  It is an encoding only: with NRS_NO_H4=1 the kernels
    read the 8-byte headers, and every trial and every output must be the same to the last bit."""
    p = S.make_dba_problem(5000, 8, 21, model)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    nrs.debug_set("NRS_NO_H4", None)
    h4, t4, q4, x4, r4 = _solve(monkeypatch, False, p, e, cam, qt, iters=4)
    nrs.debug_set("NRS_NO_H4", "1")
    h8, t8, q8, x8, r8 = _solve(monkeypatch, False, p, e, cam, qt, iters=4)
    assert h4[20] != h8[20], "the two runs must take the two encodings (the flag is part of the scalar checksum)"
    assert [h4[i] for i in range(24) if i != 20] == [h8[i] for i in range(24) if i != 20]
    assert [(t["accepted"], t["inner"], t["lam"], t["chi"], t["chi_new"]) for t in t4] == [(t["accepted"], t["inner"], t["lam"], t["chi"], t["chi_new"]) for t in t8]
    assert np.array_equal(q4, q8) and np.array_equal(x4, x8)
    assert all(np.array_equal(a, b) for a, b in zip(r4, r8))


def test_tiny_windows_keep_the_host_path(monkeypatch):
    nrs.debug_set("NRS_HOST_PACK", None)
    p = S.make_dba_problem(100, 2, 3)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    c = nrs.Context()
    c.dba_upload(nrs.make_camera(p["model"], p["prm"]), np.concatenate([p["poses_q"], p["poses_t"]], 1), p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    assert c.dba_pack_hash()[21] == 0                    # a few hundred rows: not worth the device round trips
    c.close()


@pytest.mark.parametrize("n,k,seed,model", [(5000, 20, 1, S.PINHOLE), (5000, 5, 1, S.PINHOLE), (900, 6, 8, S.KB8), (200, 1, 4, S.PINHOLE)])
def test_solve_window_builds_the_same_edges_and_result(n, k, seed, model):
    """nrs_dba_solve_window: the edge construction of OPT:927-1137 on the device is index for index nrs_dba_build_edges (host),
    and the one-call solve returns what nrs_dba_build_edges + nrs_dba_solve return, to the last bit"""
    p = S.make_dba_problem(n, k, seed, model)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    c = nrs.Context()
    tr = nrs.Trace(64)
    pq, xyz = c.dba_solve_window(cam, qt, p["kf_points"], p["lm_xyz"], p["lm_uv"], p["nbr"], p["scale"], 5, tr)
    ed = c.dba_window_edges()
    if k >= 2:
        assert ed is not None, "this window should have taken the device path"
        for key in ("sp_ij", "sp_d0", "dm_idx", "dm_w"):
            assert np.array_equal(ed[key], e[key]), key
    c2 = nrs.Context()
    tr2 = nrs.Trace(64)
    pq2, xyz2 = c2.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], 5, tr2)
    assert [(t["accepted"], t["lam"], t["chi_new"]) for t in tr.trials] == [(t["accepted"], t["lam"], t["chi_new"]) for t in tr2.trials]
    assert np.array_equal(pq, pq2) and np.array_equal(xyz, xyz2)
    c.close(); c2.close()
