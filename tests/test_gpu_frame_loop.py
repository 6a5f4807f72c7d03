"""the frame-loop harness (LK -> motion-model seed -> pose-only -> pose+deformation ->
point reuse -> keyframe cadence) driven through the C ABI, against the same harness driven by the
oracle.  The loop logic is shared host code; what is compared is the path: after every frame the
pose, every landmark's status and position, the lost-id set and the number of reused points."""
import numpy as np
import pytest

import nrs
import nrs_frame_loop as FL
import nrs_synth as S
from frame_loop_backend import OracleBackend

pytestmark = pytest.mark.gpu

OPTS = dict(win=21, max_level=4, max_iters=10, epsilon=1e-4, min_eig=1e-4)      # SLAM/system.cc:77-84


def _run(backend, sq, n_frames, kf_every):
    proj = lambda pc: FL.project_f32(sq["model"], sq["prm"], pc)
    loop = FL.FrameLoop(backend, proj, sq["wh"], sq["scale"], sq["kp0"], sq["X0"], sq["graph"], sq["pose_q"][0],
                        sq["pose_t"][0], sq["images"][0], images_to_insert_keyframe=kf_every)
    for f in range(1, n_frames):
        assert loop.track_image(sq["images"][f])
    return loop.log


@pytest.mark.parametrize("n,frames,seed,model,dense", [(260, 5, 9, S.PINHOLE, False), (200, 4, 12, S.KB8, False),
                                                        (240, 4, 14, S.PINHOLE, True)])
def test_frame_loop_matches_oracle_loop(n, frames, seed, model, dense):
    """dense: the map's graph at the reference's density (all pairs, device resident / oracle DenseGraph) instead of a flat kNN graph"""
    sq = S.make_frame_sequence(n, frames, seed, model)
    gb = FL.GpuBackend(nrs, sq["model"], sq["prm"], OPTS, dense_graph=dense)
    try:
        glog = _run(gb, sq, frames, 2)
    finally:
        gb.close()
    olog = _run(OracleBackend(sq["model"], sq["prm"], OPTS, dense_graph=dense), sq, frames, 2)
    assert any(L["keyframe"] for L in glog) and any(L["reused"] > 0 for L in glog)
    assert any((L["status_by_map"] != FL.TRACKED_WITH_3D).any() for L in glog)      # the occluders do knock points out
    for f, (g, o) in enumerate(zip(glog, olog), 1):
        assert np.allclose(g["pose_q"], o["pose_q"], atol=2e-6, rtol=0), f
        assert np.allclose(g["pose_t"], o["pose_t"], atol=2e-5, rtol=0), f
        assert g["lost"] == o["lost"] and g["reused"] == o["reused"] and g["keyframe"] == o["keyframe"], f
        assert np.array_equal(g["status_by_map"], o["status_by_map"]), f
        assert np.allclose(g["pos_by_map"], o["pos_by_map"], atol=2e-4, rtol=0), f
        assert g["n_tracked"] == o["n_tracked"] and g["n_tracked"] > 0.8 * sq["n_points"]
        assert g["n_2d"] == o["n_2d"] and np.array_equal(g["kp_2d"], o["kp_2d"]), f      # extracted features, then LK: bit-exact
    assert glog[-1]["n_2d"] > 0                                                           # keyframes did extract new corners
    # and the loop does track the scene: reprojection of the estimated landmarks with the estimated pose
    # lands on the true image positions
    last = glog[-1]
    ok = last["status_by_map"] == FL.TRACKED_WITH_3D
    pc = FL.se3f_act((last["pose_q"], last["pose_t"]), last["pos_by_map"][ok])
    uv = FL.project_f32(sq["model"], sq["prm"], pc)
    assert np.median(np.linalg.norm(uv - sq["uv_true"][frames - 1][ok], axis=1)) < 1.0


def test_frame_loop_in_the_embedded_mode_matches_the_oracle_loop():
    """the same loop with the pose-and-deformation solve of every frame in the embedded-deformation mode (40 of 260 map points carry the
    vertices for the whole sequence, everything else is skinned: include/nrs.h nrs_track_deform_solve_embedded), device against
    oracle/embedded_oracle.py behind the same harness"""
    n, frames, m = 260, 5, 40
    sq = S.make_frame_sequence(n, frames, 19, S.PINHOLE)
    gb = FL.GpuBackend(nrs, sq["model"], sq["prm"], OPTS, n_nodes=m)
    try:
        glog = _run(gb, sq, frames, 2)
        assert gb.node_flag.sum() == m
    finally:
        gb.close()
    ob = OracleBackend(sq["model"], sq["prm"], OPTS, n_nodes=m)
    olog = _run(ob, sq, frames, 2)
    assert np.array_equal(ob.node_flag, gb.node_flag)
    for f, (g, o) in enumerate(zip(glog, olog), 1):
        assert np.allclose(g["pose_q"], o["pose_q"], atol=2e-6, rtol=0), f
        assert np.allclose(g["pose_t"], o["pose_t"], atol=2e-5, rtol=0), f
        assert g["lost"] == o["lost"] and g["reused"] == o["reused"] and g["keyframe"] == o["keyframe"], f
        assert np.array_equal(g["status_by_map"], o["status_by_map"]), f
        assert np.allclose(g["pos_by_map"], o["pos_by_map"], atol=2e-4, rtol=0), f
        assert g["n_tracked"] == o["n_tracked"] and g["n_tracked"] > 0.8 * sq["n_points"]
    last = glog[-1]
    ok = last["status_by_map"] == FL.TRACKED_WITH_3D
    pc = FL.se3f_act((last["pose_q"], last["pose_t"]), last["pos_by_map"][ok])
    uv = FL.project_f32(sq["model"], sq["prm"], pc)
    assert np.median(np.linalg.norm(uv - sq["uv_true"][frames - 1][ok], axis=1)) < 1.5
