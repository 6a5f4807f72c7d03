"""GPU parity, bit-exact: nrs_shi_extract (reference modules/features/shi_tomasi.cc:38-409 + the mask filter
of modules/tracking/tracking.cc:118-134) against the oracle -- keypoints, class ids, and the extractor's
gradient / score buffers after every call of a stateful sequence."""
import os

import numpy as np
import pytest

import nrs
import nrs_synth as S
import shi_oracle as SH

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shi_160x120.npz")


def _check_buffers(ctx, ex):
    sc, xg, yg = ctx.shi_buffers()
    assert np.array_equal(xg, ex.Xg) and np.array_equal(yg, ex.Yg)
    assert np.array_equal(sc, ex.scores, equal_nan=True)


def test_golden(ctx):
    g = np.load(GOLD)
    ctx.shi_configure(int(g["nms"]))
    xy0, id0, n0 = ctx.shi_extract(g["im0"], None, g["mask"])
    assert n0 == len(g["out_xy0"]) and np.array_equal(xy0, g["out_xy0"]) and np.array_equal(id0, g["out_id0"])
    xy1, id1, n1 = ctx.shi_extract(g["im1"], g["prev1"], g["mask"])
    assert np.array_equal(xy1, g["out_xy1"]) and np.array_equal(id1, g["out_id1"])
    sc, xg, yg = ctx.shi_buffers()
    assert np.array_equal(sc, g["out_scores1"], equal_nan=True) and np.array_equal(xg, g["out_xg1"]) and np.array_equal(yg, g["out_yg1"])


@pytest.mark.parametrize("wh,nms,seed", [((640, 480), 5, 1), ((320, 320), 3, 2), ((97, 41), 2, 3), ((9, 5), 1, 4)])
def test_sequence_matches_oracle(ctx, wh, nms, seed):
    """Three frames through ONE extractor on both sides; the frame keeps every second new keypoint."""
    sq = S.make_lk_sequence(10, seed, wh=(max(wh[0], 160), max(wh[1], 120)), flow_px=4.0)
    ims = [sq["im0"][:wh[1], :wh[0]], sq["im1"][:wh[1], :wh[0]], sq["im0"][:wh[1], :wh[0]][::-1].copy()]
    rng = np.random.default_rng(seed)
    mask = (rng.uniform(size=(wh[1], wh[0])) > 0.1).astype(np.uint8)
    ctx.shi_configure(nms)
    ex = SH.ShiTomasi(nms)
    held = np.zeros((0, 2), np.float32)
    found = 0
    for im in ims:
        xy, ids, n = ctx.shi_extract(im, held, mask)
        oxy, oids = ex.extract(im, held, mask)
        assert n == len(oxy) and np.array_equal(xy, oxy) and np.array_equal(ids, oids)
        _check_buffers(ctx, ex)
        held = np.concatenate([held, xy[::2] + np.float32(0.25)])     # sub-pixel positions: round() picks the cell
        found += n
    assert found > 0 or wh[0] < 32


def test_new_extractor_starts_clean(ctx):
    sq = S.make_lk_sequence(10, 7, wh=(160, 120))
    ctx.shi_configure(5)
    a = ctx.shi_extract(sq["im0"])
    ctx.shi_extract(sq["im1"], a[0])
    ctx.shi_configure(5)
    b = ctx.shi_extract(sq["im0"])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and b[1][0] == 0


def test_capacity_and_bad_arguments(ctx):
    sq = S.make_lk_sequence(10, 8, wh=(160, 120))
    ctx.shi_configure(5)
    xy, ids, n = ctx.shi_extract(sq["im0"], capacity=5)
    assert n > 5 and len(xy) == 5
    with pytest.raises(nrs.NrsError):
        ctx.shi_extract(sq["im0"].T.copy())                          # portrait: the reference indexes out of bounds
    with pytest.raises(nrs.NrsError):
        ctx.shi_extract(sq["im0"], np.array([[400.0, 3.0]], np.float32))
    with pytest.raises(nrs.NrsError):
        ctx.shi_configure(40)


def test_strided_image_and_mask(ctx):
    """cv::Mat rows are `step` bytes apart: image and mask with padded rows give the result of the packed ones."""
    import ctypes as C
    sq = S.make_lk_sequence(10, 9, wh=(160, 120))
    im = sq["im0"]
    h, w = im.shape
    rng = np.random.default_rng(3)
    mask = (rng.uniform(size=(h, w)) > 0.3).astype(np.uint8)
    ctx.shi_configure(4)
    ref_xy, ref_id, n_ref = ctx.shi_extract(im, None, mask)
    pad_im = np.full((h, w + 37), 255, np.uint8); pad_im[:, :w] = im
    pad_mk = np.zeros((h, w + 11), np.uint8); pad_mk[:, :w] = mask
    ctx.shi_configure(4)
    xy = np.zeros((4096, 2), np.float32); ids = np.zeros(4096, np.int32); n = C.c_int32(0)
    rc = ctx.lib.nrs_shi_extract(ctx.h, pad_im.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int32(w), C.c_int32(h), C.c_int32(w + 37),
                                 pad_mk.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int32(w + 11), C.c_int32(0), None,
                                 C.c_int32(4096), xy.ctypes.data_as(C.POINTER(C.c_float)), ids.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n))
    assert rc == 0 and n.value == n_ref > 10
    assert np.array_equal(xy[:n.value], ref_xy) and np.array_equal(ids[:n.value], ref_id)
