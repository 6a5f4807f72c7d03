"""GPU parity: f2 batched DeformableTriangulation (reference g2o_optimization.cc:559-814) through the C ABI against
oracle/triang_oracle.py on flat temporal buffers (nrs_synth.make_temporal_buffer).

What can be held: the function's outcome.  The reprojection edge's Jacobian is g2o's numeric one (delta = 1e-9)
through the fp32 projection -- zero except where an estimate sits within 1e-9 of a float rounding boundary, where it
is +-1.5e4 -- so LM iterates are a function of the last bits of every solve and are not compared.
Tolerances: status codes (the InternalError cases) identical; triangulated point within 2e-3 map units (the points
differ by a few 1e-5 when a spike falls on different sides; 2e-3 is 0.07 % of the scene depth); the neighbour
selection / gates / seeds are exact fp32 twins, so every early-exit code must match exactly."""
import collections

import numpy as np
import pytest

import nrs
import nrs_synth as S
import triang_oracle as T

pytestmark = pytest.mark.gpu


def _compare(ctx, tb, n_max=None):
    cam = nrs.make_camera(tb["model"], tb["prm"])
    cand = tb["cand"] if n_max is None else tb["cand"][:n_max]
    st, xyz, dbg = ctx.triangulate_batch(cam, tb, cand, 5, debug=True)
    ref = [T.deformable_triangulation(tb, int(c), tb["model"], tb["prm"]) for c in cand]
    rst = np.array([r[0] for r in ref])
    rxyz = np.array([r[1] for r in ref])
    assert np.array_equal(st, rst), (collections.Counter(st.tolist()), collections.Counter(rst.tolist()))
    ok = st == 0
    err = np.linalg.norm(xyz[ok] - rxyz[ok], axis=1)
    if ok.any():
        assert err.max() <= 2e-3, err.max()
        assert np.median(err) <= 1e-4
    assert np.all(xyz[~ok] == 0)
    return st, xyz, dbg, err


@pytest.mark.parametrize("seed,model", [(3, S.PINHOLE), (4, S.PINHOLE), (5, S.KB8)])
def test_triangulation_matches_oracle(ctx, seed, model):
    tb = S.make_temporal_buffer(12, seed, model)
    st, xyz, dbg, err = _compare(ctx, tb)
    assert (st == 0).sum() >= 20 and (st == T.E_SHORT).sum() > 0
    # the solved problems: <= 12 vertices, hundreds of regulariser edges, all ten LM iterations unless converged
    assert dbg[st == 0, 3].min() > 50 and dbg[st == 0, 1].max() <= 10
    # sanity of the result itself: the triangulated depth follows the neighbours' (the reference's construction)
    # (pinhole only: with KannalaBrandt8 the reference scales a UNIT ray by the neighbours' mean z, OPT:653-675 -- its own bias)
    if model == S.PINHOLE:
        d = np.linalg.norm(xyz[st == 0] - tb["truth"][st == 0], axis=1)
        assert np.median(d) < 0.15


def test_full_buffer_and_gates(ctx):
    tb = S.make_temporal_buffer(21, 7, baseline=0.3)               # the reference's buffer size: 21 snapshots, 63 unknowns
    st, *_ = _compare(ctx, tb, 60)
    assert (st == 0).sum() > 10
    # an (almost) static camera: no parallax
    tb0 = S.make_temporal_buffer(10, 8, baseline=0.002)
    st, *_ = _compare(ctx, tb0, 40)
    assert set(st.tolist()) <= {T.E_PARALLAX, T.E_SHORT, T.E_REPROJ1, T.E_REPROJ2} and (st == T.E_PARALLAX).sum() > 0
    # dense features: a map point within 20 px of every candidate
    tbd = S.make_temporal_buffer(10, 9, spacing=14.0)
    st, *_ = _compare(ctx, tbd, 40)
    assert (st == T.E_CLOSE).sum() > 20
    # map points that vanish from the older snapshots: "Found no neighbours in a temporal point."
    tbn = S.make_temporal_buffer(12, 10)
    tbn["has_lm"][:3] = False
    st, *_ = _compare(ctx, tbn, 40)
    assert (st == T.E_NO_NEIGHBOUR).sum() > 0


def test_bad_arguments(ctx):
    tb = S.make_temporal_buffer(8, 11)
    cam = nrs.make_camera(tb["model"], tb["prm"])
    with pytest.raises(nrs.NrsError):
        ctx.triangulate_batch(cam, tb, [tb["has_kp"].shape[1] + 5])
    absent = int(np.where(~tb["has_kp"][-1])[0][0]) if (~tb["has_kp"][-1]).any() else None
    if absent is not None:
        with pytest.raises(nrs.NrsError):
            ctx.triangulate_batch(cam, tb, [absent])
    st, xyz = ctx.triangulate_batch(cam, tb, [])
    assert len(st) == 0
