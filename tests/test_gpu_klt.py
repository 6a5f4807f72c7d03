"""GPU parity: a21-a23 LucasKanadeTracker (reference modules/matching/lucas_kanade_tracker.cc) through
the C ABI against the oracle (oracle/lk_oracle.py), given this build's own restatement of the
OpenCV pyramid (parity unpinned at the pyramid boundary).

Bars: templates (int16 windows), status codes and n_good: exact; means and tracked positions:
exact as well (same fixed-point sampling, same sequential float32 sums, no FMA contraction) -- the
tests allow 1e-3 px / 1e-6 relative as the stated tolerance but assert bit-equality where it holds."""
import numpy as np
import pytest

import lk_oracle as LK
import nrs
import nrs_synth as S

pytestmark = pytest.mark.gpu


def _setup(ctx, n, seed, **kw):
    sq = S.make_lk_sequence(n, seed, **kw)
    ctx.klt_configure()
    ctx.klt_set_reference(sq["im0"], sq["pts"])
    lk = LK.LucasKanadeOracle()
    lk.set_reference(sq["im0"], sq["pts"])
    return sq, lk


def test_templates_match_oracle(ctx):
    sq, lk = _setup(ctx, 60, 3)
    assert ctx.klt_num_points() == len(sq["pts"])
    for i in (0, 7, len(sq["pts"]) - 1):
        t = ctx.klt_get_template(i)
        for level in range(5):
            assert bool(t["valid"][level]) == (lk.Iref[level][i] is not None)
            if lk.Iref[level][i] is not None:
                assert np.array_equal(t["gray"][level], lk.Iref[level][i])           # pyramid + sampling: bit-exact
                assert np.array_equal(t["grad"][level], lk.Idref[level][i])
                assert t["mean"][level, 0] == lk.meanI[level][i] and t["mean"][level, 1] == lk.meanI2[level][i]
            else:
                assert t["mean"][level, 0] == -1 and t["mean"][level, 1] == -1


@pytest.mark.parametrize("n,seed,flow", [(80, 5, 6.0), (150, 6, 3.0), (150, 7, 12.0)])
def test_track_matches_oracle(ctx, n, seed, flow):
    sq, lk = _setup(ctx, n, seed, flow_px=flow)
    st = np.zeros(len(sq["pts"]), np.int32)
    st[::17] = 3                                        # BAD points are skipped and left untouched
    guess = sq["pts"] + np.float32(0.7)
    xy, st2, good, ssim = ctx.klt_track(sq["im1"], guess, st)
    oxy, ost, ogood, ossim = lk.track(sq["im1"], guess.copy(), st)
    assert np.array_equal(st2, ost)                     # status codes: exact
    assert good == ogood
    ok = np.isin(ost, (0, 1, 2))
    assert np.allclose(xy[ok], oxy[ok], atol=1e-3, rtol=0)
    assert np.array_equal(xy, oxy)                      # and in fact bit-identical
    assert np.allclose(ssim[ok], ossim[ok], atol=1e-5)
    assert np.array_equal(xy[st == 3], guess[st == 3])
    # sanity: the tracker found the synthetic flow
    err = np.linalg.norm(xy[ost == 0] - sq["truth"][ost == 0], axis=1)
    assert np.median(err) < 0.5


def test_track_without_initial_flow_and_failures(ctx):
    sq, lk = _setup(ctx, 120, 9, flow_px=25.0)          # large flow: drift cap / SSIM failures appear
    st = np.zeros(len(sq["pts"]), np.int32)
    xy, st2, good, _ = ctx.klt_track(sq["im1"], sq["pts"], st, initial_flow=False, min_ssim=0.9)
    oxy, ost, ogood, _ = lk.track(sq["im1"], sq["pts"].copy(), st, initial_flow=False, min_ssim=0.9)
    assert np.array_equal(st2, ost) and good == ogood
    assert np.array_equal(xy, oxy)
    assert len(set(ost.tolist())) > 1                   # more than one status code is exercised


def test_border_points_and_mask(ctx):
    sq = S.make_lk_sequence(40, 11)
    h, w = sq["im0"].shape
    pts = np.concatenate([sq["pts"], np.array([[2.0, 3.0], [w - 3.0, h - 2.0], [w + 40.0, 10.0], [15.5, 15.5]], np.float32)])
    mask = np.full((h, w), 255, np.uint8)
    mask[200:260, 300:380] = 0
    ctx.klt_configure()
    ctx.klt_set_reference(sq["im0"], pts, mask)
    lk = LK.LucasKanadeOracle()
    lk.set_reference(sq["im0"], pts, mask)
    for i in range(len(pts)):
        t = ctx.klt_get_template(i)
        assert [bool(v) for v in t["valid"]] == [lk.Iref[l][i] is not None for l in range(5)], i
    st = np.zeros(len(pts), np.int32)
    xy, st2, good, _ = ctx.klt_track(sq["im1"], pts, st)
    oxy, ost, ogood, _ = lk.track(sq["im1"], pts.copy(), st)
    assert np.array_equal(st2, ost) and good == ogood
    assert (ost == 4).sum() >= 2                        # OUT_IMAGE_BOUNDARIES


def test_photometric_round_trip_and_clear(ctx):
    sq, lk = _setup(ctx, 30, 13)
    t = ctx.klt_get_template(4)
    n0 = ctx.klt_num_points()
    ctx.klt_insert_template(t)                          # PointReuse re-inserts stored templates (tracking.cc:473-503)
    assert ctx.klt_num_points() == n0 + 1
    t2 = ctx.klt_get_template(n0)
    for k in ("xy", "gray", "grad", "mean", "valid"):
        assert np.array_equal(t[k], t2[k])
    pts = np.concatenate([sq["pts"], sq["pts"][4:5]])
    xy, st, good, _ = ctx.klt_track(sq["im1"], pts, np.zeros(n0 + 1, np.int32))
    assert np.array_equal(xy[4], xy[n0]) and st[4] == st[n0]
    ctx.klt_clear()
    assert ctx.klt_num_points() == 0
    with pytest.raises(nrs.NrsError):
        ctx.klt_track(sq["im1"], pts, np.zeros(n0 + 1, np.int32))


def test_full_size_properties(ctx):
    """5k points on 640x480 (the size of BASELINE configs): tracking a frame against itself is the
    identity (size-independent property), and every point keeps its status."""
    sq = S.make_lk_sequence(5000, 21)
    ctx.klt_configure()
    ctx.klt_set_reference(sq["im0"], sq["pts"])
    st = np.zeros(len(sq["pts"]), np.int32)
    xy, st2, good, ssim = ctx.klt_track(sq["im0"], sq["pts"], st)
    assert good == len(sq["pts"]) and np.all(st2 == 0)
    assert np.max(np.abs(xy - sq["pts"])) < 0.05
    assert np.nanmin(ssim) > 0.99


def test_batched_templates_equal_single_point_calls(ctx):
    """nrs_klt_get_templates / nrs_klt_insert_templates are the single-point calls in a loop."""
    sq, lk = _setup(ctx, 40, 13)
    n = ctx.klt_num_points()
    batch = ctx.klt_get_templates(0, n)
    for i in (0, 5, n - 1):
        one = ctx.klt_get_template(i)
        for k in ("xy", "gray", "grad", "mean", "valid"):
            assert np.array_equal(batch[i][k], one[k]), k
    # re-insert a few into a fresh two-level tracker, in one call and one by one
    sel = [batch[i] for i in (1, 7, 20)]
    res = []
    for batched in (True, False):
        ctx.klt_clear()
        ctx.klt_configure(21, 1)
        if batched:
            ctx.klt_insert_templates(sel)
        else:
            for t in sel:
                ctx.klt_insert_template(t)
        assert ctx.klt_num_points() == 3
        pts = np.stack([t["xy"] for t in sel])
        res.append(ctx.klt_track(sq["im1"], pts + np.float32(0.5), np.zeros(3, np.int32)))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    ctx.klt_clear()
    ctx.klt_configure()


def test_archived_templates_equal_the_host_hand_over(ctx):
    """nrs_klt_archive_templates / nrs_klt_insert_archived: the map's photometric information kept in device memory by map point id.
    Archive some slots under arbitrary keys, overwrite one entry, then insert from the archive into (i) a fresh two-level tracker of
    ANOTHER context (PointReuse's tracker) and (ii) the archiving tracker itself -- byte for byte what get_templates + insert_templates
    hand over, and the same track results.  Unknown keys, bad slots and a tracker with more levels than the archive are refused."""
    sq, lk = _setup(ctx, 40, 13)
    n = ctx.klt_num_points()
    host = ctx.klt_get_templates(0, n)
    slots, keys = np.array([3, 7, 20, 11], np.int32), np.array([900, 2, 57, 900], np.int32)     # (key 900 is archived twice: the later slot stands)
    ctx.klt_archive_templates(slots, keys)
    other = nrs.Context()
    try:
        use_keys, use_slots = [2, 900, 57], [7, 11, 20]
        xy = np.stack([host[s]["xy"] for s in use_slots]) + np.float32(0.5)
        res = []
        for archived in (True, False):
            other.klt_clear()
            other.klt_configure(21, 1)
            if archived:
                other.klt_insert_archived(ctx, use_keys, xy)
            else:
                other.klt_insert_templates([dict(host[s], xy=xy[i]) for i, s in enumerate(use_slots)])
            assert other.klt_num_points() == 3
            got = other.klt_get_templates(0, 3)
            res.append((got, other.klt_track(sq["im1"], xy, np.zeros(3, np.int32))))
        for a, b in zip(res[0][0], res[1][0]):
            for k in ("xy", "gray", "grad", "mean", "valid"):
                assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(res[0][1][0], res[1][1][0]) and np.array_equal(res[0][1][1], res[1][1][1])
        # into the archiving tracker itself (a reused point becomes a new slot of the frame's tracker): all five levels
        ctx.klt_insert_archived(ctx, [57], host[20]["xy"][None])
        t = ctx.klt_get_template(n)
        for k in ("xy", "gray", "grad", "mean", "valid"):
            assert np.array_equal(t[k], host[20][k]), k
        with pytest.raises(nrs.NrsError):
            other.klt_insert_archived(ctx, [5], xy[:1])              # nothing archived under key 5
        with pytest.raises(nrs.NrsError):
            ctx.klt_archive_templates([n + 5], [1])                 # no such slot
        two = nrs.Context()
        try:
            two.klt_configure(21, 1)
            two.klt_set_reference(sq["im0"], sq["pts"][:4])
            two.klt_archive_templates([0], [0])
            with pytest.raises(nrs.NrsError):
                ctx.klt_insert_archived(two, [0], xy[:1])           # a five-level tracker cannot take a two-level entry
        finally:
            two.close()
    finally:
        other.close()
    ctx.klt_clear()
    ctx.klt_configure()
