"""GPU: one deformable-BA window sharded over several ranks (include/nrs.h "multi-GPU").

The ranks of these tests are THREADS of this process, their contexts on the one GPU of the test box
(nrs_comm_init_local: the exchange steps are device copies + a host barrier).  The arithmetic is the
multi-GPU path's: every rank runs the row kernels for its own keyframes only, the pose blocks of the
normal equations / chi2 / dot products are summed over the ranks and the boundary-keyframe rows are
exchanged.  Held against the unsharded solve of the same problem at the tolerances the unsharded
solve is held to the oracle with (sums are taken in a different order, nothing else differs), and
against the oracle itself.  The RCCL back end is exercised with world = 1 (a 1-GPU box cannot host
two RCCL ranks); tests/test_dist_cpu.py covers the rank bookkeeping with gloo."""
import threading

import numpy as np
import pytest

import nrs
import nrs_oracle as O
import nrs_synth as S

pytestmark = pytest.mark.gpu


def _setup(n, k, seed, model=S.PINHOLE):
    p = S.make_dba_problem(n, k, seed, model)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    return p, e, cam, qt


def _run_sharded(world, p, e, cam, qt, iters=5, exact=0, resets=0):
    group = nrs.LocalGroup(world)
    out, errs = [None] * world, []
    stats = _run_sharded.stats = [None] * world

    def rank_main(r):
        try:
            c = nrs.Context(exact_trials=exact)
            c.comm_init_local(group, r)
            assert c.comm_rank() == (r, world)
            c.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
            tr = nrs.Trace()
            c.dba_optimize(iters, tr)
            for _ in range(resets):
                c.dba_reset()
                tr = nrs.Trace()
                c.dba_optimize(iters, tr)
            pq, xyz = c.dba_download()
            rr, rs, rd = c.dba_residuals()
            out[r] = (tr.trials, pq, xyz, rr, rs, rd)
            stats[r] = c.dba_stats()
            c.close()
        except Exception as ex:                      # a failed rank would leave the others in the barrier
            errs.append((r, ex))
            raise

    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not errs, errs
    assert all(o is not None for o in out), "a rank did not finish"
    group.close()
    return out


def _same_trials(a, b, rtol=1e-6):
    """Same decisions, lambdas and chi2.  WHERE a rejected trial stops being solved (the early-rejection
    looks) depends on the batching of the inner solve, which differs between the two paths (no convergence
    look-ahead when sharded): a trial cut short on one side and solved out on the other is rejected on both,
    only its chi2_new (an approximation when cut short) is not comparable (tools/sharded_sweep.py, seed 26)."""
    assert [t["accepted"] for t in a] == [t["accepted"] for t in b]
    for x, y in zip(a, b):
        assert abs(x["lam"] - y["lam"]) <= rtol * abs(y["lam"])
        assert abs(x["chi"] - y["chi"]) <= rtol * abs(y["chi"])
        assert not (x["early"] or y["early"]) or not (x["accepted"] or y["accepted"])
        if not y["early"] and not x["early"]:
            assert abs(x["chi_new"] - y["chi_new"]) <= rtol * abs(y["chi_new"])


@pytest.mark.parametrize("world,n,k,seed,model", [(2, 300, 4, 41, S.PINHOLE), (3, 400, 7, 42, S.PINHOLE),
                                                  (4, 250, 9, 43, S.KB8), (8, 150, 8, 44, S.PINHOLE)])
def test_sharded_matches_unsharded(ctx, world, n, k, seed, model):
    p, e, cam, qt = _setup(n, k, seed, model)
    ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    tr = nrs.Trace()
    ctx.dba_optimize(5, tr)
    pq0, xyz0 = ctx.dba_download()
    out = _run_sharded(world, p, e, cam, qt)
    for r in range(world):
        trials, pq, xyz, rr, rs, rd = out[r]
        _same_trials(trials, tr.trials)
        assert np.allclose(pq[:, :4], pq0[:, :4], atol=1e-6, rtol=0) and np.allclose(pq[:, 4:], pq0[:, 4:], atol=1e-5, rtol=0)
        assert np.allclose(xyz, xyz0, atol=1e-4, rtol=0)
    # every rank holds the same complete result, bit for bit (the all-reduce hands every rank the same sums)
    for r in range(1, world):
        assert [(t["lam"], t["chi"], t["chi_new"]) for t in out[r][0]] == [(t["lam"], t["chi"], t["chi_new"]) for t in out[0][0]]
        assert np.array_equal(out[r][1], out[0][1]) and np.array_equal(out[r][2], out[0][2])
        for a, b in zip(out[r][3:], out[0][3:]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("exact", [0, 1])
def test_sharded_matches_oracle(exact):
    p, e, cam, qt = _setup(300, 4, 32)
    out = _run_sharded(2, p, e, cam, qt, exact=exact)
    otr = []
    oq, ot, opts, _ = O.dba_solve(p["model"], p["prm"], p["poses_q"], p["poses_t"], p["lm_xyz"], p["lm_kf"],
                                  p["lm_uv"], e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"], p["scale"], 5, otr)
    trials, pq, xyz = out[0][:3]
    assert [t["accepted"] for t in trials] == [t["accepted"] for t in otr]
    for g, o in zip(trials, otr):
        assert abs(g["lam"] - o["lam"]) <= 1e-6 * abs(o["lam"])
        assert abs(g["chi"] - o["chi"]) <= 1e-6 * abs(o["chi"])
    assert np.allclose(pq[:, :4], oq, atol=1e-6) and np.allclose(pq[:, 4:], ot, atol=1e-5)
    assert np.allclose(xyz, opts, atol=1e-4)


def test_sharded_reset_is_reproducible():
    p, e, cam, qt = _setup(300, 5, 45)
    a = _run_sharded(2, p, e, cam, qt, resets=0)
    b = _run_sharded(2, p, e, cam, qt, resets=2)
    assert np.array_equal(a[0][1], b[0][1]) and np.array_equal(a[0][2], b[0][2])


def test_rccl_backend_single_rank(ctx):
    """The RCCL binding (dlopen, ncclCommInitRank, all-reduce on the context's stream) with world = 1:
    the sharded code path with every exchange step going through librccl; bitwise the same sums as the
    plain path except for their order."""
    p, e, cam, qt = _setup(300, 4, 46)
    ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    tr0 = nrs.Trace()
    ctx.dba_optimize(5, tr0)
    pq0, xyz0 = ctx.dba_download()
    c = nrs.Context()
    c.comm_init_rccl(1, 0, nrs.comm_unique_id())
    c.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    tr = nrs.Trace()
    c.dba_optimize(5, tr)
    pq, xyz = c.dba_download()
    c.close()
    _same_trials(tr.trials, tr0.trials)
    assert np.allclose(pq, pq0, atol=1e-6, rtol=0) and np.allclose(xyz, xyz0, atol=1e-4, rtol=0)


def test_sharded_rejects_too_many_ranks():
    p, e, cam, qt = _setup(120, 2, 47)
    group = nrs.LocalGroup(3)
    c = nrs.Context()
    c.comm_init_local(group, 0)
    with pytest.raises(nrs.NrsError) as ei:
        c.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    assert ei.value.code == -1
    c.close()
    group.close()


def test_sharded_c2_size(ctx):
    """The bench window (C2: 92k rows, 2 lanes per row, two-kernel PCG) split over 4 ranks, against the
    plain solve: same decisions, lambdas and chi2 to 1e-6, same result."""
    p = S.make_dba_problem("C2")
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    tr = nrs.Trace()
    ctx.dba_optimize(5, tr)
    pq0, xyz0 = ctx.dba_download()
    out = _run_sharded(4, p, e, cam, qt)
    trials, pq, xyz = out[0][:3]
    _same_trials(trials, tr.trials)
    assert np.allclose(pq[:, :4], pq0[:, :4], atol=1e-6, rtol=0) and np.allclose(pq[:, 4:], pq0[:, 4:], atol=1e-5, rtol=0)
    assert np.allclose(xyz, xyz0, atol=1e-4, rtol=0)
    assert np.array_equal(out[3][2], out[0][2])


def test_sharded_with_two_tile_classes(ctx, monkeypatch):
    """Tiles with outlier halos run as a second launch with its own LDS size; a rank's share
    of each class is a sub-range of that class's list."""
    p, e, cam, qt = _setup(500, 6, 48)
    ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    tr = nrs.Trace()
    ctx.dba_optimize(5, tr)
    pq0, xyz0 = ctx.dba_download()
    nrs.debug_set("NRS_TILE_CUT_PCT", "60")
    out = _run_sharded(3, p, e, cam, qt)
    trials, pq, xyz = out[0][:3]
    _same_trials(trials, tr.trials)
    assert np.allclose(pq, pq0, atol=1e-5, rtol=0) and np.allclose(xyz, xyz0, atol=1e-4, rtol=0)


def test_every_rank_packs_only_its_keyframe_range(ctx):
    """storage and set-up scale with 1 / ranks: a rank holds the incidence records of its own rows only (nrs_dba_stats)"""
    p, e, cam, qt = _setup(600, 8, 47)
    ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    whole = ctx.dba_stats()
    assert whole["packed_rows"] == whole["rows"]
    world = 4
    out = _run_sharded(world, p, e, cam, qt)
    st = _run_sharded.stats
    kb = nrs.shard_plan(8, p["lm_kf"], world)
    assert sum(x["packed_rows"] for x in st) == whole["rows"]                      # the ranges tile the window
    assert sum(x["spring_slots"] for x in st) <= whole["spring_slots"] + 64 * 4 * world
    assert sum(x["damper_slots"] for x in st) <= whole["damper_slots"] + 64 * 4 * world
    for r in range(world):
        own_kf = np.isin(p["lm_kf"], np.arange(kb[r], kb[r + 1]))
        assert st[r]["rows"] == whole["rows"] and own_kf.sum() <= st[r]["packed_rows"] < own_kf.sum() + 256 * (kb[r + 1] - kb[r])
        assert st[r]["spring_slots"] < 0.45 * whole["spring_slots"] and st[r]["device_bytes"] < 0.6 * whole["device_bytes"]


def test_a_rank_holds_the_rows_of_its_own_keyframes_and_two_ghosts_only(monkeypatch):
    """The per-row arrays (state, PCG vectors, diagonal blocks: ~410 bytes a row) of a rank cover its own keyframes and one ghost
    keyframe either side -- addressed by the global row index all the same (ArenaPlan::get_rows) -- instead of every row of the window
    (NRS_SHARD_FULL_VECTORS=1: the round-1..4 form).  Same launches, same exchanges: the two forms agree bit for bit (trials, poses,
    landmarks, residual taps, after a reset too), and what a rank holds shrinks with the rank count."""
    p, e, cam, qt = _setup(600, 16, 45)
    nrs.debug_set("NRS_SHARD_FULL_VECTORS", "1")
    full = _run_sharded(4, p, e, cam, qt, resets=1)
    full_bytes = [s["device_bytes"] for s in _run_sharded.stats]
    nrs.debug_set("NRS_SHARD_FULL_VECTORS", None)
    own = _run_sharded(4, p, e, cam, qt, resets=1)
    own_bytes = [s["device_bytes"] for s in _run_sharded.stats]
    for r in range(4):
        assert [(t["accepted"], t["lam"], t["chi"], t["chi_new"]) for t in own[r][0]] == [(t["accepted"], t["lam"], t["chi"], t["chi_new"]) for t in full[r][0]]
        for a, b in zip(own[r][1:], full[r][1:]):
            assert np.array_equal(a, b)
    # 16 keyframes over 4 ranks: 4 own + <= 2 ghost keyframes of 16 -> the row arrays are <= 6 / 16 of the full-length ones
    assert max(own_bytes) < 0.75 * min(full_bytes), (own_bytes, full_bytes)
