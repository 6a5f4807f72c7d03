"""Full-size runs beyond what the oracle can follow (BASELINE config C3: 10k points x 50 keyframes,
455k landmarks), held to size-independent properties: bit-reproducibility, identical LM decisions in
the default (early-rejecting) and exact trial modes, a monotone chi2 over the accepted steps, and
agreement of the two operator paths (LDS-staged factored form vs global-gather fallback)."""
import numpy as np
import pytest

import nrs
import nrs_synth as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3():
    n_points, n_kf, seed, model = S.CONFIGS["C3"]
    p = S.make_dba_problem(n_points, n_kf, seed, model)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    return p, e, nrs.make_camera(p["model"], p["prm"]), np.concatenate([p["poses_q"], p["poses_t"]], 1)


def _solve(c, c3, iters=3):
    p, e, cam, qt = c3
    tr = nrs.Trace(128)
    pq, xyz = c.dba_solve(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"], iters, tr)
    return pq, xyz, tr.trials


def test_c3_reproducible_and_mode_independent(ctx, ctx_exact, c3):
    a = _solve(ctx, c3)
    b = _solve(ctx, c3)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])                 # bit-reproducible
    assert [(t["accepted"], t["inner"], t["chi_new"]) for t in a[2]] == [(t["accepted"], t["inner"], t["chi_new"]) for t in b[2]]
    x = _solve(ctx_exact, c3)
    assert [t["accepted"] for t in a[2]] == [t["accepted"] for t in x[2]]
    assert all(abs(s["lam"] - t["lam"]) <= 1e-12 * t["lam"] for s, t in zip(a[2], x[2]))
    assert np.array_equal(a[0], x[0]) and np.array_equal(a[1], x[1])                 # accepted steps are solved alike
    acc = [t for t in a[2] if t["accepted"]]
    assert len(acc) >= 3 and all(t["chi_new"] < t["chi"] for t in acc)
    assert any(t["early"] for t in a[2])
    assert np.isfinite(a[1]).all() and np.isfinite(a[0]).all()


def test_c3_operator_paths_agree(ctx, c3, monkeypatch):
    a = _solve(ctx, c3, 2)
    nrs.debug_set("NRS_NO_LDS", "1")                                            # global-gather operator, stored H blocks
    g = _solve(ctx, c3, 2)
    assert [t["accepted"] for t in a[2]] == [t["accepted"] for t in g[2]]
    assert np.allclose(a[0], g[0], atol=1e-9, rtol=0) and np.allclose(a[1], g[1], atol=1e-7, rtol=0)


def test_c3_sharded_over_thread_ranks(ctx, c3):
    """C3 split over 5 ranks (threads on this one GPU, tests/test_gpu_sharded.py): 10 keyframes per rank,
    5760+ tiles, the hierarchical reductions -- same LM decisions and result as the plain solve."""
    import threading
    p, e, cam, qt = c3
    ref = _solve(ctx, c3, 3)
    world = 5
    group = nrs.LocalGroup(world)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            c = nrs.Context()
            c.comm_init_local(group, r)
            c.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
            tr = nrs.Trace(128)
            c.dba_optimize(3, tr)
            pq, xyz = c.dba_download()
            out[r] = (pq, xyz.astype(np.float32), tr.trials)
            c.close()
        except Exception as ex:
            errs.append((r, ex))
            raise
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    assert not errs and all(o is not None for o in out), errs
    group.close()
    pq, xyz, trials = out[0]
    assert [t["accepted"] for t in trials] == [t["accepted"] for t in ref[2]]
    assert all(abs(s["lam"] - t["lam"]) <= 1e-6 * t["lam"] and abs(s["chi"] - t["chi"]) <= 1e-6 * t["chi"] for s, t in zip(trials, ref[2]))
    assert np.allclose(pq, ref[0], atol=1e-6, rtol=0) and np.allclose(xyz, ref[1], atol=1e-4, rtol=0)
    assert np.array_equal(out[4][0], pq) and np.array_equal(out[4][1], xyz)


def test_host_packing_is_thread_count_independent(ctx, c3, monkeypatch):
    """engine set-up runs on a few host threads (row layout, incidence packing, edge lists, masks): the packed problem --
    hence every sum of the solve -- is the same bits for any thread count"""
    p, e, cam, qt = c3
    runs = []
    for nt in ("1", "7", "16"):
        nrs.debug_set("NRS_HOST_THREADS", nt)
        ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
        tr = nrs.Trace()
        ctx.dba_optimize(2, tr)
        runs.append((tr.trials, *ctx.dba_download(), ctx.dba_stats()))
    for r in runs[1:]:
        key = lambda tr: [(t["accepted"], t["inner"], t["lam"], t["chi"], t["chi_new"]) for t in tr]
        assert key(r[0]) == key(runs[0][0]) and np.array_equal(r[1], runs[0][1]) and np.array_equal(r[2], runs[0][2])
        assert r[3] == runs[0][3]
