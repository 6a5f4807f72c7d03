"""GPU parity of the embedded BA window's keyframe-block factorisation (nrs_options.embedded_solver = 0; nr-slam_amd/csrc/nrs_engine_kft.hpp) --
the GPU counterpart of g2o's exact solve per LM trial (linear_solver_eigen.h:92-173) -- against the oracle's assembled system
(oracle/embedded_oracle.py dba_graph_embedded -> Graph.build_system): every diagonal block A_k, every coupling T_k, and the factorisation
applied to a vector, M^-1 (H + lambda I) x = x."""
import numpy as np
import pytest

import embedded_oracle as E
import nrs
import nrs_synth as S

pytestmark = pytest.mark.gpu


def _setup(n, k, m, seed, model=S.PINHOLE):
    p = S.make_dba_problem(n, k, seed, model)
    flag, nb = S.embedded_problem(p, m)
    e = nrs.dba_build_edges_embedded(p["kf_points"], flag, nb)
    w = S.embedded_window(p, e)
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    G, _ = E.dba_graph_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"],
                                e["dm_idx"], e["dm_w"], w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"])
    G.initialize(0)
    G.compute_active_errors()
    H, b = G.build_system()
    return p, e, w, cam, qt, H.tocsr(), b


@pytest.mark.parametrize("n,k,m,seed,model", [(300, 4, 40, 61, S.PINHOLE), (500, 5, 70, 62, S.KB8), (600, 7, 90, 63, S.PINHOLE)])
def test_blocks_couplings_and_the_applied_factorisation_match_the_oracle_system(ctx, n, k, m, seed, model):
    p, e, w, cam, qt, H, b = _setup(n, k, m, seed, model)
    ctx.dba_upload_embedded(cam, qt, w, e, p["scale"])
    info = ctx.debug_kft_info()
    assert info["on"] and info["K"] == k and info["ld"] % 64 == 0
    idx = ctx.debug_kft_index()                                    # node copy -> (keyframe, compact node)
    assert (idx[:, 0] == w["lm_kf"]).all() and (idx[:, 1] >= 0).all()
    lam = 1e-5 * np.abs(H.diagonal()).max()
    K6 = 6 * k
    scale = np.abs(H.diagonal()).max()
    for kf in range(k):
        nf = int(info["nf"][kf])
        mine = np.where(idx[:, 0] == kf)[0]
        assert len(mine) == nf
        # unknowns of keyframe kf in the device's compact order -> the oracle's indices (6 per pose, then 3 per node copy)
        cols = np.zeros(3 * nf + 6, np.int64)
        for v in mine:
            cols[3 * idx[v, 1]:3 * idx[v, 1] + 3] = K6 + 3 * v + np.arange(3)
        cols[3 * nf:] = 6 * kf + np.arange(6)
        A = ctx.debug_kft_block(lam, kf)
        ref = H[cols][:, cols].toarray() + lam * np.eye(len(cols))
        nk = len(cols)
        assert np.abs(A[:nk, :nk] - ref).max() <= 1e-9 * scale, kf
        assert np.array_equal(A[nk:, nk:], np.eye(info["ld"] - nk)) and not A[:nk, nk:].any() and not A[nk:, :nk].any()
        if kf + 1 < k:
            nxt = np.where(idx[:, 0] == kf + 1)[0]
            cn = np.zeros(3 * len(nxt), np.int64)
            for v in nxt:
                cn[3 * idx[v, 1]:3 * idx[v, 1] + 3] = K6 + 3 * v + np.arange(3)
            T = ctx.debug_kft_block(lam, kf, coupling=True)
            reft = H[cols[:3 * nf]][:, cn].toarray()
            assert np.abs(T[:3 * nf, :len(cn)] - reft).max() <= 1e-9 * scale, kf
            assert not T[3 * nf:].any() and not T[:, len(cn):].any()
            assert not H[cols[3 * nf:]][:, cn].nnz                 # (poses do not couple across keyframes)
    # the factorisation applied to a vector: M^-1 (H + lam I) x = x
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 1, H.shape[0])
    r = H @ x + lam * x
    u = ctx.debug_kft_apply(lam, r)
    assert np.linalg.norm(u - x) <= 1e-7 * np.linalg.norm(x)
    # ... and to the window's own right-hand side: the LM step of the oracle
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    step = spla.spsolve((H + lam * sp.identity(H.shape[0])).tocsc(), b)
    u = ctx.debug_kft_apply(lam, b)
    assert np.linalg.norm(u - step) <= 1e-7 * np.linalg.norm(step)


@pytest.mark.parametrize("k", [1, 2, 3, 5])
def test_short_windows_and_both_launch_forms(ctx_emb_direct, k):
    """one, two and three keyframes (no chain / one chain / both chains meeting at once) and an odd count: M^-1 (H + lam I) x = x, the
    solve against the oracle; the two-launch form of a sweep step (NRS_KFT_TWO_LAUNCHES=1) gives the same result up to rounding"""
    c = ctx_emb_direct
    p, e, w, cam, qt, H, b = _setup(260, k, 36, 70 + k)
    c.dba_upload_embedded(cam, qt, w, e, p["scale"])
    assert c.debug_kft_info()["on"]
    lam = 1e-5 * np.abs(H.diagonal()).max()
    rng = np.random.default_rng(k)
    x = rng.normal(0, 1, H.shape[0])
    u1 = c.debug_kft_apply(lam, H @ x + lam * x)
    assert np.linalg.norm(u1 - x) <= 1e-7 * np.linalg.norm(x)
    nrs.debug_set("NRS_KFT_TWO_LAUNCHES", "1")
    u2 = c.debug_kft_apply(lam, H @ x + lam * x)
    nrs.debug_set("NRS_KFT_TWO_LAUNCHES", None)
    assert np.linalg.norm(u2 - x) <= 1e-7 * np.linalg.norm(x) and np.linalg.norm(u1 - u2) <= 1e-9 * np.linalg.norm(x)
    tr, otr = nrs.Trace(), []
    pq, xyz, sk = c.dba_solve_embedded(cam, qt, w, e, p["scale"], 5, tr)
    oq, ot, opts, osk, nit = E.dba_solve_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"],
                                                  e["dm_idx"], e["dm_w"], w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"], 5, otr)
    assert tr.iterations == nit and [t["accepted"] for t in tr.trials] == [t["accepted"] for t in otr]
    assert sum(t["inner"] for t in tr.trials) <= 3 * len(tr.trials)
    assert np.allclose(pq[:, :4], oq, atol=1e-6, rtol=0) and np.allclose(pq[:, 4:], ot, atol=1e-5, rtol=0)
    assert np.allclose(xyz, opts, atol=1e-4, rtol=0) and np.allclose(sk, osk, atol=1e-4, rtol=0)


def test_factorisation_is_bit_reproducible(ctx_emb_direct):
    c = ctx_emb_direct
    p, e, w, cam, qt, H, b = _setup(400, 5, 60, 81)
    c.dba_upload_embedded(cam, qt, w, e, p["scale"])
    lam = 1e-5 * np.abs(H.diagonal()).max()
    u = [c.debug_kft_apply(lam, b) for _ in range(3)]
    assert np.array_equal(u[0], u[1]) and np.array_equal(u[0], u[2])


def test_residual_test_after_the_first_step_gives_the_iterate_of_the_pcgs_own_test(ctx_emb_direct):
    """the PCG's first step with the factorisation as preconditioner is tested by its residual (k_kft_rnorm) instead of by r . M^-1 r an
    iteration later (NRS_KFT_NO_RESIDUAL_TEST=1): the same iterate, bit for bit, the same trials; and the sweep of the pivot block in
    16-pivot steps against the 4-pivot register form (NRS_KFT_SCALAR_SWEEP=1): the same solve up to rounding; eight-wave panel workgroups
    against four-wave ones (NRS_KFT_FOUR_WAVES=1): bit-identical"""
    c = ctx_emb_direct
    p, e, w, cam, qt, H, b = _setup(420, 5, 64, 83)
    out = []
    for name in (None, "NRS_KFT_NO_RESIDUAL_TEST", "NRS_KFT_SCALAR_SWEEP", "NRS_KFT_FOUR_WAVES"):
        if name:
            nrs.debug_set(name, "1")
        try:
            tr = nrs.Trace()
            pq, xyz, sk = c.dba_solve_embedded(cam, qt, w, e, p["scale"], 5, tr)
        finally:
            if name:
                nrs.debug_set(name, None)
        out.append((pq, xyz, sk, [(t["accepted"], t["chi"], t["chi_new"], t["lam"], t["inner"]) for t in tr.trials]))
    assert all(t[4] == 1 for t in out[0][3])                       # (one step per trial: the factorisation is exact)
    assert out[0][3] == out[1][3] and all(np.array_equal(a, b2) for a, b2 in zip(out[0][:3], out[1][:3]))
    # a panel workgroup of eight waves (four take the C_I half of the look-ahead) against four: the same arithmetic, the same bits
    assert out[0][3] == out[3][3] and all(np.array_equal(a, b2) for a, b2 in zip(out[0][:3], out[3][:3]))
    assert [t[0] for t in out[0][3]] == [t[0] for t in out[2][3]]
    assert np.allclose(out[0][0], out[2][0], atol=1e-9, rtol=0) and np.allclose(out[0][1], out[2][1], atol=1e-8, rtol=0)


@pytest.mark.parametrize("n,k,m,seed", [(90, 2, 8, 93), (200, 6, 21, 94), (400, 9, 45, 95)])
def test_single_block_and_mixed_block_counts(ctx_emb_direct, n, k, m, seed):
    """keyframe blocks of ONE 64-pivot block (a sweep step without trailing tiles) and windows whose keyframes need different block counts
    (40 .. 43 node copies: 126 .. 135 unknowns, two or three blocks -- the two chains of a step then sweep different counts): M^-1 (H + lam I) x = x"""
    c = ctx_emb_direct
    p, e, w, cam, qt, H, b = _setup(n, k, m, seed)
    c.dba_upload_embedded(cam, qt, w, e, p["scale"])
    info = c.debug_kft_info()
    assert info["on"] and info["K"] == k
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 1, H.shape[0])
    for scl in (1e-5, 1e-7):
        lam = scl * np.abs(H.diagonal()).max()
        u = c.debug_kft_apply(lam, H @ x + lam * x)
        assert np.linalg.norm(u - x) <= (1e-7 if scl == 1e-5 else 1e-5) * np.linalg.norm(x)
