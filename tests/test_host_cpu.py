"""CPU-side checks: the C-ABI library loads and exports every symbol of include/nrs.h, the host
edge builder reproduces the oracle's restatement of OPT:927-1137 index for index, and compute
entry points refuse to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import nrs_oracle as O
import nrs_synth as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols(lib_built):
    nrs = lib_built
    lib = nrs.load_library()
    hdr = open(os.path.join(ROOT, "include", "nrs.h")).read()
    declared = set(re.findall(r"\b(nrs_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(nrs.SYMBOLS), declared ^ set(nrs.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_no_cpu_fallback(lib_built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    nrs = lib_built
    with pytest.raises(nrs.NrsError) as ei:
        nrs.Context()
    assert ei.value.code == -2            # NRS_ERR_NO_DEVICE


@pytest.mark.parametrize("n,k,seed,bad", [(150, 3, 1, 0.0), (400, 5, 2, 0.1), (400, 2, 3, 0.3), (50, 1, 4, 0.0)])
def test_edge_builder_matches_oracle(lib_built, n, k, seed, bad):
    nrs = lib_built
    p = S.make_dba_problem(n, k, seed, dropout=0.15)
    g = dict(p["nbr"])
    st = g["status"].copy()
    rng = np.random.default_rng(seed)
    st[rng.uniform(size=len(st)) < bad] = S.GRAPH_BAD
    g["status"] = st
    eo = O.dba_build(p["kf_points"], g["rowptr"], g["col"], g["w"], g["d0"], g["status"])
    e = nrs.dba_build_edges(p["kf_points"], g)
    for key in ("sp_ij", "sp_d0", "dm_idx", "dm_w"):
        assert np.array_equal(e[key], eo[key]), key          # integer/index work: bit-exact
    assert np.array_equal(eo["lm_kf"], p["lm_kf"]) and np.array_equal(eo["lm_pt"], p["lm_pt"])
    if k == 1:
        assert len(e["dm_idx"]) == 0


def test_edge_builder_matches_oracle_full_size_c2(lib_built):
    """BASELINE configs[1] at full size (91 749 landmarks, 570k springs, 513k dampers): the product's host builder
    against the oracle's restatement of OPT:927-1137, index for index.  The C2 golden and the GPU tests feed the
    ORACLE's edge list; this is what makes the product's equal to it."""
    nrs = lib_built
    p = S.make_dba_problem("C2")
    g = p["nbr"]
    eo = O.dba_build(p["kf_points"], g["rowptr"], g["col"], g["w"], g["d0"], g["status"])
    e = nrs.dba_build_edges(p["kf_points"], g)
    for key in ("sp_ij", "sp_d0", "dm_idx", "dm_w"):
        assert np.array_equal(e[key], eo[key]), key
    assert np.array_equal(eo["lm_kf"], p["lm_kf"]) and len(eo["sp_ij"]) > 500000


def test_edge_builder_rejects_bad_input(lib_built):
    nrs = lib_built
    p = S.make_dba_problem(60, 2, 5)
    g = dict(p["nbr"])
    g["col"] = g["col"].copy()
    g["col"][0] = 10 ** 6
    with pytest.raises(nrs.NrsError):
        nrs.dba_build_edges(p["kf_points"], g)


def test_graph_wire_format():
    """The synthetic graph is in the ordered form GetEdges returns (regularization_graph.cc:61-87):
    symmetric, rows sorted by weight descending, cut at min_weight, no float ties."""
    sc = S.make_scene(800, 2, 9)
    G = sc["graph"]
    g = S.ordered_view(G)
    rp, col, w = g["rowptr"], g["col"], g["w"]
    assert abs(G["min_w"] - float(O.min_weight(G["sigma"]))) < 1e-7
    pairs = set()
    for i in range(len(rp) - 1):
        ww = w[rp[i]:rp[i + 1]]
        assert np.all(np.diff(ww) < 0), "ties or wrong order in row %d" % i
        assert np.all(ww >= G["min_w"])
        for c in col[rp[i]:rp[i + 1]]:
            pairs.add((i, int(c)))
    assert all((b, a) in pairs for a, b in pairs)
    # weights are the reference's fp32 InterpolationWeight of the rest distance
    assert np.array_equal(G["e_w"], O.interpolation_weight(G["e_d0"], G["sigma"]))
    # the oracle's GetEdges on the raw (index-ordered) row reproduces the ordered row, also after
    # some edges went BAD / lost weight
    rng = np.random.default_rng(3)
    G["e_status"][rng.uniform(size=len(G["e_status"])) < 0.2] = S.GRAPH_BAD
    G["e_w"][rng.uniform(size=len(G["e_w"])) < 0.2] *= np.float32(0.4)
    G.update(S.ordered_neighbours(G))
    for i in (0, 17, 399, 799):
        sl = slice(G["rowptr"][i], G["rowptr"][i + 1])
        e = G["eid"][sl]
        pos = O.get_edges(G["col"][sl], G["e_w"][e], G["e_status"][e], G["min_w"])
        assert np.array_equal(G["col"][sl][pos], G["o_col"][G["o_rowptr"][i]:G["o_rowptr"][i + 1]])


def test_frame_loop_harness_with_the_oracle_backend():
    """SURVEY.md 8(f1) host logic on CPU: the frame loop (motion-model seed, point reuse, keyframe
    cadence) driven by the oracle backend on a tiny consistent sequence."""
    import nrs_frame_loop as FL
    import nrs_synth as S
    from frame_loop_backend import OracleBackend
    sq = S.make_frame_sequence(90, 4, 9)
    opts = dict(win=21, max_level=4, max_iters=10, epsilon=1e-4, min_eig=1e-4)
    loop = FL.FrameLoop(OracleBackend(sq["model"], sq["prm"], opts), lambda pc: FL.project_f32(sq["model"], sq["prm"], pc),
                        sq["wh"], sq["scale"], sq["kp0"], sq["X0"], sq["graph"], sq["pose_q"][0], sq["pose_t"][0],
                        sq["images"][0], images_to_insert_keyframe=1)
    for f in range(1, 4):
        assert loop.track_image(sq["images"][f])
    assert [L["keyframe"] for L in loop.log] == [False, True, False]
    assert loop.log[-1]["n_tracked"] > 0.7 * sq["n_points"]
    # SE3f algebra of the motion model: T * T^-1 = identity to float accuracy
    q, t = FL.se3f_mul(loop.pose, FL.se3f_inv(loop.pose))
    assert np.allclose(q, [0, 0, 0, 1], atol=1e-6) and np.allclose(t, 0, atol=1e-5)


@pytest.mark.parametrize("counts,world", [([300] * 8, 2), ([300] * 8, 8), ([100, 900, 100, 100, 2000, 10, 10], 3),
                                          ([5] * 5, 5), ([4000, 1, 1, 1], 4), ([256, 257, 512, 1, 700, 90], 1)])
def test_shard_plan_partitions_keyframes(lib_built, counts, world):
    """nrs_shard_plan (host only): contiguous keyframe ranges that cover the window, every rank at least
    one keyframe, and no rank's padded row count further from the ideal share than its largest
    neighbouring keyframe (boundaries are the closest keyframe boundary to the ideal split)."""
    nrs = lib_built
    lm_kf = np.repeat(np.arange(len(counts)), counts).astype(np.int32)
    kb = nrs.shard_plan(len(counts), lm_kf, world)
    assert kb[0] == 0 and kb[-1] == len(counts) and np.all(np.diff(kb) >= 1)
    rows = np.array([max(1, -(-c // 256)) for c in counts])
    cum = np.concatenate([[0], np.cumsum(rows)])
    for r in range(1, world):
        ideal = cum[-1] * r // world
        k = kb[r]
        forced_lo, forced_hi = kb[r - 1] + 1, len(counts) - (world - r)
        # a better boundary would have to be allowed by the "one keyframe per rank" rule
        for alt in (k - 1, k + 1):
            if forced_lo <= alt <= forced_hi:
                assert abs(cum[k] - ideal) <= abs(cum[alt] - ideal) or alt == k + 1 and cum[k] >= ideal
    assert nrs.shard_plan(3, np.zeros(0, np.int32), 3).tolist() == [0, 1, 2, 3]


def test_shard_plan_rejects_bad_arguments(lib_built):
    nrs = lib_built
    with pytest.raises(nrs.NrsError):
        nrs.shard_plan(2, np.array([0, 1], np.int32), 3)          # more ranks than keyframes
    with pytest.raises(nrs.NrsError):
        nrs.shard_plan(2, np.array([0, 2], np.int32), 2)          # keyframe index out of range


def test_comm_needs_a_context(lib_built):
    """The communicator entry points take a context; with a null context they fail cleanly (no device needed)."""
    nrs = lib_built
    lib = nrs.load_library()
    assert lib.nrs_comm_init_local(None, None, 0) == -1
    assert lib.nrs_comm_init_rccl(None, 1, 0, None, 0) == -1
    g = nrs.LocalGroup(2)
    g.close()
    with pytest.raises(nrs.NrsError):
        nrs.LocalGroup(9)


def test_skinned_status_demotes_everything_but_the_nodes():
    """skinned mode (include/nrs.h N2): nodes keep TRACKED_WITH_3D, other tracked points of the frame become TRACKED, slots
    without a map point and other statuses are left alone"""
    import nrs
    f_status = np.array([0, 0, 0, 1, 3, 0, 2, 0], np.int32)
    f_map = np.array([5, 2, 7, 1, 0, -1, 3, 4], np.int32)
    st = nrs.skinned_status(f_status, f_map, [2, 4])
    assert st.tolist() == [1, 0, 1, 1, 3, 0, 2, 0]
    assert f_status.tolist() == [0, 0, 0, 1, 3, 0, 2, 0]                      # the input is not modified


def test_embedded_window_builder_matches_the_oracle(lib_built):
    """nrs_dba_build_edges_embedded (N2b) against oracle/embedded_oracle.py dba_build_embedded: node copies, springs / dampers
    between them, skinned observations with their node copies and normalised weights -- index for index, weights to the last bit;
    with every point a node: the lists of nrs_dba_build_edges."""
    import embedded_oracle as E
    nrs = lib_built
    for n, k, m, seed in ((300, 4, 40, 9), (500, 6, 77, 10), (200, 3, 200, 11)):
        p = S.make_dba_problem(n, k, seed)
        flag, nb = S.embedded_problem(p, m)
        a = nrs.dba_build_edges_embedded(p["kf_points"], flag, nb)
        b = E.dba_build_embedded(p["kf_points"], flag, nb["rowptr"], nb["col"], nb["w"], nb["d0"], nb["status"])
        for key in ("lm_obs", "sp_ij", "sp_d0", "dm_idx", "dm_w", "sk_obs", "sk_node", "sk_omega"):
            assert np.array_equal(a[key], b[key]), key
        if m == n:
            e = nrs.dba_build_edges(p["kf_points"], nb)
            assert len(a["sk_obs"]) == 0 and all(np.array_equal(a[key], e[key]) for key in ("sp_ij", "sp_d0", "dm_idx", "dm_w"))
