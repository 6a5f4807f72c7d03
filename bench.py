#!/usr/bin/env python3
"""bench.py -- deformable-BA LM iterations/sec on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic batch = one reference-shaped
LocalDeformableBundleAdjustment solve, optimize(5) (reference g2o_optimization.cc:1141-1143).  Inputs are uploaded
to HBM once, outside the timed region; each timed step is {reset estimates (device-to-device), optimize(5)}.

N = 1: BASELINE.json configs[1] (C2: 5k map points x 20 keyframes, pinhole), every point a deformation-graph node
(parity mode).  The line also carries `value_exact_trials` (every LM trial solved to 1e-10, as
g2o does; `value` uses the early-rejection heuristic of nrs_options), the rooflines of the two dominant kernels
(HIP events on the context's stream), the tracked-fps half of the metric and the CPU baseline (C++ restatement).

N > 1 (launched by torch.distributed.run, one rank per GPU): `value` comes from ONE window -- BASELINE.json
configs[3] (C4: 50k points x 200 keyframes) -- sharded over the N ranks by keyframes, in this process: RCCL
all-reduce of the pose blocks of the normal equations / the PCG sums + boundary-keyframe exchange per iteration
(include/nrs.h "multi-GPU"), "scaling": "strong".  Secondary fields: `independent_windows` (every rank its own C2
window, no data-path collective: what a node does with independent LocalDeformableBundleAdjustment calls) and
`single_gpu_same_window` (rank 0 alone on the same C4 window, for a like-for-like strong-scaling ratio).
A watchdog prints the line with the independent-window figure as `value` if the sharded section does not finish.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd", "py"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def unique_blocks(n_lm, sp_ij, dm_idx):
    """Number of distinct off-diagonal 3x3 landmark blocks of H_ll (upper triangle, as g2o stores
    it: block_solver.hpp:108-266)."""
    d = dm_idx.astype(np.int64)
    pairs = [sp_ij.astype(np.int64)]
    for a, b in ((0, 1), (2, 3), (0, 2), (1, 3), (0, 3), (1, 2)):
        pairs.append(d[:, [a, b]])
    p = np.concatenate(pairs)
    lo, hi = np.minimum(p[:, 0], p[:, 1]), np.maximum(p[:, 0], p[:, 1])
    return int(len(np.unique(lo * n_lm + hi)))


def algorithmic_bytes(n_lm, n_sp, n_dm, n_blocks):
    """SURVEY.md 8(d) per-unit figures (fp32 storage of inputs and H blocks, each input once,
    each distinct output once) x the units one launch processes."""
    lin = 136 * n_lm + 48 * n_sp + 24 * n_dm              # fused linearise+assemble, BA form
    spmv = 40 * (n_blocks + n_lm) + 76 * n_lm             # BSR SpMV: 3x3 blocks (off-diag + diag), 6x3 H_pl
    return lin, spmv


def gpu_iters_per_s(ctx, p, e, reps=10):
    """the product path, resident like `value`, on a given window (for like-for-like ratios next to CPU samples)"""
    import nrs
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    ctx.dba_optimize(5)
    g_it = 0
    t1 = time.perf_counter()
    for _ in range(reps):
        ctx.dba_reset()
        tr = nrs.Trace(64)
        ctx.dba_optimize(5, tr)
        g_it += tr.iterations
    return g_it / (time.perf_counter() - t1)


def cpu_baseline(p2, e2, ctx=None, ctx_exact=None):
    """The CPU side of the metric on THIS host: the C++ restatement of the reference's deformable BA
    (oracle/nrs_cpu.cpp, kind "port"; rebuilt here with -O3 -march=native; checked against the NumPy oracle, the
    g2o known-answer system and the C2 golden in tests/test_oracle_cpp_cpu.py), on bounded samples:
      value            C2 itself, optimize(5), block-Jacobi PCG to 1e-10 on all cores (OpenMP): the fastest CPU
                       form of the same LM (same trials, same iterates as the GPU's exact mode)
      one_core         the reference's own window cap (5 keyframes x 5000 points), PCG, 1 core
      sparse_cholesky  what the reference actually runs (full sparse Cholesky per LM trial, 1 thread: g2o OpenMP is
                       off): AMD-ordered block Cholesky on 1000 points x 5 keyframes; C2 itself needs 8.3 TFLOP per
                       factorisation (symbolic count, profiles/r02_cpu_baseline.json) = hours per optimize(5)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nrs
    import nrs_cpu as CPU
    import nrs_synth as S
    try:
        CPU.build(native=True)
        lib, flags = CPU.load(native=True), "-O3 -march=native -fopenmp"
    except Exception:                                        # no compiler on this host: the copy that travelled with the repo
        lib, flags = CPU.load(), "-O3 -march=x86-64-v3 -fopenmp"
    cores = CPU.max_threads(lib)

    def run(p, e, solver, threads, max_trials=0):
        t0 = time.perf_counter()
        _, _, _, tr, st = CPU.dba_solve(p["model"], p["prm"], p["poses_q"], p["poses_t"], p["lm_xyz"], p["lm_kf"], p["lm_uv"],
                                        e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"], p["scale"], 5, solver, 1e-10, threads, max_trials, lib)
        return time.perf_counter() - t0, st

    def shape(p, e):
        return "%d landmarks, %d springs, %d dampers" % (len(p["lm_kf"]), len(e["sp_ij"]), len(e["dm_idx"]))
    # the PCG is memory-bound: more threads than memory channels make it slower (128 threads on the GPU box: 16 s, 32: ~5 s).
    # The thread count is the fastest of a short sweep on the first LM trial of C2.
    cand = sorted({t for t in (8, 16, 32, 64, cores) if t <= cores})
    best = min(cand, key=lambda t: run(p2, e2, 1, t, 1)[1]["t_solve"])
    avail, cores = cores, best
    dt, st = run(p2, e2, 1, cores)
    out = dict(value=st["n_iters"] / dt, unit="LM iters/s", cores=cores, cores_available=avail, kind="port",
               sample="C2 itself (%s), one optimize(5): %d LM trials, %d PCG iterations, %.1f s; C++ restatement "
                      "oracle/nrs_cpu.cpp (%s), block-Jacobi PCG 1e-10, OpenMP %d threads"
                      % (shape(p2, e2), st["n_trials"], st["n_pcg_iters"], dt, flags, cores),
               linearize_ms_per_lm_iter=1e3 * st["t_linearize"] / max(1, st["n_iters"]))
    if ctx_exact is not None:
        out["gpu_same_sample_exact_trials"] = gpu_iters_per_s(ctx_exact, p2, e2, 3)       # every trial solved to 1e-10 on both sides
    p5 = S.make_dba_problem(5000, 5, 1)
    e5 = nrs.dba_build_edges(p5["kf_points"], p5["nbr"])
    dt, st = run(p5, e5, 1, 1)
    out["one_core"] = dict(value=st["n_iters"] / dt, unit="LM iters/s", cores=1, solver="block-Jacobi PCG 1e-10",
                           sample="5000 points x 5 keyframes = the reference's window cap (%s), optimize(5), %.1f s" % (shape(p5, e5), dt))
    if ctx is not None:
        out["one_core"]["gpu_same_sample"] = gpu_iters_per_s(ctx, p5, e5)
    p1 = S.make_dba_problem(1000, 5, 1)
    e1 = nrs.dba_build_edges(p1["kf_points"], p1["nbr"])
    dt, st = run(p1, e1, 0, 1)
    out["sparse_cholesky"] = dict(value=st["n_iters"] / dt, unit="LM iters/s", cores=1,
                                  solver="full sparse block Cholesky per LM trial, AMD ordering on the block pattern (what g2o + Eigen SimplicialLLT do)",
                                  sample="1000 points x 5 keyframes (%s), optimize(5): %d factorisations of %.1f GFLOP at %.1f GFLOP/s, %.1f s"
                                         % (shape(p1, e1), st["n_factor"], st["chol_flops"] / 1e9,
                                            st["chol_flops"] * st["n_factor"] / max(1e-9, st["t_factor"]) / 1e9, dt))
    if ctx is not None:
        out["sparse_cholesky"]["gpu_same_sample"] = gpu_iters_per_s(ctx, p1, e1)
    comp = cpu_tracked_fps_compiled(lib, ctx is not None)
    big = comp[-1]                                               # the frame at the bench's point count (5000 map points, ~4.5k tracked)
    out["tracked_fps"] = dict(value=big["cpu_frames_per_s"], unit="frames/s", cores=1, kind="port",
                              sample="1 tracked frame of %d tracked points (LK Track + a1 + a2), C++ restatement with a full sparse Cholesky per LM trial "
                                     "(oracle/nrs_cpu_track.hpp, oracle/nrs_cpu_lk.hpp), %.0f ms" % (big["tracked"], big["cpu_ms"]),
                              gpu_same_sample=big.get("gpu_frames_per_s"), compiled=comp, oracle_numpy=cpu_tracked_fps(ctx is not None))
    return out


def cpu_tracked_fps_compiled(lib, with_gpu):
    """A tracked frame = LK Track + a1 CameraPoseOptimization + a2 CameraPoseAndDeformationOptimization (with its graph walks
    and graph update), SURVEY.md 8d's unit, in the C++ restatements (oracle/nrs_cpu_lk.hpp, oracle/nrs_cpu_track.hpp, kind
    "port": g2o's LM with a full AMD-ordered sparse Cholesky per trial, which is what the reference runs; 1 core, as the
    reference) on single synthetic frames of 600 / 1150 / 5000 map points, next to the product's three calls on the same
    inputs (flat kNN-16 graph, host buffers in and out).  `value` of the enclosing object is the NumPy-driven whole loop."""
    import nrs
    import nrs_cpu as CPU
    import nrs_synth as S
    out = []
    for n in (600, 1150, 5000):
        tp = S.make_tracking_problem(n, 5)
        m = tp["status"] == 0
        fm = np.arange(n)
        t0 = time.perf_counter()
        q, t, _, _, _ = CPU.pose_only_solve(tp["model"], tp["prm"], tp["uv"][m], tp["X_prev"][m], tp["pose_q"], tp["pose_t"], lib)
        r = CPU.track_deform_solve(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], q, t, tp["scale"], lib)
        dt = time.perf_counter() - t0
        st = r["stats"]
        # LK data association of the same number of points on a 640x480 pair (LucasKanadeTracker::Track, templates cached)
        sq = S.make_lk_sequence(int(m.sum()), 5)
        lk = CPU.LucasKanadeCpp(lib=lib)
        lk.set_reference(sq["im0"], sq["pts"])
        t0 = time.perf_counter()
        lk.track(sq["im1"], sq["pts"], np.zeros(len(sq["pts"]), np.int32))
        dt_lk = time.perf_counter() - t0
        lk.close()
        dt += dt_lk
        row = dict(map_points=n, tracked=int(m.sum()), cpu_ms=1e3 * dt, cpu_lk_track_ms=1e3 * dt_lk, cpu_frames_per_s=1.0 / dt, cores=1, kind="port",
                   lm_trials=st["n_trials"], factorisations=st["n_factor"], gflop_per_factorisation=st["chol_flops"] / 1e9,
                   cpu_factor_ms=1e3 * st["t_factor"])
        if with_gpu:
            c = nrs.Context()
            cam = nrs.make_camera(tp["model"], tp["prm"])
            c.klt_configure()
            c.klt_set_reference(sq["im0"], sq["pts"])
            def three_calls(reps):
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    c.klt_track(sq["im1"], sq["pts"], np.zeros(len(sq["pts"]), np.int32))
                    gq, gt, _ = c.pose_only_solve(cam, tp["uv"][m], tp["X_prev"][m], tp["pose_q"], tp["pose_t"])
                    c.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], gq, gt, tp["scale"])
                    ts.append(time.perf_counter() - t0)
                return ts
            # the same frame is repeated: without the switch every repeat after the first would take the direct solver's symbolic
            # factorisation from the context's cache.  gpu_ms is a frame that builds it (what the CPU side does per frame, too);
            # gpu_ms_plan_reused a frame whose structure equals an earlier one's
            nrs.debug_set("NRS_ND_NO_CACHE", "1")
            try:
                ts = three_calls(3)
            finally:
                nrs.debug_set("NRS_ND_NO_CACHE", None)
            ts_hit = three_calls(3)[1:]
            c.close()
            row.update(gpu_ms=1e3 * min(ts), gpu_frames_per_s=1.0 / min(ts), gpu_over_cpu=dt / min(ts), gpu_ms_plan_reused=1e3 * min(ts_hit),
                       gpu_over_cpu_plan_reused=dt / min(ts_hit))
            c = nrs.Context(direct_solve=2)                         # the same three calls with the PCG as a2's linear solver
            c.klt_configure()
            c.klt_set_reference(sq["im0"], sq["pts"])
            ts = []
            for _ in range(2):
                t0 = time.perf_counter()
                c.klt_track(sq["im1"], sq["pts"], np.zeros(len(sq["pts"]), np.int32))
                gq, gt, _ = c.pose_only_solve(cam, tp["uv"][m], tp["X_prev"][m], tp["pose_q"], tp["pose_t"])
                c.track_deform_solve(cam, tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], gq, gt, tp["scale"])
                ts.append(time.perf_counter() - t0)
            c.close()
            row.update(gpu_ms_pcg_solver=1e3 * min(ts))
        out.append(row)
    return out


def cpu_embedded_frame(gpu_frames_per_s):
    """cpu_baseline of the embedded frame loop (tracked_fps_5k_x_500): ONE frame of 5000 map points with 500 nodes in the C++ restatement
    (oracle/nrs_cpu_lk.hpp LK Track, a1, and the embedded a2 of oracle/nrs_cpu.cpp nrs_cpu_track_deform_solve_embedded -- held to
    oracle/embedded_oracle.py by tests/test_oracle_cpp_track_cpu.py), 1 core, full sparse Cholesky per LM trial.  The CPU side walks a flat
    kNN-96 graph (~10 nodes among a point's neighbours; the GPU leg's all-pairs graph is 25 M connections on a CPU): the same unknowns
    (6 + 3 x 500) and the same observations."""
    import nrs_cpu as CPU
    import nrs_synth as S
    import skin_oracle as K
    lib = CPU.load(native=True)
    n = 5000
    tp = S.make_tracking_problem(n, 3)
    g = S.build_graph(tp["X_prev"], tp["graph"]["sigma"], 96, tp["graph"]["stretch_th"])
    node = np.zeros(n, np.uint8)
    node[K.select_nodes(tp["X_prev"], 500, tp["status"] == 0)] = 1
    m = tp["status"] == 0
    fm = np.arange(n)
    t0 = time.perf_counter()
    q, t, _, _, _ = CPU.pose_only_solve(tp["model"], tp["prm"], tp["uv"][m], tp["X_prev"][m], tp["pose_q"], tp["pose_t"], lib)
    r = CPU.track_deform_solve_embedded(tp["model"], tp["prm"], g, tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], node, q, t, tp["scale"], lib)
    dt = time.perf_counter() - t0
    sq = S.make_lk_sequence(int(m.sum()), 5)
    lk = CPU.LucasKanadeCpp(lib=lib)
    lk.set_reference(sq["im0"], sq["pts"])
    t0 = time.perf_counter()
    lk.track(sq["im1"], sq["pts"], np.zeros(len(sq["pts"]), np.int32))
    dt_lk = time.perf_counter() - t0
    lk.close()
    st = r["stats"]
    return dict(value=1.0 / (dt + dt_lk), unit="frames/s", cores=1, kind="port",
                sample="1 frame, %d tracked points, %d nodes, %d skinned points: LK Track %.0f ms + a1 + embedded a2 %.0f ms (%d LM trials, %d factorisations of "
                       "%.2f GFLOP, %d unknowns)" % (int(m.sum()), r["n_nodes"], r["n_skinned"], 1e3 * dt_lk, 1e3 * dt, st["n_trials"], st["n_factor"],
                                                    st["chol_flops"] / 1e9, st["unknowns_max"]),
                gpu_over_cpu=gpu_frames_per_s * (dt + dt_lk))


def cpu_tracked_fps(with_gpu, n_points=600):
    """The frame loop driven by the oracle (1 core) on a bounded sample: ONE tracked frame of a 640x480
    sequence with 600 map points (about 10 s; the metric's 4.4k points take minutes per frame in NumPy),
    and the product path on the very same frames."""
    import nrs
    import nrs_frame_loop as FL
    import nrs_synth as S
    from frame_loop_backend import OracleBackend
    sq = S.make_frame_sequence(n_points, 3, 21)
    opts = dict(win=21, max_level=4, max_iters=10, epsilon=1e-4, min_eig=1e-4)

    def run(backend, frames):
        loop = FL.FrameLoop(backend, lambda pc: FL.project_f32(sq["model"], sq["prm"], pc), sq["wh"], sq["scale"], sq["kp0"],
                            sq["X0"], sq["graph"], sq["pose_q"][0], sq["pose_t"][0], sq["images"][0])
        ts = []
        for f in frames:
            t0 = time.perf_counter()
            loop.track_image(sq["images"][f])
            ts.append(time.perf_counter() - t0)
        return ts
    t = run(OracleBackend(sq["model"], sq["prm"], opts, dense_graph=True), [1])      # all-pairs graph, as the product run below
    out = dict(value=1.0 / t[0], unit="frames/s", cores=1, kind="port",
               sample="1 tracked frame, %d map points, 640x480 (LK + pose-only + pose-and-deformation + point reuse), %.1f s" % (sq["n_points"], t[0]))
    if with_gpu:
        gb = FL.GpuBackend(nrs, sq["model"], sq["prm"], opts, dense_graph=True, cap_per_point=128)
        tg = run(gb, [1, 2])
        gb.close()
        out["gpu_same_sample"] = 1.0 / tg[0]
    return out


def reduce_over_ranks(dist, dt, units, device=None):
    """Whole-job figures from per-rank ones: elapsed = MAX over ranks, units = SUM over ranks.
    Works with any torch.distributed backend (RCCL on the GPU box, gloo in the CPU tests)."""
    if dist is None:
        return float(dt), float(units)
    import torch
    t = torch.tensor([float(dt)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


_SEQ_CACHE = {}


def frame_sequence(n_points, n_frames, seed=21):
    """the synthetic 640x480 sequence of a tracked-fps leg (nrs_synth.make_frame_sequence), generated once per (points, frames): the legs
    that differ in graph / solver / node set run on the very same images"""
    import nrs_synth as S
    key = (n_points, n_frames, seed)
    if key not in _SEQ_CACHE:
        _SEQ_CACHE[key] = S.make_frame_sequence(n_points, n_frames, seed)
    return _SEQ_CACHE[key]


def tracked_fps(n_points=5000, frames=31, dense_graph=False, direct_solve=0, n_nodes=0):
    """Secondary figure of BASELINE.json's metric: tracked frames/s, end to end through the frame-loop
    harness (nr-slam_amd/py/nrs_frame_loop.py = reference tracking.cc:72-112 minus image decode) on a consistent
    synthetic 640x480 sequence with n_points map points: LK data association, motion-model seed, pose-only solve,
    pose-and-deformation solve (with its graph update), PointReuse (tracking.cc:394-506: moving occluders hide points,
    which fail the tracker's SSIM gate and are re-found once the patch has moved on), keyframe insertion every sixth
    frame (Shi-Tomasi extraction, new LK reference, template archive).  frames - 1 frames are timed (the first warms the
    code objects up): `value` is their mean rate; median / p95 frame times and the per-stage means sit next to it.
    Every call takes host buffers (a frame arrives from the host): PCIe-inclusive."""
    import nrs
    import nrs_frame_loop as FL
    sq = frame_sequence(n_points, frames + 1)
    opts = dict(win=21, max_level=4, max_iters=10, epsilon=1e-4, min_eig=1e-4)
    # dense_graph: the map's graph at the reference's density (all pairs, resident on the device) instead of the generator's kNN-16
    # n_nodes > 0: the embedded-deformation mode of the pose-and-deformation solve (n_nodes map points carry the vertices, N2a)
    gb = FL.GpuBackend(nrs, sq["model"], sq["prm"], opts, dense_graph=dense_graph, cap_per_point=128, direct_solve=direct_solve, n_nodes=n_nodes)
    stage = {}

    def wrap(name):
        fn = getattr(gb, name)

        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            stage[name] = stage.get(name, 0.0) + time.perf_counter() - t0
            return r
        setattr(gb, name, w)
    # (PointReuse tracks its candidates from the device-side template archive: reuse_track_archived + insert_archived; a keyframe is
    # extract_features + klt_set_reference + archive_templates)
    for nme in ("klt_track", "pose_only", "track_deform", "reuse_track", "reuse_track_archived", "insert_archived", "extract_features",
                "klt_set_reference", "archive_templates"):
        wrap(nme)
    loop = FL.FrameLoop(gb, lambda pc: FL.project_f32(sq["model"], sq["prm"], pc), sq["wh"], sq["scale"], sq["kp0"], sq["X0"],
                        sq["graph"], sq["pose_q"][0], sq["pose_t"][0], sq["images"][0])
    ts, trials, inner = [], 0, 0
    for f in range(1, frames + 1):
        if f == 2:
            stage.clear()                                  # frame 1 warms the code objects up
        t0 = time.perf_counter()
        loop.track_image(sq["images"][f])
        if f > 1:
            ts.append(time.perf_counter() - t0)
            trials += len(gb.last_trace.trials)
            inner += sum(t["inner"] for t in gb.last_trace.trials)
    nd_reused, nd_built = gb.ctx.nd_cache_stats()
    gb.close()
    nf = len(ts)
    log = loop.log[1:]
    # the pose-and-deformation solve: per LM trial one linear solve -- the nested-dissection Cholesky (default, `inner` = 1 per trial) or,
    # with direct_solve=2, a run of single-launch PCG iterations bounded by launch latency (6.9 us floor per iteration)
    direct = inner <= trials
    us_unit = 1e6 * stage.get("track_deform", 0) / max(1, trials if direct else inner)
    latency = dict(linear_solver="nested-dissection multifrontal Cholesky (k_nd_level / k_nd_back, fronts on v_mfma_f64_16x16x4)" if direct
                   else "block-Jacobi / two-level PCG, one launch per iteration (k_pcg_fused)",
                   solves_per_frame=trials / nf, inner_iterations_per_frame=inner / nf,
                   lm_trials="inside a run of rejections the next trials go out as a batch on two shadow sets and their own streams (same trials, same bits; "
                             "NRS_SPEC_TRIALS=0: one at a time -- tools/spec_trials_probe.py, profiles/r06_a2_spec_trials.txt)" if direct else "one at a time",

                   us_all_in_per_lm_trial=us_unit if direct else None, us_all_in_per_pcg_iteration=None if direct else us_unit,
                   # symbolic factorisations over the whole sequence (two single-frame problems per frame): built anew / taken from the
                   # context's cache because the frame's optimised set, edges and fixed flags equalled an earlier frame's
                   symbolic_plans=dict(built=nd_built, reused=nd_reused))
    ms = 1e3 * np.sort(np.asarray(ts))
    reuse_s = stage.get("reuse_track", 0) + stage.get("reuse_track_archived", 0) + stage.get("insert_archived", 0)
    kf_s = stage.get("extract_features", 0) + stage.get("klt_set_reference", 0) + stage.get("archive_templates", 0)
    return dict(value=nf / sum(ts), unit="frames/s", points=int(sq["n_points"]), nodes=int(n_nodes) if n_nodes else int(sq["n_points"]), frames=nf,
                ms_per_frame_mean=float(ms.mean()), ms_per_frame_median=float(np.median(ms)), ms_per_frame_p95=float(ms[min(nf - 1, int(np.ceil(0.95 * nf)) - 1)]),
                ms_per_frame_max=float(ms[-1]), keyframes=int(sum(1 for l in log if l["keyframe"])),
                frames_with_point_reuse=int(sum(1 for l in log if l["reused"] > 0)), points_reused=int(sum(l["reused"] for l in log)),
                points_lost=int(sum(len(l["lost"]) for l in log)), a2_solver=latency,
                tracked_last_frame=int(loop.log[-1]["n_tracked"]), tracked_min=int(min(l["n_tracked"] for l in log)),
                ms_klt_track=1e3 * stage.get("klt_track", 0) / nf, ms_pose_only=1e3 * stage.get("pose_only", 0) / nf,
                ms_pose_and_deformation=1e3 * stage.get("track_deform", 0) / nf, ms_point_reuse=1e3 * reuse_s / nf,
                ms_keyframe_work=1e3 * kf_s / nf, ms_keyframe_extract=1e3 * stage.get("extract_features", 0) / nf,
                ms_host_harness=1e3 * (sum(ts) - sum(stage.values())) / nf, features_2d_last_frame=int(loop.log[-1]["n_2d"]),
                lm_trials_per_frame=trials / nf, pcg_iters_per_frame=inner / nf)


def flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def make_window(workload, seed_offset=0, n_points=None, n_kf=None):
    import nrs
    import nrs_synth as S
    np_, nk_, seed, model = S.CONFIGS[workload]
    p = S.make_dba_problem(n_points or np_, n_kf or nk_, seed + seed_offset, model)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    return p, e, nrs.make_camera(p["model"], p["prm"]), np.concatenate([p["poses_q"], p["poses_t"]], 1)


def timed_steps(ctx, steps, warmup, barrier):
    """W untimed steps, then exactly K timed steps bracketed by barrier(); returns this rank's figures"""
    import nrs
    for _ in range(warmup):
        ctx.dba_reset()
        ctx.dba_optimize(5)
    barrier()
    t0 = time.perf_counter()
    lm_iters = trials = inner = 0
    for _ in range(steps):
        ctx.dba_reset()
        tr = nrs.Trace(64)
        ctx.dba_optimize(5, tr)              # returns after the last host read-back: the stream is idle
        lm_iters += tr.iterations
        trials += tr.c.count
        inner += sum(t["inner"] for t in tr.trials)
    barrier()
    return dict(dt=time.perf_counter() - t0, lm_iters=lm_iters, trials=trials, inner=inner)


def run_sharded(dist, rank, world, local_rank, steps, warmup, barrier, window, dev):
    """ONE window over all ranks, in this process.  The RCCL unique id is made by rank 0's library and broadcast
    with torch.distributed; every rank uploads the same window and owns a contiguous keyframe range of it
    (nrs_shard_plan); reset / optimize are collective.  Returns this rank's timing."""
    import torch
    import nrs
    p, e, cam, qt = window
    msg = torch.zeros(1 + nrs.COMM_ID_BYTES, dtype=torch.uint8, device=dev)      # [ok, id bytes]
    if rank == 0:
        try:
            msg = torch.tensor([1] + list(nrs.comm_unique_id()), dtype=torch.uint8, device=dev)
        except Exception:                                       # librccl not loadable / no device: every rank must learn it
            pass
    dist.broadcast(msg, src=0)
    raw = bytes(msg.cpu().tolist())
    if raw[0] != 1:
        raise RuntimeError("rank 0 could not create an RCCL unique id")
    ctx = nrs.Context(device=local_rank)
    ctx.comm_init_rccl(world, rank, raw[1:])
    t_up = time.perf_counter()
    ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    t_up = time.perf_counter() - t_up
    res = timed_steps(ctx, steps, warmup, barrier)
    kb = nrs.shard_plan(len(qt), p["lm_kf"], world)
    res.update(upload_s=t_up, keyframes_of_rank=[int(kb[rank]), int(kb[rank + 1])], rccl_world=ctx.comm_rank()[1])
    ctx.close()
    return res


def sharded_section(dist, rank, world, local_rank, steps, warmup, barrier, window, dev):
    """run_sharded + agreement: returns (result, None) on every rank, or (None, error text) on every rank.  Failures
    that every rank sees (no device, librccl missing, a window that cannot be split) come back as errors; a rank that
    dies inside a collective is the watchdog's business (main)."""
    import torch
    err, res = None, None
    try:
        res = run_sharded(dist, rank, world, local_rank, steps, warmup, barrier, window, dev)
    except Exception as ex:
        err = "rank %d: %r" % (rank, ex)
    ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if ok.item() > 0:
        seen = torch.tensor([float(res["rccl_world"])], dtype=torch.float64, device=dev)
        dist.all_reduce(seen, op=dist.ReduceOp.MIN)
        res["rccl_ranks_seen"] = int(seen.item())
        return res, None
    return None, err or "another rank failed"


def rgraph_bench(n=5000):
    """a19 / a20 at the reference's density (all-pairs graph over n points): all-pairs initialisation,
    UpdateVertex of 90 % of the points (n - 1 connections each), GetEdges of every point -- host ids / positions in,
    results out (PCIe-inclusive) -- next to the NumPy restatement oracle/rgraph_oracle.py on a 600-point graph (1 core)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nrs
    import rgraph_oracle as RG
    rng = np.random.default_rng(5)
    pos = np.stack([rng.uniform(-22, 22, n), rng.uniform(-17, 17, n), 60 + rng.normal(0, 1, n)], 1).astype(np.float32)
    ids = np.arange(n, dtype=np.int32)
    upd = np.sort(rng.choice(n, int(0.9 * n), replace=False)).astype(np.int32)
    ctx = nrs.Context()
    g = nrs.RGraph(ctx, n, 2.2, 1.1)
    g.add_edges(pos, ids, ids)

    def t(fn, reps=5):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return 1e3 * (time.perf_counter() - t0) / reps
    out = dict(points=n, connections_per_point=n - 1, dense_state_MB=13 * n * n / 1e6,
               ms_add_all_pairs=t(lambda: g.add_edges(pos, ids, ids)),
               ms_update_vertices=t(lambda: g.update(pos + np.float32(0.01), upd)), updated_points=int(len(upd)),
               ms_get_edges_all_points=t(lambda: g.get_edges(ids, 256)))
    g.close()
    ctx.close()
    m = 600
    D = RG.DenseGraph(m, 2.2 * np.sqrt(5000 / m), 1.1)
    pm = pos[:m]
    t0 = time.perf_counter()
    D.add_edges(pm, np.arange(m), np.arange(m))
    t1 = time.perf_counter()
    for i in range(0, m, 2):
        D.update_vertex(pm, i)
    t2 = time.perf_counter()
    for i in range(m):
        D.get_edges(i)
    t3 = time.perf_counter()
    out["cpu_oracle"] = dict(points=m, kind="port (NumPy, 1 core)", ms_add_all_pairs=1e3 * (t1 - t0), ms_update_vertices=1e3 * (t2 - t1),
                             updated_points=m // 2, ms_get_edges_all_points=1e3 * (t3 - t2))
    return out


def skinned_bench(n=5000, m=500, n_kf=20):
    """N2, the metric's "5k pts x 500 graph nodes" read as the skinned mode (include/nrs.h): m farthest-point nodes carry the
    free variables, the other tracked points follow them through the reference's stage 2 (OPT:476-553).  (i) one frame of
    CameraPoseAndDeformationOptimization on the device-resident all-pairs graph of the n points, (ii) the local BA window
    over the m nodes x n_kf keyframes.  The default bench line is the parity mode (every point a node)."""
    import nrs
    import nrs_synth as S
    tp = S.make_tracking_problem(n, 3)
    cam = nrs.make_camera(tp["model"], tp["prm"])
    ctx = nrs.Context()
    fm = np.arange(n, dtype=np.int32)
    t0 = time.perf_counter()
    nodes = ctx.skin_select_nodes(tp["X_prev"], m, tp["status"] == 0)
    t_sel = time.perf_counter() - t0
    t0 = time.perf_counter()
    nodes = ctx.skin_select_nodes(tp["X_prev"], m, tp["status"] == 0)
    t_sel = min(t_sel, time.perf_counter() - t0)
    st = nrs.skinned_status(tp["status"], fm, nodes)
    g = nrs.RGraph(ctx, n, tp["graph"]["sigma"], tp["graph"]["stretch_th"])
    g.add_edges(tp["X_prev"], fm, fm)
    ms, r = [], None
    for rep in range(4):
        g.add_edges(tp["X_prev"], fm, fm)                           # the graph as it was before the frame
        tr = nrs.Trace(1024)
        t0 = time.perf_counter()
        r = ctx.track_deform_solve_rg(cam, g, tp["X_prev"], fm, st, tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], tr, 512)
        ms.append(1e3 * (time.perf_counter() - t0))
    out = dict(points=n, nodes=m, tracked_points=int((tp["status"] == 0).sum()), ms_select_nodes=1e3 * t_sel,
               ms_pose_and_deformation=float(np.median(ms[1:])), skinned_points=len(r["lost"]),
               lm_trials=len(tr.trials), pcg_iters=int(sum(x["inner"] for x in tr.trials)))
    # ---- the same frame in the EMBEDDED-DEFORMATION mode (N2 as SURVEY.md 8d words it; include/nrs.h nrs_track_deform_solve_embedded): the m
    # nodes carry the free variables, the observations of all other tracked points constrain them through <= 11 nodes each
    node = np.zeros(n, np.uint8)
    node[nodes] = 1
    ms2, r2, tr2 = [], None, None
    for rep in range(4):
        g.add_edges(tp["X_prev"], fm, fm)
        tr2 = nrs.Trace(1024)
        t0 = time.perf_counter()
        r2 = ctx.track_deform_solve_embedded(cam, g, tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], node, tp["pose_q"], tp["pose_t"], tp["scale"], tr2)
        ms2.append(1e3 * (time.perf_counter() - t0))
    out["embedded"] = dict(points=n, nodes=m, tracked_points=int((tp["status"] == 0).sum()), ms_pose_and_deformation=float(np.median(ms2[1:])),
                           frames_per_s_of_this_call=1e3 / float(np.median(ms2[1:])), lm_trials=len(tr2.trials),
                           linear_solver="nested-dissection Cholesky over %d node blocks + pose (k_nd_level / k_nd_back)" % m,
                           note="every tracked point's reprojection edge is in the problem (skinned to <= 11 nodes, normalised weights); "
                                "unknowns 6 + 3 x %d; held to oracle/embedded_oracle.py at 600 x 80, 1500 x 200, 900 x 120 KB8 (tests/test_gpu_embedded.py) "
                                "and at 5000 x 500 pinhole + KB8 by committed goldens (tests/test_gpu_embedded5k.py)" % m)
    g.close()
    # ---- N2b: BASELINE configs[1] AS WRITTEN -- the C2 window (5k points x 20 keyframes) with m nodes: the node copies carry the
    # vertices, every other observation is skinned to <= 11 node copies of its keyframe and constrains them and the pose
    # (include/nrs.h nrs_dba_*_embedded; PCG with the observations' blocks as hyper-edges, csrc/nrs_engine_skin.hpp)
    p = S.make_dba_problem("C2")
    flag, nb = S.embedded_problem(p, m)
    e = nrs.dba_build_edges_embedded(p["kf_points"], flag, nb)
    w = S.embedded_window(p, e)
    camw = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    ctx.dba_upload_embedded(camw, qt, w, e, p["scale"])
    kft_default = ctx.debug_kft_info()["on"]                      # (nrs_options.embedded_solver = 0: the library's cost model chose)
    rr = timed_steps(ctx, 10, 2, lambda: None)
    out["ba_window"] = dict(workload="C2 embedded: %d points x %d nodes x %d keyframes" % (p["n_points"], m, p["n_kf"]), node_copies=int(len(e["lm_obs"])),
                            skinned_observations=int(len(e["sk_obs"])), springs=int(len(e["sp_ij"])), dampers=int(len(e["dm_idx"])),
                            unknowns=int(6 * p["n_kf"] + 3 * len(e["lm_obs"])), value=rr["lm_iters"] / rr["dt"], unit="LM iters/s",
                            ms_per_step=1e3 * rr["dt"] / 10, lm_trials_per_step=rr["trials"] / 10, pcg_iters_per_step=rr["inner"] / 10,
                            linear_solver=("the default (nrs_options.embedded_solver = 0, cost model): exact keyframe-block factorisation on the matrix cores as the PCG's "
                                           "preconditioner (csrc/nrs_engine_kft.hpp), 1-2 PCG iterations per LM trial") if kft_default else
                                          "the default (nrs_options.embedded_solver = 0, cost model): block-Jacobi PCG, the skinned observations applied as hyper-edges",
                            note="every observation of the window is in the problem; held to oracle/embedded_oracle.py dba_solve_embedded at 300 x 40 x 4 .. "
                                 "600 x 80 x 6 (tests/test_gpu_embedded_ba.py) and at this size by the golden tests/golden/dba_C2_embedded%d_trace.npz; "
                                 "the mode has no reference counterpart beyond every-point-a-node (there it is the plain window, bit for bit)" % m)
    # ---- its roofline entry: the two launches of a PCG iteration (k_spmv_f_skin<8>: the regularisers' operator + the observations' pass;
    # k_pcg_update<true>: the observations' row pass + the vector update), HIP events on the context's stream (nrs_options.profile).
    # Algorithmic bytes per iteration: operator -- per skinned observation A_o (6) + B_o (18) doubles, its 11 (row, weight)
    # pairs and the 24-byte g_o it leaves = 348 B; per node row u, w, the linearisation point and the 32-byte reprojection factors = 104 B;
    # 12 B per spring incidence, 16 B per damper incidence.  Update -- per node row 6 vectors read, 5 written, M^-1 = 312 B; per (row, observation)
    # list entry weight + index + g_o = 36 B.  Both stay in the 256 MB Infinity Cache at this size: the fraction is of the HBM peak all the same.
    bctx = nrs.Context(embedded_solver=2)
    bctx.dba_upload_embedded(camw, qt, w, e, p["scale"])
    rb = timed_steps(bctx, 10, 2, lambda: None)
    bctx.close()
    out["ba_window"]["block_jacobi_pcg"] = dict(value=rb["lm_iters"] / rb["dt"], unit="LM iters/s", ms_per_step=1e3 * rb["dt"] / 10, pcg_iters_per_step=rb["inner"] / 10,
                                                note="nrs_options.embedded_solver = 2: k_spmv_f_skin / k_pcg_update<true>, two launches per iteration; the two roofline entries below are its kernels")
    pctx = nrs.Context(profile=1, embedded_solver=2)
    pctx.dba_upload_embedded(camw, qt, w, e, p["scale"])
    pctx.dba_optimize(2)
    pctx.reset_profile()
    pctx.dba_reset()
    pctx.dba_optimize(2)
    prof = pctx.profile()
    pctx.close()
    emb_traffic = {}                                              # HBM bytes per launch from the PMC passes of tools/profile_r06.sh (not this run)
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        emb_traffic = {k: v.get("bytes_per_launch") for k, v in json.load(open(tpath)).get("embedded_C2", {}).items()}
    n_obs, n_rows, n_ent = int(len(e["sk_obs"])), int(len(e["lm_obs"])), int((np.asarray(e["sk_node"]) >= 0).sum())
    op_b = 348 * n_obs + 104 * n_rows + 12 * 2 * len(e["sp_ij"]) + 16 * 4 * len(e["dm_idx"])
    up_b = 312 * n_rows + 36 * n_ent
    op_us = 1e3 * prof["spmv_ms"] / max(1, prof["spmv_launches"])
    up_us = 1e3 * prof["vec_ms"] / max(1, prof["vec_launches"])
    out["ba_window"]["roofline"] = dict(kernel="k_spmv_f_skin<8> (operator of the regularisers + the skinned observations' pass, one launch)", bound="hbm",
                                        achieved=op_b / (op_us * 1e-6) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=op_b / (op_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                        traffic=emb_traffic.get("k_spmv_f_skin"), avg_us=op_us, algorithmic_bytes=int(op_b), launches=int(prof["spmv_launches"]),
                                        regime="27 MB of operands per launch: Infinity-Cache resident, bound by the chains of dependent loads (latency), not by bytes")
    out["ba_window"]["roofline_update"] = dict(kernel="k_pcg_update<true> (the observations' row pass + the PCG vector update)", bound="hbm",
                                               achieved=up_b / (up_us * 1e-6) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=up_b / (up_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                               traffic=emb_traffic.get("k_pcg_update_skin"), avg_us=up_us, algorithmic_bytes=int(up_b), launches=int(prof["vec_launches"]),
                                               traffic_source="profiles/traffic.json embedded_C2 (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_r06.sh, 2 * FETCH + WRITE)")
    out["ba_window"]["pcg_iterations_per_lm_trial"] = rr["inner"] / max(1, rr["trials"])
    # ---- the same window on the exact keyframe-block factorisation forced (nrs_options.embedded_solver = 1)
    fctx = nrs.Context(embedded_solver=1)
    fctx.dba_upload_embedded(camw, qt, w, e, p["scale"])
    info = fctx.debug_kft_info()
    rf = timed_steps(fctx, 10, 2, lambda: None)
    fctx.close()
    nbk, K = info["nb"], info["K"]
    flops_trial = K * (nbk ** 3) * 2.0 * 64 ** 3                  # K inversions, nb steps of nb^2 rank-64 tile updates (full squares)
    tf = flops_trial * rf["trials"] / rf["dt"] / 1e12
    out["ba_window"]["keyframe_block_factorisation"] = dict(
        value=rf["lm_iters"] / rf["dt"], unit="LM iters/s", ms_per_step=1e3 * rf["dt"] / 10, pcg_iters_per_step=rf["inner"] / 10, lm_trials_per_step=rf["trials"] / 10,
        block_dimension=info["ld"], keyframe_blocks=K, factor_mib=info["mib"], gflop_per_trial=flops_trial / 1e9,
        roofline=dict(kernel="k_kft_step (one launch per 64-pivot sweep step: panel + trailing rank-64 update on v_mfma_f64_16x16x4)", bound="mfma", achieved=tf,
                      peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / FP64_MFMA_PEAK_TFLOPS, traffic=emb_traffic.get("k_kft_step"),
                      note="whole-trial rate (assembly, Schur updates, solves included): a CRITICAL-PATH computation -- %d dependent launches per trial, "
                           "each bounded by a 64-pivot sweep in one workgroup" % (((K + 1) // 2) * (nbk + 1) + nbk + 1)),
        note="exact solve per LM trial (what the reference's LinearSolverEigen does): block tridiagonal over the keyframes, csrc/nrs_engine_kft.hpp; "
             "1-2 PCG iterations per trial; faster than the PCG below ~450 nodes x 20 keyframes (profiles/r06_kft_crossover.txt)")
    out["ba_window_inputs"] = (p, e, w)                            # (the CPU leg of main() runs the same window; removed before printing)
    ctx.close()
    return out


def cpu_baseline_embedded(p, e, w, gpu_value):
    """cpu_baseline of the embedded C2 window: the C++ restatement of the embedded form (oracle/nrs_cpu.cpp nrs_cpu_dba_solve_embedded, held to
    oracle/embedded_oracle.py by tests/test_oracle_cpp_cpu.py), the whole optimize(5) with the block-Jacobi PCG on all host threads (the faster
    CPU algorithm), one thread next to it, and ONE trial of the reference's own linear solve (sparse Cholesky: 146 GFLOP per factorisation)."""
    import nrs_cpu as CPU
    CPU.build(native=True)
    lib = CPU.load(native=True)
    nt = CPU.max_threads(lib)

    def run(solver, threads, max_trials):
        t0 = time.perf_counter()
        r = CPU.dba_solve_embedded(p["model"], p["prm"], p["poses_q"], p["poses_t"], w["lm_xyz"], w["lm_kf"], w["lm_uv"], e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"],
                                   w["sk_kf"], w["sk_uv"], w["sk_xyz"], e["sk_node"], e["sk_omega"], p["scale"], 5, solver, 1e-10, threads, max_trials, lib)
        return time.perf_counter() - t0, r[5]
    # (memory-bound and small: more threads than memory channels make it slower -- the fastest of a short sweep on the first LM trial)
    cand = sorted({t for t in (4, 8, 16, 32, nt) if t <= nt})
    best = min(cand, key=lambda t: run(1, t, 1)[1]["t_solve"])
    dt, st = run(1, best, 0)
    rate = st["n_iters"] / (st["t_total"] - st["t_structure"])
    out = dict(value=rate, unit="LM iters/s", cores=best, cores_available=nt, kind="port",
               sample="the whole optimize(5) of the embedded C2 window (5000 points x 500 nodes x 20 keyframes), C++ restatement oracle/nrs_cpu.cpp, block-Jacobi PCG "
                      "to 1e-10 on %d threads: %d LM iterations, %d PCG iterations, %.1f s (+ %.1f s structure, once per window)"
                      % (best, st["n_iters"], st["n_pcg_iters"], st["t_total"] - st["t_structure"], st["t_structure"]),
               gpu_over_cpu=gpu_value / rate)
    dt1, st1 = run(1, 1, 0)
    out["one_core"] = dict(value=st1["n_iters"] / (st1["t_total"] - st1["t_structure"]), unit="LM iters/s", cores=1,
                           sample="the same optimize(5) on one thread (%d PCG iterations), %.1f s" % (st1["n_pcg_iters"], st1["t_total"] - st1["t_structure"]))
    dt0, st0 = run(0, 1, 1)
    out["sparse_cholesky_one_trial"] = dict(value=st0["n_trials"] / max(1e-9, st0["t_factor"] + st0["t_solve"] + st0["t_linearize"] + st0["t_errors"]), unit="LM trials/s", cores=1,
                                            gflop_per_factorisation=st0["chol_flops"] / 1e9, seconds_factorisation=st0["t_factor"],
                                            note="what the reference runs per LM trial (linear_solver_eigen.h:92-173), here the oracle's own AMD-ordered up-looking block Cholesky")
    return out


FP64_MFMA_PEAK_TFLOPS = 78.6        # dense fp64 matrix peak of MI355X (MI355X_MICROARCH.md: 32 flop / clk / SIMD x 1024 SIMDs x 2.4 GHz)


def direct_solver_leg(sizes=(1013, 4446)):
    """Roofline entry of the kernel pair that decides tracked fps: the nested-dissection multifrontal Cholesky of a2's system
    (k_nd_level: one launch per tree level, k_nd_back: one launch), timed with HIP events over 50 back-to-back factorise + solve
    sequences on a2-like block systems (nrs_synth.nd_block_system: kNN-11 couplings + the pose, the flat graph's structure) through
    the tap nrs_debug_nd_solve.  It is a CRITICAL-PATH kernel, not a throughput one: `frac` of the fp64 matrix peak is reported
    because the contract asks for it, the model that explains the time is levels x per-level latency."""
    import nrs
    import nrs_synth as S
    ctx = nrs.Context()
    out = {}
    for n in sizes:
        pos, last, pairs, Dn, Vp, bn = S.nd_block_system(n)
        ok, x, st, ms = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, 0.1, repeats=50)
        tf = st["flops"] / (ms * 1e-3) / 1e12
        out["%d_points" % n] = dict(ok=bool(ok), plan_flops=int(st["flops"]), fronts=int(st["fronts"]), levels=int(st["levels"]), workgroups=int(st["workgroups"]),
                                    largest_front=int(st["max_s"]), largest_boundary=int(st["max_b"]), us_per_factorise_and_solve=1e3 * ms,
                                    achieved=tf, frac=tf / FP64_MFMA_PEAK_TFLOPS,
                                    critical_path=dict(levels=int(st["levels"]), us_per_level_all_in=1e3 * ms / max(1, st["levels"]),
                                                       note="one launch per level (a level's latency = one workgroup's panel factorisation of <= 96 columns + its Schur tile) "
                                                            "+ one back-pass launch; the time is levels x latency, not flops / peak"))
    ctx.close()
    big = out["%d_points" % sizes[-1]]
    traffic = None                                             # HBM bytes per launch of the two kernels at 4446 points (own PMC passes: profiles/traffic.json)
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get("direct_solver_4446_points")
    return {"kernel": "k_nd_level<512> + k_nd_tile + k_nd_back (multifrontal Cholesky of a2's system on the nested-dissection plan, fronts on v_mfma_f64_16x16x4)",
            "bound": "mfma", "achieved": big["achieved"], "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": big["frac"], "traffic": traffic,
            "workload": "a2-like block system, %d points + pose, kNN-11 couplings" % sizes[-1], "regime": "latency (critical path): see critical_path", "sizes": out}


def hbm_regime_leg(device, workload="C4"):
    """The two roofline kernels on a window that does NOT fit the 256 MB Infinity Cache (C4: 50k points x 200 keyframes,
    ~14 GB resident): HIP events on the context's own stream around back-to-back full launches (nrs_options.profile), one
    optimize(2) -- a bounded extra leg (window generation + upload dominate: ~30 s)."""
    import nrs
    p, e, cam, qt = make_window(workload)
    c = nrs.Context(device=device, profile=1)
    c.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    c.dba_optimize(1)
    c.reset_profile()
    c.dba_reset()
    t0 = time.perf_counter()
    tr = nrs.Trace(64)
    c.dba_optimize(2, tr)
    dt = time.perf_counter() - t0
    prof = c.profile()
    st = c.dba_stats()
    c.close()
    n_lm, n_sp, n_dm = len(p["lm_kf"]), len(e["sp_ij"]), len(e["dm_idx"])
    lin_b, spmv_b = algorithmic_bytes(n_lm, n_sp, n_dm, unique_blocks(n_lm, e["sp_ij"], e["dm_idx"]))
    spmv_us = 1e3 * prof["spmv_ms"] / max(1, prof["spmv_launches"])
    lin_us = 1e3 * prof["linearize_ms"] / max(1, prof["linearize_launches"])
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")       # PMC passes (tools/profile_r0x.sh), not this run: lower bounds, profiles/README.md
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(workload, {})
    return {"workload": "%s: %d landmarks, %d springs, %d dampers, %.1f GB resident" % (workload, n_lm, n_sp, n_dm, st["device_bytes"] / 1e9),
            "traffic": {"k_spmv_f": traffic.get("k_spmv"), "linearize": traffic.get("linearize"), "source": "profiles/traffic.json (rocprofv3 --pmc, bytes per full launch)"},
            "operator": {"kernel": "k_spmv_f", "avg_us": spmv_us, "algorithmic_bytes": spmv_b, "achieved": spmv_b / (spmv_us * 1e-6) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": spmv_b / (spmv_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "launches": prof["spmv_launches"]},
            "linearize": {"kernel": "k_lin_plain", "avg_us": lin_us, "algorithmic_bytes": lin_b, "achieved": lin_b / (lin_us * 1e-6) / 1e9,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lin_b / (lin_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "launches": prof["linearize_launches"]},
            "note": "profiling context: launches serialised by events, LM trials %d in %.2f s" % (len(tr.trials), dt)}


def oneshot_leg(device, p, e, cam, qt, reps=5):
    """What a drop-in caller pays per LocalDeformableBundleAdjustment call (mapping.cc:57: a NEW window every keyframe): the
    one-shot nrs_dba_solve -- host buffers in and out, PCIe-inclusive, including the packing of the incidence streams --
    on the bench window and on a reference-sized window (5 keyframes, OPT:894), next to the resident reset + optimize
    that `value` times.  Also nrs_dba_build_edges (host only: OPT:927-1137's edge construction)."""
    import nrs
    import nrs_synth as S
    out = {}
    p5 = S.make_dba_problem(5000, 5, 1)
    for name, pp, ee in (("bench_window", p, e), ("reference_window_5kf", p5, None)):
        t0 = time.perf_counter()
        e2 = nrs.dba_build_edges(pp["kf_points"], pp["nbr"])
        t_edges = time.perf_counter() - t0
        ee = ee or e2
        cm = nrs.make_camera(pp["model"], pp["prm"])
        q = np.concatenate([pp["poses_q"], pp["poses_t"]], 1)
        c = nrs.Context(device=device)
        c.dba_solve(cm, q, pp["lm_xyz"], pp["lm_kf"], pp["lm_uv"], ee, pp["scale"], 5)         # first call: arena allocation, code objects
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            c.dba_solve(cm, q, pp["lm_xyz"], pp["lm_kf"], pp["lm_uv"], ee, pp["scale"], 5)
            ts.append(time.perf_counter() - t0)
        c.dba_solve_window(cm, q, pp["kf_points"], pp["lm_xyz"], pp["lm_uv"], pp["nbr"], pp["scale"], 5)
        tw = []
        for _ in range(reps):                                # the whole reference call: edge construction included (on the device)
            t0 = time.perf_counter()
            c.dba_solve_window(cm, q, pp["kf_points"], pp["lm_xyz"], pp["lm_uv"], pp["nbr"], pp["scale"], 5)
            tw.append(time.perf_counter() - t0)
        c.dba_upload(cm, q, pp["lm_xyz"], pp["lm_kf"], pp["lm_uv"], ee, pp["scale"])
        c.dba_optimize(5)
        tr_ = []
        for _ in range(reps):
            c.dba_reset()
            t0 = time.perf_counter()
            c.dba_optimize(5)
            tr_.append(time.perf_counter() - t0)
        c.close()
        out[name] = {"landmarks": int(len(pp["lm_kf"])), "keyframes": int(len(q)), "oneshot_ms": 1e3 * min(ts), "resident_optimize_ms": 1e3 * min(tr_),
                     "build_edges_ms": 1e3 * t_edges, "window_call_ms": 1e3 * min(tw),
                     "note": "oneshot_ms = nrs_dba_solve (edge lists given); window_call_ms = nrs_dba_solve_window (keyframes + neighbour lists in: edge construction included)"}
    return out


def triangulation_bench():
    """f2: every triangulation candidate of a frame in one call (21 buffered snapshots, host buffers in, points out),
    next to the NumPy restatement (oracle/triang_oracle.py, 1 core) on the first 10 candidates."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nrs
    import nrs_synth as S
    import triang_oracle as T
    tb = S.make_temporal_buffer(21, 7, baseline=0.3, spacing=22.0)
    ctx = nrs.Context()
    cam = nrs.make_camera(tb["model"], tb["prm"])
    st, _ = ctx.triangulate_batch(cam, tb, tb["cand"])
    t0 = time.perf_counter()
    for _ in range(5):
        st, _ = ctx.triangulate_batch(cam, tb, tb["cand"])
    gpu_ms = 1e3 * (time.perf_counter() - t0) / 5
    ctx.close()
    t0 = time.perf_counter()
    for c in tb["cand"][:10]:
        T.deformable_triangulation(tb, int(c), tb["model"], tb["prm"])
    cpu_ms = 1e3 * (time.perf_counter() - t0) / 10
    return dict(candidates=int(len(tb["cand"])), triangulated=int((st == 0).sum()), snapshots=21, keypoint_ids=int(tb["has_kp"].shape[1]),
                ms_per_batch=gpu_ms, us_per_candidate=1e3 * gpu_ms / max(1, len(tb["cand"])), cpu_oracle_ms_per_candidate=cpu_ms,
                cpu_kind="port (NumPy, 1 core)")


def shi_extract_bench(reps=20):
    """Shi-Tomasi extraction on a 640x480 frame holding 1500 keypoints (host image in,
    keypoints out: PCIe-inclusive), next to the oracle's per-cell NumPy form on this host (1 core)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nrs
    import nrs_synth as S
    import shi_oracle as SH
    sq = S.make_lk_sequence(10, 5)
    ctx = nrs.Context()
    ctx.shi_configure(5)
    held = ctx.shi_extract(sq["im0"])[0][:1500]
    ctx.shi_extract(sq["im1"], held)
    t0 = time.perf_counter()
    for i in range(reps):
        xy, _, n = ctx.shi_extract(sq["im1"] if i % 2 else sq["im0"], held)
    gpu_ms = 1e3 * (time.perf_counter() - t0) / reps
    ctx.close()
    ex = SH.ShiTomasi(5)
    ex.extract(sq["im0"], held)
    t0 = time.perf_counter()
    for i in range(3):
        ex.extract(sq["im1"] if i % 2 else sq["im0"], held)
    cpu_ms = 1e3 * (time.perf_counter() - t0) / 3
    return dict(image="640x480", held_keypoints=int(len(held)), new_keypoints=int(n), ms_per_call=gpu_ms,
                cpu_oracle_ms_per_call=cpu_ms, cpu_kind="port (NumPy closed form, 1 core)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C2", help="N = 1: the window `value` is measured on")
    ap.add_argument("--sharded-workload", default="C4", help="N > 1: the ONE window that is sharded over the ranks")
    ap.add_argument("--sharded-points", type=int, default=0, help="override the sharded window's map points (tests)")
    ap.add_argument("--sharded-kf", type=int, default=0, help="override the sharded window's keyframes (tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-regime", action="store_true", help="skip the bounded C4 leg (operator + lineariser in the HBM regime)")
    ap.add_argument("--no-sharded", action="store_true", help="N > 1: independent windows only")
    ap.add_argument("--sharded-timeout", type=float, default=900.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    import torch
    multi = world > 1 or bool(os.environ.get("NRS_BENCH_FORCE_DIST"))   # (the switch: plumbing check of the N > 1 path on a 1-GPU box)
    if multi:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl")
    import nrs
    import nrs_synth as S

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    n_points, n_kf, seed, model = S.CONFIGS[args.workload]
    p, e, cam, qt = make_window(args.workload, 1000 * rank)
    ctx = nrs.Context(device=local_rank)          # fails loudly without a HIP device
    ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    r = timed_steps(ctx, args.steps, args.warmup, barrier)
    dev = "cuda" if dist is not None else None
    dt, lm_iters_all = reduce_over_ranks(dist, r["dt"], r["lm_iters"], dev)
    arith = "fp64 state / normal equations / PCG, fp32 projection and projection Jacobian (as the reference)"
    n_lm, n_sp, n_dm = len(p["lm_kf"]), len(e["sp_ij"]), len(e["dm_idx"])

    def describe(name, npts, nkf, mdl):
        return "%s: %d map points x %d keyframes, %s, every point a graph node; optimize(5) per step" % (
            name, npts, nkf, "pinhole" if mdl == 0 else "KannalaBrandt8")

    out = None
    if rank == 0:
        out = {"metric": "deformable-BA LM iters/sec", "value": lm_iters_all / dt, "unit": "LM iters/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": describe(args.workload, n_points, n_kf, model), "arithmetic": arith,
                          "landmarks": n_lm, "springs": n_sp, "dampers": n_dm, "lm_trials_per_step": r["trials"] / args.steps,
                          "pcg_iters_per_step": r["inner"] / args.steps,
                          "parallelism": "independent BA window per GPU" if world > 1 else "1 GPU"}}
    if rank == 0:
        # which library ran: the in-tree build (gitignored, travels with the snapshot) and how __graft_entry__.build() last produced it
        import hashlib
        bm = None
        try:
            bm = json.load(open(os.path.join(ROOT, "nr-slam_amd", "build", "build_mode.json")))
        except Exception:
            pass
        with open(nrs.LIB_PATH, "rb") as fh:
            out["build"] = {"library": os.path.relpath(nrs.LIB_PATH, ROOT), "sha256_16": hashlib.sha256(fh.read()).hexdigest()[:16],
                            "last_build": bm or "no record: the library travelled prebuilt with the snapshot"}
    if not multi and rank == 0:
        # ---- the same steps with every LM trial solved to pcg_rtol (g2o's behaviour; `value` rejects hopeless trials early)
        xctx = nrs.Context(device=local_rank, exact_trials=1)
        xctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
        n_exact = max(2, min(25, args.steps // 4))
        rx = timed_steps(xctx, n_exact, 1, barrier)
        out["value_exact_trials"] = rx["lm_iters"] / rx["dt"]
        out["config"]["pcg_iters_per_step_exact_trials"] = rx["inner"] / n_exact
        # ---- roofline of the dominant kernels: HIP events on the context's own stream ----------
        pctx = nrs.Context(device=local_rank, profile=1)
        pctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
        pctx.dba_optimize(5)
        pctx.reset_profile()
        pctx.dba_reset()
        pctx.dba_optimize(5)
        prof = pctx.profile()
        pctx.close()
        nblk = unique_blocks(n_lm, e["sp_ij"], e["dm_idx"])
        lin_b, spmv_b = algorithmic_bytes(n_lm, n_sp, n_dm, nblk)
        spmv_us = 1e3 * prof["spmv_ms"] / max(1, prof["spmv_launches"])
        lin_us = 1e3 * prof["linearize_ms"] / max(1, prof["linearize_launches"])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")      # PMC pass, see profiles/README.md
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(args.workload, {})
        spmv_gbs = spmv_b / (spmv_us * 1e-6) / 1e9
        lin_gbs = lin_b / (lin_us * 1e-6) / 1e9
        tsrc = "profiles/traffic.json (rocprofv3 --pmc passes of %s, not this run; lower bound: profiles/README.md)" % (
            json.load(open(tpath)).get("source", "tools/profile_r03.sh") if os.path.exists(tpath) else "-")
        out["roofline"] = {"kernel": "k_spmv_f (PCG operator apply, dominant: see profiles/)", "bound": "hbm",
                           "achieved": spmv_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": spmv_gbs / HBM_PEAK_GBS,
                           "traffic": (traffic or {}).get("k_spmv"), "traffic_source": tsrc, "avg_us": spmv_us,
                           "algorithmic_bytes": spmv_b, "launches": prof["spmv_launches"],
                           "regime": "%s's working set (~150 MB at C2) sits in the 256 MB Infinity Cache: see roofline_hbm_regime for the HBM-resident window" % args.workload}
        out["roofline_linearize"] = {"kernel": "k_lin_plain<T> (residuals + Jacobians + Huber + per-incidence factors + row blocks, one fused pass; k_reg<T, true, true> specialised for plain BA windows)",
                                     "bound": "hbm", "achieved": lin_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": lin_gbs / HBM_PEAK_GBS, "traffic": (traffic or {}).get("linearize"), "traffic_source": tsrc,
                                     "avg_us": lin_us, "algorithmic_bytes": lin_b, "launches": prof["linearize_launches"]}
        if not args.no_hbm_regime:
            out["roofline_hbm_regime"] = hbm_regime_leg(local_rank)
        out["roofline_direct_solver"] = direct_solver_leg()
        out["oneshot"] = oneshot_leg(local_rank, p, e, cam, qt)
        # the reference's map graph connects every pair of map points (map.cc:148-166): that is the graph `tracked_fps` runs on,
        # resident on the device; the generator's kNN-16 flat graph (what round 1 measured) stays next to it
        out["tracked_fps"] = tracked_fps(dense_graph=True)
        out["tracked_fps"]["graph"] = "all pairs (4999 connections per point), device resident"
        keys = ("value", "unit", "points", "frames", "ms_per_frame_median", "ms_per_frame_p95", "keyframes", "frames_with_point_reuse", "points_reused",
                "ms_klt_track", "ms_pose_only", "ms_pose_and_deformation", "ms_point_reuse", "ms_keyframe_work", "lm_trials_per_frame", "pcg_iters_per_frame",
                "tracked_last_frame")
        tf = tracked_fps(dense_graph=True, direct_solve=2)           # the same frames with the PCG as linear solver (round 3's path)
        out["tracked_fps_pcg_solver"] = {k: tf[k] for k in keys}
        tf = tracked_fps(n_points=1150, dense_graph=True)            # the reference's own scale (C1: ~1k tracked points)
        out["tracked_fps_1k_points"] = {k: tf[k] for k in keys}
        tf = tracked_fps(dense_graph=False)
        out["tracked_fps_flat_knn16_graph"] = {k: tf[k] for k in keys}
        # the metric's "tracked fps, 5k pts x 500 graph nodes" end to end: the same frame loop with the pose-and-deformation solve in the
        # embedded-deformation mode (500 map points carry the vertices, every other tracked point is skinned to <= 11 of them; N2a)
        tf = tracked_fps(dense_graph=True, n_nodes=500)
        out["tracked_fps_5k_x_500"] = dict({k: tf[k] for k in keys}, nodes=500, mode="embedded deformation (nrs_track_deform_solve_embedded)",
                                           note="held to oracle/embedded_oracle.py at 600 x 80 .. 1500 x 200 (tests/test_gpu_embedded.py) and at 5000 x 500, pinhole + KB8 "
                                                "(tests/test_gpu_embedded5k.py, committed goldens); no reference counterpart beyond every-point-a-node")
        out["shi_extract"] = shi_extract_bench()
        out["graph_dense"] = rgraph_bench()
        out["triangulation"] = triangulation_bench()
        out["skinned"] = skinned_bench()
        emb_inputs = out["skinned"].pop("ba_window_inputs")
        out["value_5k_x_500"] = out["skinned"]["ba_window"]["value"]      # BASELINE.json's config as written (5k points x 500 nodes x 20 keyframes), LM iters/s
        if not args.no_cpu_baseline:
            out["skinned"]["ba_window"]["cpu_baseline"] = cpu_baseline_embedded(*emb_inputs, out["value_5k_x_500"])
            out["tracked_fps_5k_x_500"]["cpu_baseline"] = cpu_embedded_frame(out["tracked_fps_5k_x_500"]["value"])
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(p, e, ctx=ctx, ctx_exact=xctx)
            cb = out["cpu_baseline"]
            if cb.get("gpu_same_sample_exact_trials"):
                # like for like: the same window, every LM trial solved to 1e-10 on both sides (`value` / cpu is NOT this ratio:
                # `value` rejects hopeless trials early)
                out["speedup_exact_vs_cpu"] = cb["gpu_same_sample_exact_trials"] / cb["value"]
        xctx.close()
    ctx.close()

    if multi and not args.no_sharded:
        # ---- ONE window sharded over the ranks: this is `value` at N > 1 ------------------------------------
        import threading
        if rank == 0:
            out["independent_windows"] = {"value": out["value"], "unit": "LM iters/s", "ms_per_step": out["ms_per_step"],
                                          "workload": out["config"]["workload"], "note": "every rank its own window, no data-path collective"}
        done = threading.Event()

        def watchdog():
            if done.wait(args.sharded_timeout):
                return
            if rank == 0:                                     # the sharded section hangs: the independent-window figure stands
                out["sharded_error"] = "the sharded window did not finish within %.0f s" % args.sharded_timeout
                flush_c_stdio()
                print(json.dumps(out), flush=True)
            os._exit(3)
        threading.Thread(target=watchdog, daemon=True).start()
        sn, sk, _, smodel = S.CONFIGS[args.sharded_workload]
        sn, sk = args.sharded_points or sn, args.sharded_kf or sk
        win = make_window(args.sharded_workload, 0, sn, sk)          # the same window on every rank
        res, err = sharded_section(dist, rank, world, local_rank, args.steps, args.warmup, barrier, win, "cuda")
        if res is not None:
            sdt, _ = reduce_over_ranks(dist, res["dt"], 0.0, "cuda")
            single = None
            if rank == 0 and world > 1:
                # rank 0 alone on the same window (the other ranks wait at the barrier below)
                c1 = nrs.Context(device=local_rank)
                pw, ew, camw, qtw = win
                c1.dba_upload(camw, qtw, pw["lm_xyz"], pw["lm_kf"], pw["lm_uv"], ew, pw["scale"])
                n_single = max(2, min(10, args.steps // 4))
                r1 = timed_steps(c1, n_single, 1, lambda: torch.cuda.synchronize())
                c1.close()
                single = {"value": r1["lm_iters"] / r1["dt"], "unit": "LM iters/s", "ms_per_step": 1e3 * r1["dt"] / n_single}
            if rank == 0:
                pw, ew = win[0], win[1]
                out.update({"value": res["lm_iters"] / sdt, "ms_per_step": 1e3 * sdt / args.steps, "scaling": "strong"})
                out["config"] = {"workload": "ONE window sharded over %d ranks by keyframes -- " % world + describe(args.sharded_workload, sn, sk, smodel),
                                 "arithmetic": arith, "landmarks": len(pw["lm_kf"]), "springs": len(ew["sp_ij"]), "dampers": len(ew["dm_idx"]),
                                 "lm_trials_per_step": res["trials"] / args.steps, "pcg_iters_per_step": res["inner"] / args.steps,
                                 "parallelism": "keyframe ranges over %d GPUs; RCCL all-reduce of H_pp / b_p / chi2 per linearisation and of 3 + 6 K "
                                                "PCG sums per iteration, boundary-keyframe rows with rank +-1" % world,
                                 "rccl_ranks_seen": res["rccl_ranks_seen"], "upload_s_rank0": res["upload_s"],
                                 "keyframes_of_rank0": res["keyframes_of_rank"]}
                if single:
                    out["single_gpu_same_window"] = single
        elif rank == 0:
            out["sharded_error"] = err or "another rank failed"
        done.set()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        flush_c_stdio()                      # RCCL's start-up banner sits in the C stdio buffer: the JSON line comes last
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
