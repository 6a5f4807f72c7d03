#!/usr/bin/env python3
"""bench.py -- deformable-BA LM iterations/sec on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic batch = one reference-shaped
LocalDeformableBundleAdjustment solve, optimize(5) (reference g2o_optimization.cc:1141-1143), on
BASELINE.json configs[1]: 5k map points x 20 keyframes (pinhole), every point a deformation-graph
node (parity mode).  Inputs are uploaded to HBM once, outside the timed
region; each timed step is {reset estimates (device-to-device), optimize(5)}.

N > 1 (launched by torch.distributed.run, one rank per GPU): `value` is measured with every rank
solving its own, independent BA window of the same size (weak scaling over independent windows, no
data-path collective -- DESIGN.md "Multi-GPU"); ranks are bracketed by a barrier and the slowest
rank's time is used.  After that the same line gets a "sharded" object: ONE window of C2's points
x 20*N keyframes split over the N ranks by keyframes (include/nrs.h "multi-GPU": RCCL all-reduce of
the pose blocks / PCG sums + boundary-keyframe exchange per iteration), run by one child process
per rank so that a failure of that path cannot take the benchmark line with it.

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


def cpu_baseline(seconds_budget=20.0, ctx=None):
    """The oracle (kind "port": NumPy/SciPy restatement of the reference, oracle/nrs_oracle.py)
    timed on this host, 1 core, on a bounded sample: the same generator at the reference's own
    window size (5 keyframes, g2o_optimization.cc:894) with 400 points -- C2 itself (275k unknowns,
    full sparse Cholesky per trial) does not finish in minutes on a CPU."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nrs
    import nrs_oracle as O
    import nrs_synth as S
    p = S.make_dba_problem(400, 5, 1)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    t0 = time.perf_counter()
    iters = 0
    runs = 0
    while True:
        _, _, _, nit = O.dba_solve(p["model"], p["prm"], p["poses_q"], p["poses_t"], p["lm_xyz"], p["lm_kf"],
                                   p["lm_uv"], e["sp_ij"], e["sp_d0"], e["dm_idx"], e["dm_w"], p["scale"], 5)
        iters += nit
        runs += 1
        if time.perf_counter() - t0 > seconds_budget or runs >= 5:
            break
    dt = time.perf_counter() - t0
    out = dict(value=iters / dt, unit="LM iters/s", cores=1, kind="port",
               sample="optimize(5) on 400 points x 5 keyframes (%d landmarks, %d springs, %d dampers), %d runs, %.1f s"
                      % (len(p["lm_kf"]), len(e["sp_ij"]), len(e["dm_idx"]), runs, dt))
    if ctx is not None:
        # the GPU path on the very same sample (resident, like `value`), for a like-for-like ratio
        cam = nrs.make_camera(p["model"], p["prm"])
        qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
        ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
        g_it, reps = 0, 20
        ctx.dba_optimize(5)
        t1 = time.perf_counter()
        for _ in range(reps):
            ctx.dba_reset()
            tr = nrs.Trace(64)
            ctx.dba_optimize(5, tr)
            g_it += tr.iterations
        out["gpu_same_sample"] = g_it / (time.perf_counter() - t1)
    out["tracked_fps"] = cpu_tracked_fps(ctx is not None)
    return out


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
    t = run(OracleBackend(sq["model"], sq["prm"], opts), [1])
    out = dict(value=1.0 / t[0], unit="frames/s", cores=1, kind="port",
               sample="1 tracked frame, %d map points, 640x480 (LK + pose-only + pose-and-deformation + point reuse), %.1f s" % (sq["n_points"], t[0]))
    if with_gpu:
        gb = FL.GpuBackend(nrs, sq["model"], sq["prm"], opts)
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


def tracked_fps(n_points=5000, frames=7):
    """Secondary figure of BASELINE.json's metric: tracked frames/s, end to end through the frame-loop
    harness (nr-slam_amd/py/nrs_frame_loop.py = reference tracking.cc:72-112 minus image decode and
    feature extraction) on a consistent synthetic 640x480 sequence with n_points map points: LK data
    association, motion-model seed, pose-only solve, pose-and-deformation solve (with its graph
    update), point reuse.  Every call takes host buffers (a frame arrives from the host): PCIe-inclusive."""
    import nrs
    import nrs_frame_loop as FL
    import nrs_synth as S
    sq = S.make_frame_sequence(n_points, frames + 1, 21)
    opts = dict(win=21, max_level=4, max_iters=10, epsilon=1e-4, min_eig=1e-4)
    gb = FL.GpuBackend(nrs, sq["model"], sq["prm"], opts)
    stage = {}

    def wrap(name):
        fn = getattr(gb, name)

        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            stage[name] = stage.get(name, 0.0) + time.perf_counter() - t0
            return r
        setattr(gb, name, w)
    for nme in ("klt_track", "pose_only", "track_deform", "reuse_track", "extract_features"):
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
    gb.close()
    nf = len(ts)
    return dict(value=nf / sum(ts), unit="frames/s", points=int(sq["n_points"]), frames=nf,
                tracked_last_frame=int(loop.log[-1]["n_tracked"]),
                ms_klt_track=1e3 * stage.get("klt_track", 0) / nf, ms_pose_only=1e3 * stage.get("pose_only", 0) / nf,
                ms_pose_and_deformation=1e3 * stage.get("track_deform", 0) / nf, ms_point_reuse=1e3 * stage.get("reuse_track", 0) / nf,
                ms_keyframe_extract=1e3 * stage.get("extract_features", 0) / nf, features_2d_last_frame=int(loop.log[-1]["n_2d"]),
                lm_trials_per_frame=trials / nf, pcg_iters_per_frame=inner / nf)


def flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def sharded_child(a):
    """One rank of the sharded window (no torch in this process: ctypes + librccl only).  Prints one
    JSON line with this rank's timing; the parent ranks reduce them."""
    import nrs
    import nrs_synth as S
    _, _, seed, model = S.CONFIGS[a.workload]
    n_points, n_kf = a.sh_points, a.sh_kf
    p = S.make_dba_problem(n_points, n_kf * a.sh_world, seed, model)       # the same window on every rank
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    ctx = nrs.Context(device=a.sh_device)
    # the RCCL id is made by the rank-0 CHILD and handed to the others through a file on this node: all
    # children load the same librccl (the parents' copy, PyTorch's, may be another version)
    if a.sh_rank == 0:
        with open(a.sh_uid_file + ".tmp", "wb") as f:
            f.write(nrs.comm_unique_id())
        os.replace(a.sh_uid_file + ".tmp", a.sh_uid_file)
    t_wait = time.perf_counter()
    while not os.path.exists(a.sh_uid_file):
        if time.perf_counter() - t_wait > 60:
            raise RuntimeError("no RCCL id from rank 0 after 60 s")
        time.sleep(0.01)
    ctx.comm_init_rccl(a.sh_world, a.sh_rank, open(a.sh_uid_file, "rb").read())
    t_up = time.perf_counter()
    ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
    t_up = time.perf_counter() - t_up
    for _ in range(a.warmup):
        ctx.dba_reset()
        ctx.dba_optimize(5)                  # collective: the ranks leave it together
    t0 = time.perf_counter()
    lm_iters = trials = inner = 0
    for _ in range(a.steps):
        ctx.dba_reset()
        tr = nrs.Trace(64)
        ctx.dba_optimize(5, tr)
        lm_iters += tr.iterations
        trials += tr.c.count
        inner += sum(t["inner"] for t in tr.trials)
    dt = time.perf_counter() - t0
    kb = nrs.shard_plan(n_kf * a.sh_world, p["lm_kf"], a.sh_world)
    ctx.close()
    flush_c_stdio()
    print(json.dumps(dict(dt=dt, lm_iters=lm_iters, trials=trials, inner=inner, upload_s=t_up, n_kf=n_kf * a.sh_world,
                          landmarks=len(p["lm_kf"]), springs=len(e["sp_ij"]), dampers=len(e["dm_idx"]),
                          keyframes_of_rank=[int(kb[a.sh_rank]), int(kb[a.sh_rank + 1])])), flush=True)


def run_sharded(args, dist, rank, world, local_rank, timeout_s=120, n_points=None, kf_per_rank=None):
    """All parent ranks: start this rank's child of the sharded window, collect its line.  Returns the
    "sharded" object on rank 0 (an {"error": ...} object if any rank's child failed or timed out).
    Window: n_points map points x kf_per_rank * world keyframes (default: the workload's own size per rank)."""
    import nrs_synth as S0
    n_points = n_points or S0.CONFIGS[args.workload][0]
    kf_per_rank = kf_per_rank or S0.CONFIGS[args.workload][1]
    import subprocess
    import torch
    import nrs
    import tempfile
    dev = "cuda" if torch.cuda.is_available() else "cpu"     # cpu: the gloo test of this bookkeeping
    # rendezvous file for the children's RCCL id (one node): rank 0 picks the name, everybody learns it
    name = torch.zeros(256, dtype=torch.uint8, device=dev)
    if rank == 0:
        fd, path = tempfile.mkstemp(prefix="nrs_rccl_id_", suffix=".bin")
        os.close(fd)
        os.unlink(path)                                        # the rank-0 child creates it (atomically)
        raw = path.encode()[:255]
        name[:len(raw)] = torch.tensor(list(raw), dtype=torch.uint8, device=dev)
    dist.broadcast(name, src=0)
    uid_file = bytes(name.cpu().tolist()).rstrip(b"\0").decode()
    cmd = [sys.executable, os.path.abspath(__file__), "--sharded-child", "--workload", args.workload,
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--sh-world", str(world), "--sh-rank", str(rank),
           "--sh-device", str(local_rank), "--sh-uid-file", uid_file,
           "--sh-points", str(n_points), "--sh-kf", str(kf_per_rank)]
    res, err = None, None
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        if r.returncode != 0:
            err = "rank %d: exit %d: %s" % (rank, r.returncode, r.stderr.strip()[-300:])
        else:
            res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])   # (RCCL prints a banner on stdout)
    except subprocess.TimeoutExpired:
        err = "rank %d: no result after %d s" % (rank, timeout_s)
    except Exception as ex:                                   # malformed output etc.
        err = "rank %d: %r" % (rank, ex)
    ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        for f in (uid_file, uid_file + ".tmp"):
            if os.path.exists(f):
                os.unlink(f)
    if ok.item() < 1.0:
        return {"error": err or "another rank's child failed"} if rank == 0 else None
    dt, _ = reduce_over_ranks(dist, res["dt"], 0.0, dev)
    if rank != 0:
        return None
    it_s = res["lm_iters"] / dt
    return {"workload": "ONE window: %d map points x %d keyframes (%d per rank), keyframes split over %d ranks"
                        % (n_points, res["n_kf"], kf_per_rank, world),
            "exchange": "RCCL: per linearisation all-reduce of H_pp/b_p/chi2 (27 K + 10 doubles) + boundary-keyframe rows; "
                        "per PCG iteration boundary rows of u + all-reduce of 3 + 6 K doubles",
            "lm_iters_per_s": it_s, "per_rank_windows_equivalent_iters_per_s": it_s * world, "ms_per_step": 1e3 * dt / args.steps,
            "lm_trials_per_step": res["trials"] / args.steps, "pcg_iters_per_step": res["inner"] / args.steps,
            "us_per_pcg_iter_incl_lm": 1e6 * dt / max(1, res["inner"]),
            "landmarks": res["landmarks"], "springs": res["springs"], "dampers": res["dampers"], "upload_s": res["upload_s"]}


def single_gpu_reference(ctx_device, workload, n_points, n_kf, steps, warmup):
    """The per-rank share of a sharded window as a plain single-GPU window (what N = 1 would run)."""
    import nrs
    import nrs_synth as S
    _, _, seed, model = S.CONFIGS[workload]
    p = S.make_dba_problem(n_points, n_kf, seed, model)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    ctx = nrs.Context(device=ctx_device)
    ctx.dba_upload(nrs.make_camera(p["model"], p["prm"]), np.concatenate([p["poses_q"], p["poses_t"]], 1), p["lm_xyz"], p["lm_kf"],
                   p["lm_uv"], e, p["scale"])
    for _ in range(warmup):
        ctx.dba_reset()
        ctx.dba_optimize(5)
    t0 = time.perf_counter()
    its = 0
    for _ in range(steps):
        ctx.dba_reset()
        tr = nrs.Trace(64)
        ctx.dba_optimize(5, tr)
        its += tr.iterations
    dt = time.perf_counter() - t0
    ctx.close()
    return dict(lm_iters_per_s=its / dt, ms_per_step=1e3 * dt / steps, landmarks=len(p["lm_kf"]))


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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sharded", action="store_true", help="N > 1: skip the sharded-window measurement")
    ap.add_argument("--sharded-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--sh-world", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--sh-rank", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--sh-device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--sh-uid-file", default="", help=argparse.SUPPRESS)
    ap.add_argument("--sh-points", type=int, default=5000, help=argparse.SUPPRESS)
    ap.add_argument("--sh-kf", type=int, default=20, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.sharded_child:
        return sharded_child(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    import torch
    if world > 1 or os.environ.get("NRS_BENCH_FORCE_DIST"):   # (the switch: plumbing check of the N > 1 path on a 1-GPU box)
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl")
    import nrs
    import nrs_synth as S

    n_points, n_kf, seed, model = S.CONFIGS[args.workload]
    p = S.make_dba_problem(n_points, n_kf, seed + 1000 * rank, model)
    e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
    cam = nrs.make_camera(p["model"], p["prm"])
    qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
    ctx = nrs.Context(device=local_rank)          # fails loudly without a HIP device
    ctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.dba_reset()
        ctx.dba_optimize(5)
    barrier()
    t0 = time.perf_counter()
    lm_iters = 0
    trials = 0
    inner = 0
    for _ in range(args.steps):
        ctx.dba_reset()
        tr = nrs.Trace(64)
        ctx.dba_optimize(5, tr)              # returns after the last host read-back: stream is idle
        lm_iters += tr.iterations
        trials += tr.c.count
        inner += sum(t["inner"] for t in tr.trials)
    barrier()
    dt = time.perf_counter() - t0
    dt, lm_iters_all = reduce_over_ranks(dist, dt, lm_iters, "cuda" if dist is not None else None)

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernels: HIP events on the context's own stream ----------
        pctx = nrs.Context(device=local_rank, profile=1)
        pctx.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
        pctx.dba_optimize(5)
        pctx.reset_profile()
        pctx.dba_reset()
        pctx.dba_optimize(5)
        prof = pctx.profile()
        pctx.close()
        n_lm, n_sp, n_dm = len(p["lm_kf"]), len(e["sp_ij"]), len(e["dm_idx"])
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
        out = {
            "metric": "deformable-BA LM iters/sec", "value": lm_iters_all / dt, "unit": "LM iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d map points x %d keyframes, %s, every point a graph node; "
                                   "optimize(5) per step" % (args.workload, n_points, n_kf,
                                                             "pinhole" if model == 0 else "KannalaBrandt8"),
                       "arithmetic": "fp64 state / normal equations / PCG, fp32 projection and projection Jacobian (as the reference)",
                       "landmarks": n_lm, "springs": n_sp, "dampers": n_dm,
                       "lm_trials_per_step": trials / args.steps, "pcg_iters_per_step": inner / args.steps,
                       "parallelism": "independent BA window per GPU" if world > 1 else "1 GPU"},
            "roofline": {"kernel": "k_spmv (PCG operator apply, dominant: see profiles/)", "bound": "hbm",
                         "achieved": spmv_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": spmv_gbs / HBM_PEAK_GBS,
                         "traffic": (traffic or {}).get("k_spmv"), "avg_us": spmv_us,
                         "algorithmic_bytes": spmv_b, "launches": prof["spmv_launches"]},
            "roofline_linearize": {"kernel": "k_reproj<true> + k_reg<true> (residual/Jacobian + assemble)", "bound": "hbm",
                                   "achieved": lin_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": lin_gbs / HBM_PEAK_GBS, "traffic": (traffic or {}).get("linearize"),
                                   "avg_us": lin_us, "algorithmic_bytes": lin_b, "launches": prof["linearize_launches"]},
        }
        if world == 1:
            out["tracked_fps"] = tracked_fps()
            out["shi_extract"] = shi_extract_bench()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ctx=ctx)
    ctx.close()
    if dist is not None and not args.no_sharded:
        dist.barrier()
        sh = run_sharded(args, dist, rank, world, local_rank)
        # the same with 2.5x larger shards (10k points x 25 keyframes per rank): where the exchange steps weigh less
        big_pts, big_kf = 10000, 25
        ok1 = torch.tensor([1.0 if (rank != 0 or (isinstance(sh, dict) and "error" not in sh)) else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(ok1, op=dist.ReduceOp.MIN)               # every rank learns whether the first window worked
        sh2 = run_sharded(args, dist, rank, world, local_rank, n_points=big_pts, kf_per_rank=big_kf) if ok1.item() > 0 else (
            {"error": "skipped: the first sharded window failed"} if rank == 0 else None)
        if rank == 0:
            out["sharded"] = sh
            if isinstance(sh2, dict) and "error" not in sh2:
                sh2["single_gpu_same_shard"] = single_gpu_reference(local_rank, args.workload, big_pts, big_kf, args.steps, args.warmup)
            out["sharded_large_shards"] = sh2
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        flush_c_stdio()                      # RCCL's start-up banner sits in the C stdio buffer: the JSON line comes last
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
