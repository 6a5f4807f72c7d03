"""N > 1 path on CPU: two gloo ranks run bench.py's cross-rank reduction (elapsed = MAX over ranks,
units = SUM over ranks) and the per-rank problem seeding the multi-GPU bench uses (one independent
BA window per rank, DESIGN.md "Multi-GPU")."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd", "py"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import nrs_synth as S
    n_points, n_kf, seed, model = 120, 3, 1, 0
    p = S.make_dba_problem(n_points, n_kf, seed + 1000 * rank, model)      # same seeding rule as bench.main
    dt, units = bench.reduce_over_ranks(dist, 1.0 + rank, 10 * (rank + 1))
    dist.barrier()
    q.put((rank, dt, units, float(p["lm_xyz"].sum()), len(p["lm_kf"])))
    dist.destroy_process_group()


def test_two_rank_reduction_and_seeding():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 2.0          # MAX of (1.0, 2.0)
    assert res[0][2] == res[1][2] == 30.0         # SUM of (10, 20)
    assert res[0][3] != res[1][3]                 # every rank solves its own window


def _worker_sharded(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd", "py"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import nrs
    win = bench.make_window("C2", 0, 120, 4)                  # the same small window on both ranks
    kb = nrs.shard_plan(4, win[0]["lm_kf"], world)
    res, err = bench.sharded_section(dist, rank, world, rank, 1, 0, dist.barrier, win, "cpu")
    dist.barrier()
    q.put((rank, res, err, kb.tolist(), float(win[0]["lm_xyz"].sum())))
    dist.destroy_process_group()


def test_sharded_window_bookkeeping_without_gpu():
    """bench.sharded_section (what `value` comes from at N > 1) on two gloo ranks of a box without a GPU: the RCCL id
    (or the fact that there is none) is broadcast, both ranks build the SAME window and the same keyframe plan, the
    upload fails on every rank (no HIP device: there is no CPU fallback), every rank learns that through the MIN
    all-reduce and gets an error text instead of a hang or a crash."""
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] is None and res[1][1] is None
    assert isinstance(res[0][2], str) and isinstance(res[1][2], str)
    assert res[0][3] == res[1][3] == [0, 2, 4]                 # the plan: two keyframes per rank
    assert res[0][4] == res[1][4]                              # one window, not one per rank


def test_single_process_passthrough():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.reduce_over_ranks(None, 0.25, 7) == (0.25, 7.0)
