"""Scratch probe: set-up time and resident bytes per rank of ONE sharded window (thread ranks on one GPU).
  python tools/shard_pack_probe.py [workload] [world]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n, k, seed, model = S.CONFIGS[name]
p = S.make_dba_problem(n, k, seed, model)
e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
cam = nrs.make_camera(p["model"], p["prm"])
qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
c = nrs.Context()
t0 = time.perf_counter(); c.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"]); t1 = time.perf_counter()
st = c.dba_stats(); c.close()
print({"workload": name, "ranks": 1, "upload_s": t1 - t0, "device_GB": st["device_bytes"] / 1e9, "spring_slots": st["spring_slots"]}, flush=True)
group = nrs.LocalGroup(world)
res = [None] * world
def rank_main(r):
    cc = nrs.Context(); cc.comm_init_local(group, r)
    t0 = time.perf_counter(); cc.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"]); t1 = time.perf_counter()
    s = cc.dba_stats(); tr = nrs.Trace(); t2 = time.perf_counter(); cc.dba_optimize(1, tr); t3 = time.perf_counter()
    res[r] = dict(rank=r, upload_s=t1 - t0, device_GB=s["device_bytes"] / 1e9, spring_slots=s["spring_slots"], packed_rows=s["packed_rows"], optimize1_s=t3 - t2)
    cc.close()
th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
[t.start() for t in th]; [t.join(600) for t in th]
for r in res: print(r)
