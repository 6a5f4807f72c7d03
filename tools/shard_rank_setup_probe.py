"""Set-up of ONE rank of a sharded window on its own (no contention for the host): where its time goes (NRS_TIMING=1 marks).
  python tools/shard_rank_setup_probe.py [workload] [world] [rank]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n, k, seed, model = S.CONFIGS[name]
p = S.make_dba_problem(n, k, seed, model)
e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
cam = nrs.make_camera(p["model"], p["prm"])
qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
import threading
def watchdog():                                   # (the upload ends in a hand-shake with the neighbouring ranks, which do not exist here: the marks are what is wanted)
    time.sleep(float(os.environ.get("PROBE_LIMIT_S", "60")))
    print("watchdog: the upload did not return (it waits for the other ranks after engine_create); the marks above stand", flush=True)
    os._exit(0)
threading.Thread(target=watchdog, daemon=True).start()
group = nrs.LocalGroup(world)
cc = nrs.Context(); cc.comm_init_local(group, rank)
t0 = time.perf_counter(); cc.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"]); t1 = time.perf_counter()
s = cc.dba_stats()
print(dict(rank=rank, world=world, upload_s=t1 - t0, device_GB=s["device_bytes"] / 1e9), flush=True)
os._exit(0)
