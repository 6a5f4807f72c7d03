"""Debug: a device-built BA window with freshly allocated device memory poisoned (NRS_POISON=1): which call first reads memory
nobody wrote.  python tools/poison_probe.py [host]"""
import os, sys
os.environ["NRS_POISON"] = "1"
if len(sys.argv) > 1 and sys.argv[1] == "host":
    os.environ["NRS_HOST_PACK"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S
p = S.make_dba_problem(5000, 20, 1)
e = nrs.dba_build_edges(p["kf_points"], p["nbr"])
cam = nrs.make_camera(p["model"], p["prm"])
qt = np.concatenate([p["poses_q"], p["poses_t"]], 1)
c = nrs.Context()
print("upload", flush=True); c.dba_upload(cam, qt, p["lm_xyz"], p["lm_kf"], p["lm_uv"], e, p["scale"])
print("hash", flush=True); h = c.dba_pack_hash()
tr = nrs.Trace(64)
print("optimize", flush=True); c.dba_optimize(3, tr)
print([ (t["accepted"], t["chi_new"]) for t in tr.trials], flush=True)
print("download", flush=True); pq, xyz = c.dba_download()
print("residuals", flush=True); rr, rs, rd = c.dba_residuals()
print("finite:", np.isfinite(pq).all(), np.isfinite(xyz).all(), np.isfinite(rr).all(), np.isfinite(rs).all(), np.isfinite(rd).all(), flush=True)
c.close()
