import sys, os, pickle
sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/nr-slam_amd/py")
import numpy as np, scipy.sparse as sp
import nrs_oracle as O, nrs_synth as S
n = 2500
tp = S.make_tracking_problem(n, 3)
caps = []
orig = O.solve_spd
def cap(A, b, x_prev):
    caps.append((sp.csr_matrix(A).copy(), np.array(b)))
    return orig(A, b, x_prev)
O.solve_spd = cap
import inspect
print(inspect.signature(O.track_deform_solve))
fm = np.arange(n, dtype=np.int32)
try:
    r = O.track_deform_solve(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"], solver=cap)
except TypeError as e:
    print("retry", e)
    r = O.track_deform_solve(tp["model"], tp["prm"], tp["graph"], tp["X_prev"], fm, tp["status"], tp["uv"], tp["X_prev"], tp["pose_q"], tp["pose_t"], tp["scale"])
print(len(caps), caps[0][0].shape)
pickle.dump(dict(caps=caps[:12], X=tp["X_prev"], status=tp["status"]), open("/tmp/proto/caps.pkl", "wb"))
