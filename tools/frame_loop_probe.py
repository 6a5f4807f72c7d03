"""End-to-end tracked frames/s of the frame-loop harness on a synthetic sequence (host generator
excluded; LK + pose-only + pose-and-deformation + point reuse + keyframe cadence per frame)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import numpy as np, nrs, nrs_synth as S, nrs_frame_loop as FL
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sq = S.make_frame_sequence(n, frames, 21)
opts = dict(win=21, max_level=4, max_iters=10, epsilon=1e-4, min_eig=1e-4)
gb = FL.GpuBackend(nrs, sq["model"], sq["prm"], opts)
proj = lambda pc: FL.project_f32(sq["model"], sq["prm"], pc)
loop = FL.FrameLoop(gb, proj, sq["wh"], sq["scale"], sq["kp0"], sq["X0"], sq["graph"], sq["pose_q"][0], sq["pose_t"][0], sq["images"][0])
ts = []
for f in range(1, frames):
    t0 = time.perf_counter(); ok = loop.track_image(sq["images"][f]); ts.append(time.perf_counter() - t0)
    L = loop.log[-1]
    tr = gb.last_trace.trials
    print("frame %d: %.1f ms, tracked %d/%d, lost %d, reused %d, keyframe %d; a2: %d trials, %d PCG iterations" % (f, 1e3 * ts[-1], L["n_tracked"], sq["n_points"], len(L["lost"]), L["reused"], L["keyframe"], len(tr), sum(t["inner"] for t in tr)))
print("points %d: median %.1f ms per frame = %.1f frames/s (harness in Python: includes its per-point loops)" % (sq["n_points"], 1e3 * np.median(ts), 1.0 / np.median(ts)))
# per-stage timing of the last frames: wrap the backend calls
import collections
acc = collections.defaultdict(float)
def wrap(name):
    fn = getattr(gb, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[name] += time.perf_counter() - t0; return r
    setattr(gb, name, w)
for nme in ("klt_track", "pose_only", "track_deform", "reuse_track", "klt_set_reference", "klt_get_templates", "klt_insert_template"):
    wrap(nme)
sq2 = sq
loop2 = FL.FrameLoop(gb, proj, sq["wh"], sq["scale"], sq["kp0"], sq["X0"], sq["graph"], sq["pose_q"][0], sq["pose_t"][0], sq["images"][0])
acc.clear()
t0 = time.perf_counter()
for f in range(1, frames):
    loop2.track_image(sq["images"][f])
tot = time.perf_counter() - t0
print("stage totals over %d frames (ms per frame): " % (frames - 1) + ", ".join("%s %.1f" % (k, 1e3 * v / (frames - 1)) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])) + ", harness itself %.1f" % (1e3 * (tot - sum(acc.values())) / (frames - 1)))
