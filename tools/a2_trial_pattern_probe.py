"""The accept / reject pattern of a2's LM trials over a tracked sequence (tools/: measurement probe, GPU).  Per frame one line: per
optimisation round the string of trials ('A' accepted, 'r' rejected; '|' between LM iterations), then the histogram of rejection-run
lengths in front of an accepted trial over the sequence -- what a speculative second trial would save.
usage: python tools/a2_trial_pattern_probe.py [points=5000] [frames=13] [dense_graph=1]"""
import os
import sys
import collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd", "py"))
import bench  # noqa: E402
import nrs  # noqa: E402
import nrs_frame_loop as FL  # noqa: E402

n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 13
dense = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sq = bench.frame_sequence(n_points, frames + 1)
opts = dict(win=21, max_level=4, max_iters=10, epsilon=1e-4, min_eig=1e-4)
gb = FL.GpuBackend(nrs, sq["model"], sq["prm"], opts, dense_graph=bool(dense), cap_per_point=128)
loop = FL.FrameLoop(gb, lambda pc: FL.project_f32(sq["model"], sq["prm"], pc), sq["wh"], sq["scale"], sq["kp0"], sq["X0"],
                    sq["graph"], sq["pose_q"][0], sq["pose_t"][0], sq["images"][0])
runs = collections.Counter()
tot = rej = 0
for f in range(1, frames + 1):
    loop.track_image(sq["images"][f])
    s, last, run = "", None, 0
    for t in gb.last_trace.trials:
        key = (t["round"], t["iter"])
        if last is not None and key != last:
            s += " | " if key[0] != last[0] else "|"
        last = key
        s += "A" if t["accepted"] else "r"
        tot += 1
        if t["accepted"]:
            runs[run] += 1
            run = 0
        else:
            rej += 1
            run += 1
    if run:
        runs[("open", run)] += 1
    print("frame %2d: %s" % (f, s))
print("trials %d, rejected %d; rejection runs in front of an accepted trial: %s" % (tot, rej, sorted(runs.items(), key=str)))
gb.close()
