"""Golden for the C1 stand-in (BASELINE configs[0]: Hamlyn seq01, first 50 frames, ~1k tracked points).

The dataset is not in the repository and the reference cannot be built here, so the stand-in is the
synthetic sequence at C1's shape: Hamlyn intrinsics (data/hamlyn_01/settings.yaml:7-10), 640x480, ~1k map points,
50 frames, a keyframe every 5 frames (10 keyframes), the map's graph at the reference's all-pairs density.  This
script runs the ORACLE-driven frame loop (oracle/frame_loop_backend.py: NumPy restatements of LK, a1, a2, graph,
Shi-Tomasi behind the shared harness nr-slam_amd/py/nrs_frame_loop.py) once, in the build container, and stores
what tests/test_gpu_c1.py holds the GPU-driven loop to, frame by frame: pose, every landmark's status and position,
lost-id set, reused count, keyframe flag, extracted keypoints.  The oracle loop is too slow for the GPU box's
test budget (tens of seconds per frame), which is why the golden is committed.

    python tests/golden/make_c1_golden.py [n_frames]      ->  tests/golden/c1_standin_1000x50.npz
"""
import functools
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "nr-slam_amd", "py"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import nrs_frame_loop as FL
import nrs_synth as S
from frame_loop_backend import OracleBackend

N_POINTS, N_FRAMES, SEED, KF_EVERY = 1000, 50, 21, 5
OPTS = dict(win=21, max_level=4, max_iters=10, epsilon=1e-4, min_eig=1e-4)      # SLAM/system.cc:77-84


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else N_FRAMES
    sq = S.make_frame_sequence(N_POINTS, n_frames, SEED, S.PINHOLE)
    proj = functools.partial(FL.project_f32, sq["model"], sq["prm"])
    # the run takes ~40 minutes: the loop object is checkpointed after every frame (NRS_C1_CKPT=path) so that an interrupted
    # run resumes; the state after frame f is a deterministic function of the sequence, so a resumed run writes the same file
    ckpt = os.environ.get("NRS_C1_CKPT")
    loop, f0 = None, 1
    if ckpt and os.path.exists(ckpt):
        with open(ckpt, "rb") as fh:
            loop, f0 = pickle.load(fh)
        print("resuming after frame", f0 - 1, flush=True)
    if loop is None:
        loop = FL.FrameLoop(OracleBackend(sq["model"], sq["prm"], OPTS, dense_graph=True), proj, sq["wh"], sq["scale"], sq["kp0"],
                            sq["X0"], sq["graph"], sq["pose_q"][0], sq["pose_t"][0], sq["images"][0], images_to_insert_keyframe=KF_EVERY)
    t0 = time.time()
    for f in range(f0, n_frames):
        assert loop.track_image(sq["images"][f])
        L = loop.log[-1]
        print("frame %d: tracked %d, lost %d, reused %d, keyframe %d  (%.0f s)" % (f, L["n_tracked"], len(L["lost"]), L["reused"], L["keyframe"], time.time() - t0), flush=True)
        if ckpt:
            with open(ckpt + ".tmp", "wb") as fh:
                pickle.dump((loop, f + 1), fh)
            os.replace(ckpt + ".tmp", ckpt)
    log = loop.log
    out = dict(n_points=N_POINTS, n_frames=n_frames, seed=SEED, kf_every=KF_EVERY,
               pose_q=np.stack([L["pose_q"] for L in log]), pose_t=np.stack([L["pose_t"] for L in log]),
               status=np.stack([L["status_by_map"] for L in log]).astype(np.int8),
               pos=np.stack([L["pos_by_map"] for L in log]).astype(np.float32),
               reused=np.array([L["reused"] for L in log], np.int32), keyframe=np.array([L["keyframe"] for L in log], np.int8),
               n_tracked=np.array([L["n_tracked"] for L in log], np.int32), n_2d=np.array([L["n_2d"] for L in log], np.int32))
    lost = [np.array(sorted(L["lost"]), np.int32) for L in log]
    out["lost_ptr"] = np.cumsum([0] + [len(x) for x in lost]).astype(np.int32)
    out["lost_ids"] = np.concatenate(lost) if lost else np.zeros(0, np.int32)
    kp = [np.asarray(L["kp_2d"], np.float32).reshape(-1, 2) for L in log]
    out["kp_ptr"] = np.cumsum([0] + [len(x) for x in kp]).astype(np.int32)
    out["kp_2d"] = np.concatenate(kp) if kp else np.zeros((0, 2), np.float32)
    dst = os.path.join(ROOT, "tests", "golden", "c1_standin_%dx%d.npz" % (N_POINTS, n_frames))
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
