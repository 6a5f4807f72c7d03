"""Seeded sweep, bit-exact: nrs_shi_extract against the oracle (oracle/shi_oracle.py) over random image sizes
(width >= height), NMS windows, masks and 3-call stateful sequences: keypoints, ids and all three buffers."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, nrs, nrs_synth as S, shi_oracle as SH
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = nrs.Context()
bad, t0, found = 0, time.time(), 0
base = [S.make_lk_sequence(10, 900 + k, wh=(480, 360), flow_px=5.0) for k in range(4)]
for seed in range(n):
    rng = np.random.default_rng(7000 + seed)
    h = int(rng.integers(5, 360)); w = int(rng.integers(h, 481))
    nms = int(rng.integers(0, 9))
    sq = base[seed % 4]
    y0, x0 = int(rng.integers(0, 360 - h + 1)), int(rng.integers(0, 480 - w + 1))
    ims = [np.ascontiguousarray(a[y0:y0 + h, x0:x0 + w]) for a in (sq["im0"], sq["im1"], sq["im0"][::-1])]
    if seed % 5 == 0:
        ims[1] = (rng.integers(0, 256, (h, w))).astype(np.uint8)          # white noise: saturates the scores
    mask = None if seed % 3 == 0 else (rng.uniform(size=(h, w)) > 0.2).astype(np.uint8)
    ctx.shi_configure(nms); ex = SH.ShiTomasi(nms)
    held = np.zeros((0, 2), np.float32)
    for im in ims:
        xy, ids, k = ctx.shi_extract(im, held, mask, capacity=w * h)
        oxy, oids = ex.extract(im, held, mask)
        sc, xg, yg = ctx.shi_buffers()
        ok = (k == len(oxy) and np.array_equal(xy, oxy) and np.array_equal(ids, oids) and np.array_equal(xg, ex.Xg)
              and np.array_equal(yg, ex.Yg) and np.array_equal(sc, ex.scores, equal_nan=True))
        if not ok:
            bad += 1
            print("MISMATCH seed %d size %dx%d nms %d" % (seed, w, h, nms))
            break
        found += k
        held = np.concatenate([held, xy[::2] + np.float32(rng.uniform(-0.49, 0.49))])
print("%d seeds, %d keypoints, %d mismatches, %.0f s" % (n, found, bad, time.time() - t0))
