"""Times the direct solver's device kernels (factorise + solve, 2 x levels launches) on a2-like block systems of several sizes:
python tools/nd_kernel_probe.py [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("nr-slam_amd/py", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, nrs
from test_nd_cpu import block_system
ctx = nrs.Context()
for n in [int(a) for a in sys.argv[1:]] or [543, 1013, 2220, 4446]:
    pos, last, pairs, Dn, Vp, bn, A = block_system(n, 3, True, knn=11)
    ok, x, st, ms = ctx.debug_nd_solve(pos, last, pairs, Dn, Vp, bn, 0.1, repeats=50)
    print("n %d pairs/pt %.1f: %s -> %.1f us per factorise+solve (%.1f GFLOP/s)" % (n, (len(pairs) - 2 * n - 1) / n, st, 1e3 * ms, st["flops"] / ms / 1e6), flush=True)
