"""Scratch probe: the skinned-mode leg of bench.py on its own (5k points x 500 nodes: frame + node BA window)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nr-slam_amd/py"))
import bench
print(json.dumps(bench.skinned_bench()))
