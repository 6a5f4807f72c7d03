set -x
python -m pytest tests/test_gpu_dba.py tests/test_gpu_sharded.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -3
for occ in 3 4; do NRS_LIN_DBG=1 NRS_LIN_OCC=$occ python tools/lin_probe.py C4 2>&1 | grep "workload\|phases"; done
NRS_LIN_DBG=1 python tools/lin_probe.py C2 C3 2>&1 | grep "workload\|phases"
