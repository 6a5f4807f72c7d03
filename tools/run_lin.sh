set -x
python -m pytest tests/test_gpu_dba.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -5
for occ in 4 3; do NRS_LIN_OCC=$occ python tools/lin_probe.py C2 C3 C4 2>&1 | grep workload; done
NRS_NO_PLAIN=1 python tools/lin_probe.py C2 C4 2>&1 | grep workload
