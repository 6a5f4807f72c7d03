set -x
NRS_DFORM=1 python tools/lin_probe.py C2 C4 2>&1 | grep "workload"
