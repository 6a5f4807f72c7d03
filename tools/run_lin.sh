timeout 600 python -m pytest tests/test_gpu_devpack.py -x -q 2>&1 | tail -5
NRS_LIN_DBG=1 timeout 300 python tools/lin_probe.py C4 2>&1 | grep "workload\|phases"
timeout 300 python tools/lin_probe.py C2 C3 2>&1 | grep "workload"
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8
