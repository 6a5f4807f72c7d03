python -m pytest tests/test_gpu_devpack.py -x -q 2>&1 | tail -15
NRS_TIMING=1 python tools/oneshot_probe.py 2>&1 | grep "one-shot" | tail -30
python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8
