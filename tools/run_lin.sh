NRS_TIMING=1 python -m pytest tests/test_gpu_devpack.py -x -q 2>&1 | grep -v "engine_create\|a2 \|coarse level" | tail -60
