timeout 900 python -m pytest tests/test_gpu_devpack.py -x -q > gpurun_out/run_lin_tests.log 2>&1
grep -E "passed|failed|error|assert" gpurun_out/run_lin_tests.log | tail -5
