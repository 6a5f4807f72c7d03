timeout 300 python tools/lin_probe.py C2 C3 C4 2>&1 | grep "workload"
timeout 1200 python -m pytest tests/test_gpu_devpack.py tests/test_gpu_dba.py tests/test_gpu_sharded.py tests/test_gpu_scale_large.py -x -q > gpurun_out/run_lin_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/run_lin_tests.log | tail -3
