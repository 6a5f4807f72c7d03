timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/run_full_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/run_full_tests.log | tail -3
timeout 900 python tools/devpack_sweep.py 24 2>&1 | tail -3
timeout 900 python tools/sharded_sweep.py 24 2>&1 | tail -3
