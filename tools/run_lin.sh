timeout 1200 python -m pytest tests/test_gpu_track.py tests/test_gpu_goldens.py tests/test_gpu_frame_loop.py tests/test_gpu_early_reject.py tests/test_gpu_c1.py tests/test_gpu_skin.py tests/test_gpu_dba.py -x -q > gpurun_out/run_lin_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/run_lin_tests.log | tail -3
