NRS_PERSIST=1 timeout 120 python tools/small_frame_probe.py 600 1150 2>&1 | grep "^n \|rror"
NRS_PERSIST=1 timeout 600 python -m pytest tests/test_gpu_track.py tests/test_gpu_goldens.py -x -q > gpurun_out/run_lin_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/run_lin_tests.log | tail -3
