timeout 300 python tools/small_frame_probe.py 600 1150 2>&1 | grep -v "^HIP\|^ROCm\|^Host\|^Librccl"
NRS_NO_WG1=1 timeout 300 python tools/small_frame_probe.py 600 1150 2>&1 | grep -v "^HIP\|^ROCm\|^Host\|^Librccl"
timeout 900 python -m pytest tests/test_gpu_track.py tests/test_gpu_goldens.py -x -q > gpurun_out/run_lin_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/run_lin_tests.log | tail -3
