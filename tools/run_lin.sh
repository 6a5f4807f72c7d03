timeout 300 python tools/small_frame_probe.py 600 1150 1700 2500 2>&1 | grep "^n "
NRS_COARSE_MIN_TILES=0 timeout 300 python tools/small_frame_probe.py 1700 2>&1 | grep "^n "
timeout 1500 python -m pytest tests/test_gpu_track.py tests/test_gpu_goldens.py tests/test_gpu_frame_loop.py tests/test_gpu_early_reject.py tests/test_gpu_c1.py tests/test_gpu_skin.py tests/test_gpu_rgraph.py tests/test_gpu_host_mirror.py tests/test_gpu_edge_cases.py -x -q > gpurun_out/run_lin_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/run_lin_tests.log | tail -3
