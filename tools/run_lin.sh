NRS_TIMING=1 timeout 300 python tools/small_frame_probe.py 1150 2>&1 | grep "a2 \|engine_create" | tail -34
