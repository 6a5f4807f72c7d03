#!/bin/bash
# Factor recomputation (Dev::rc, NRS_RC bit 0 springs / bit 1 dampers): parity tests with it on, then the lineariser / operator probe per mode
set -u
OUT=gpurun_out/rc_sweep; mkdir -p $OUT
for RC in ${RC_TEST_MODES:-3}; do
  NRS_RC=$RC timeout 900 python -m pytest tests/test_gpu_dba.py tests/test_gpu_devpack.py tests/test_gpu_scale.py tests/test_gpu_edge_cases.py -m gpu -x -q > $OUT/tests_rc$RC.log 2>&1
  tail -3 $OUT/tests_rc$RC.log
done
for RC in ${RC_MODES:-0 1 2 3}; do
  NRS_RC=$RC timeout 400 python tools/lin_probe.py ${RC_WORKLOADS:-C2 C3 C4} 2>&1 | grep workload | sed "s/^{/{\"rc\": $RC, /" | tee -a $OUT/lin_probe.jsonl
done
