#!/bin/bash
# Round-3 profiling recipe (run on the GPU box through gpurun from the repo root):
#   1) rocprofv3 --kernel-trace --stats of the default bench command            -> per-kernel durations (C2)
#   2) separate --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass; SQ set on its own) on C2 and on C4
#      (C4 = the HBM regime: ~14 GB resident, far beyond the 256 MB Infinity Cache)
#   3) kernel-trace of one C4 optimize(1) for the full-launch durations the C4 roofline lines use
# Raw output goes to gpurun_out/r03 (scratch); tools/summarize_profile_r03.py copies the summaries into profiles/.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-regime > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
for W in C2 C4; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch_$W -o p -- python $R/tools/c4_probe.py $W 2 > $OUT/pmc_fetch_$W.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write_$W -o p -- python $R/tools/c4_probe.py $W 2 > $OUT/pmc_write_$W.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $OUT/pmc_sq_$W -o p -- python $R/tools/c4_probe.py $W 2 > $OUT/pmc_sq_$W.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -f csv -d $OUT/pmc_mfma_C2 -o p -- python $R/tools/c4_probe.py C2 2 > $OUT/pmc_mfma_C2.log 2>&1
# VALU / LDS instructions per wave of the lineariser, specialised kernel and the generic one it replaces (C4)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES -f csv -d $OUT/pmc_insts_C4 -o p -- python $R/tools/c4_probe.py C4 1 > $OUT/pmc_insts_C4.log 2>&1
NRS_NO_PLAIN=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES -f csv -d $OUT/pmc_insts_C4_generic -o p -- python $R/tools/c4_probe.py C4 1 > $OUT/pmc_insts_C4_generic.log 2>&1
cd $R
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 200 python tools/lin_probe.py C2 C3 C4 2>&1 | grep workload > $OUT/lin_probe.jsonl
find $OUT -name "*.csv" -size +20M -delete
ls -R $OUT | head -60
# launch-latency probe: kernel arguments in device memory or not (a2 = 2.2k single-launch PCG iterations per frame)
for v in 0 1; do HIP_FORCE_DEV_KERNARG=$v timeout 200 python tools/frame_probe.py 5000 2>&1 | grep "points" | tail -1 | sed "s/^/HIP_FORCE_DEV_KERNARG=$v /" >> $OUT/kernarg_probe.txt; done
