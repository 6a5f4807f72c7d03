#!/bin/bash
# Round-5 profiling recipe (run on the GPU box through gpurun from the repo root):
#   1) rocprofv3 --kernel-trace --stats of the default bench command                 -> per-kernel durations (C2 + the tracked-fps legs)
#   2) the direct solver (N1): kernel trace of a2 on 543 / 1013-point frames and of the solver tap, and -- in their own passes, no
#      other trace domain -- the matrix-core counters of its kernels (SQ_INSTS_VALU_MFMA_MOPS_F64, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES)
#   3) optional (PROFILE_C4=1): FETCH_SIZE / WRITE_SIZE / SQ passes on C2 and C4 for the lineariser and the operator
# Raw output goes to gpurun_out/r05 (scratch); tools/summarize_profile_r05.py copies the summaries into profiles/.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r05
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-regime > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/nd_a2 -o a2 -- python $R/tools/small_frame_probe.py 600 1150 > $OUT/nd_a2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/nd_tap -o tap -- python $R/tools/nd_kernel_probe.py 543 1013 2220 > $OUT/nd_tap.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -f csv -d $OUT/pmc_mfma_nd -o p -- python $R/tools/nd_kernel_probe.py 1013 > $OUT/pmc_mfma_nd.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $OUT/pmc_sq_nd -o p -- python $R/tools/nd_kernel_probe.py 1013 > $OUT/pmc_sq_nd.log 2>&1
if [ "${PROFILE_C4:-0}" = "1" ]; then
for W in C2 C4; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch_$W -o p -- python $R/tools/c4_probe.py $W 2 > $OUT/pmc_fetch_$W.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write_$W -o p -- python $R/tools/c4_probe.py $W 2 > $OUT/pmc_write_$W.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $OUT/pmc_sq_$W -o p -- python $R/tools/c4_probe.py $W 2 > $OUT/pmc_sq_$W.log 2>&1
done
fi
cd $R
NRS_TIMING=1 timeout 200 python tools/small_frame_probe.py 600 1150 2500 5000 2>&1 | grep -E "^n |direct solve: [0-9p]" > $OUT/small_frames.txt
timeout 300 python tools/nd_crossover.py > $OUT/nd_crossover.txt 2>&1
timeout 600 python tools/nd_crossover.py --dense 600 1150 2500 3500 5000 > $OUT/nd_crossover_dense.txt 2>&1
timeout 300 python tools/tracked_fps_probe.py 5000 7 > $OUT/tracked_fps_probe.txt 2>&1
NRS_ND_DBG=1 timeout 100 python tools/nd_kernel_probe.py 1013 > $OUT/nd_phases_1013.txt 2>&1
# the embedded-deformation frame (5k points x 500 nodes) and the 500-node / 4500-lost-point frame: host phases of the last call of each, then the probe's line
NRS_TIMING=1 timeout 200 python tools/skinned_probe.py > $OUT/skinned_raw.txt 2>&1; (grep -E "\] a2 |set-up thread|waited" $OUT/skinned_raw.txt | tail -24; tail -1 $OUT/skinned_raw.txt) > $OUT/embedded_phases.txt
# N2b: the embedded C2 window (kernel trace + the probe's line), the a1 forms at 90k points, the recompute experiment's sweep, kernel registers
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/emb_ba -o emb -- python $R/tools/embedded_ba_probe.py C2 500 > $OUT/embedded_ba_probe.jsonl 2> $OUT/emb_ba.log
timeout 200 python tools/embedded_ba_probe.py C2 500 > $OUT/embedded_ba_probe.jsonl 2>/dev/null
timeout 100 python -m pytest tests/test_gpu_pose_only.py -m gpu -q -s -k 100k 2>&1 | grep "a1 at" > $OUT/a1_100k.txt
python tools/kernel_regs.py > $OUT/kernel_regs.txt 2>&1
# direct solver: FETCH / WRITE counters of its kernels (own passes), phase clocks at 4446 points
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch_nd -o p -- python $R/tools/nd_kernel_probe.py 4446 > $OUT/pmc_fetch_nd.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write_nd -o p -- python $R/tools/nd_kernel_probe.py 4446 > $OUT/pmc_write_nd.log 2>&1
cd $R
NRS_ND_DBG=1 timeout 100 python tools/nd_kernel_probe.py 4446 > $OUT/nd_phases_4446.txt 2>&1
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 200 python tools/lin_probe.py C2 C3 C4 2>&1 | grep workload > $OUT/lin_probe.jsonl
# round 5 additions: one LM trial of a2 on the all-pairs frame as a kernel timeline, the sharded window's per-rank footprint (thread ranks),
# the symbolic phase and the diagonal-block micro-benchmarks, non-temporal streams on / off
cd /tmp
timeout 400 rocprofv3 --kernel-trace -f csv -d $OUT/dense_prof -o d -- python $R/tools/tracked_fps_probe.py 5000 3 > $OUT/dense_prof.log 2>&1
cd $R
python tools/a2_trial_timeline.py $(find $OUT/dense_prof -name "*kernel_trace.csv" | head -1) 60 > $OUT/a2_trial_timeline.txt 2>&1
timeout 600 python tools/shard_pack_probe.py C4 8 2>/dev/null | grep -v "^\[" > $OUT/shard_pack_probe.txt
(nproc; for a in "4221 20" "4221 10" "1013 12"; do tools/micro/bin/plan_probe $a; done) > $OUT/plan_probe.txt 2>&1
tools/micro/bin/diag_probe > $OUT/diag_probe.txt 2>&1
for NT in 0 1; do NRS_NT=$NT timeout 400 python tools/lin_probe.py C3 C4 2>&1 | grep workload | sed "s/^{/{\"nt\": $NT, /"; done > $OUT/nt_probe.jsonl
find $OUT -name "*.csv" -size +20M -delete
ls -R $OUT | head -60
