#!/bin/bash
# Round-6 profiling recipe (run on the GPU box through gpurun from the repo root):
#   1) rocprofv3 --kernel-trace --stats of the default bench command (no CPU legs)      -> per-kernel durations (C2 + the tracked-fps legs)
#   2) the embedded C2 window on both linear solvers (tools/kft_probe.py: block-Jacobi PCG and the keyframe-block factorisation forced),
#      kernel trace + stats, and the matrix-core counters of the factorisation's kernels in a pass of their own
#   3) the crossover sweep of the two solvers over nodes per keyframe (no tracer)
# Raw output goes to gpurun_out/r06 (scratch); the summaries are copied into profiles/ by hand (names r06_*).
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-regime > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/kft -o kft -- python $R/tools/kft_probe.py 5000 500 20 2 > $OUT/kft_probe_traced.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/kft300 -o kft -- python $R/tools/kft_probe.py 3000 300 20 2 > $OUT/kft300_probe_traced.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -f csv -d $OUT/pmc_mfma_kft -o p -- python $R/tools/kft_probe.py 5000 500 20 1 > $OUT/pmc_mfma_kft.log 2>&1
# HBM traffic of the embedded window's kernels (both solvers run in the probe): separate passes, 2 * FETCH_SIZE + WRITE_SIZE (profiles/README.md)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch_emb -o p -- python $R/tools/kft_probe.py 5000 500 20 1 > $OUT/pmc_fetch_emb.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write_emb -o p -- python $R/tools/kft_probe.py 5000 500 20 1 > $OUT/pmc_write_emb.log 2>&1
python $R/tools/summarize_traffic_emb.py $OUT/pmc_fetch_emb $OUT/pmc_write_emb > $OUT/traffic_embedded_C2.json 2> $OUT/traffic_embedded_C2.err
cd $R
for a in "1000 100 20" "2000 200 20" "3000 300 20" "4000 400 20" "5000 500 20" "5000 500 10" "2500 250 40"; do echo "== points nodes keyframes: $a"; python tools/kft_probe.py $a 3 2>&1 | grep embedded_solver | sed 's/; trials.*//'; done > $OUT/kft_crossover.txt
python bench.py --steps 200 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
# the merge back takes 64 MiB: the per-launch traces stay on the box (the statistics are what profiles/ keeps)
find $OUT -name '*kernel_trace.csv' -size +4M -delete
find $OUT -name '*.db' -delete
du -sh $OUT
