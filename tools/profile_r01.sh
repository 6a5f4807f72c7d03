#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun from the repo root):
#   1) rocprofv3 --kernel-trace --stats of the default bench command  -> per-kernel durations
#   2) two separate --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share a pass: TCC has 4 slots)
# Raw output goes to gpurun_out/ (scratch); summaries are copied into profiles/ by hand.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r01
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o bench -- python $R/tools/c2_probe.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o bench -- python $R/tools/c2_probe.py > $OUT/pmc_write.log 2>&1
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
ls -R $OUT | head -40
