#!/bin/bash
# one measurement cycle of the embedded BA window (N2b): kernel stats under rocprofv3, the probe's LM rate, the embedded parity tests
# usage (GPU box): bash tools/emb_ba_cycle.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o emb -- python $R/tools/embedded_ba_probe.py C2 500 > $O/probe_prof.jsonl 2> $O/prof.log
head -8 $O/prof/emb_kernel_stats.csv | cut -c1-150
timeout 200 python $R/tools/embedded_ba_probe.py C2 500 2>/dev/null | tee $O/probe.jsonl | cut -c1-400
cd $R; python -m pytest tests/test_gpu_embedded_ba.py tests/test_gpu_embedded.py tests/test_gpu_embedded5k.py -x -q 2>&1 | tail -3
