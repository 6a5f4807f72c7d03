# per-launch durations of k_kft_step inside the inversions of one embedded C2 optimize (kernel trace): bash tools/kft_launch_trace.sh
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/kft_trace; rm -rf $O; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace -f csv -d $O -o t -- python $R/tools/kft_probe.py 5000 500 20 1 > $O/log.txt 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/kft_trace/**/t_kernel_trace.csv', recursive=True)[0]
st = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
runs, cur = [], []
for s, e, n in st:
    if 'k_kft_step' in n: cur.append((s, e))
    else:
        if len(cur) >= 10: runs.append(cur)
        cur = []
for run in runs[22:26]:
    print("launches %d: us" % len(run), [round((e - s) / 1e3, 1) for s, e in run], "total %.1f us" % ((run[-1][1] - run[0][0]) / 1e3))
import collections
d = collections.defaultdict(list)
for s, e, n in st: d[n.split('(')[0][-40:]].append((e - s) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:12]: print("%-42s n %5d  mean %.1f us  total %.2f ms" % (k, len(v), sum(v) / len(v), sum(v) / 1e3))
PY
rm -rf $O
