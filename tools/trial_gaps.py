"""Host round trip between LM trials from a rocprofv3 kernel trace: the gap from the end of an evaluation (k_finalize) to the start of the next kernel, by successor.  python tools/trial_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
nm = [r['Kernel_Name'].split('(')[0].replace('void nrs::', '').replace('nrs::', '') for r in rows]
st = [int(r['Start_Timestamp']) for r in rows]; en = [int(r['End_Timestamp']) for r in rows]
g = collections.defaultdict(list)
for i in range(1, len(rows)):
    if nm[i-1].startswith('k_finalize'):
        g[(nm[i-1], nm[i][:24])].append((st[i] - en[i-1]) / 1e3)
for k, v in sorted(g.items(), key=lambda kv: -len(kv[1]))[:8]:
    v.sort(); print(k, "n", len(v), "median %.1f us" % v[len(v)//2], "p10 %.1f p90 %.1f" % (v[len(v)//10], v[9*len(v)//10]))
