"""Kernel timeline of ONE LM trial of a2's direct solver from a rocprofv3 kernel trace (csv): python tools/a2_trial_timeline.py <kernel_trace.csv> [min k_nd_back us]
Prints the launches between two consecutive back passes of the large system (start, duration, grid), then the sums per kernel."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mn = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 60e3
nm = [r['Kernel_Name'].split('(')[0].replace('void nrs::', '').replace('nrs::', '') for r in rows]
st = [int(r['Start_Timestamp']) for r in rows]; en = [int(r['End_Timestamp']) for r in rows]
idx = [i for i, n in enumerate(nm) if n.startswith('k_nd_back') and en[i] - st[i] > mn]
i0, i1 = idx[-4], idx[-3]
t0 = en[i0]; tot = {}
for i in range(i0 + 1, i1 + 1):
    print("%-34s start %8.1f dur %7.1f grid %7s" % (nm[i][:34], (st[i] - t0) / 1e3, (en[i] - st[i]) / 1e3, rows[i]['Grid_Size_X']))
    tot[nm[i]] = tot.get(nm[i], 0) + (en[i] - st[i]) / 1e3
print("trial span %.1f us; per kernel: %s" % ((en[i1] - t0) / 1e3, {k: round(v, 1) for k, v in tot.items()}))
