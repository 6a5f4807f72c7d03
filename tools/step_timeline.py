"""One optimize(5) of a resident BA window from a rocprofv3 kernel trace (tools/c2_probe.py): span, kernel-busy time, per-kernel totals and the idle gaps (host round trips).  python tools/step_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
nm = [r['Kernel_Name'].split('(')[0].replace('void nrs::', '').replace('nrs::', '') for r in rows]
st = [int(r['Start_Timestamp']) for r in rows]; en = [int(r['End_Timestamp']) for r in rows]
# last optimize: find last k_lin_plain runs; take window from 5th-last k_lin_plain to the end
lin = [i for i, n in enumerate(nm) if n.startswith('k_lin_plain')]
i0 = lin[-5]
t0 = st[i0]; busy = 0; gaps = []
for i in range(i0, len(rows)):
    busy += en[i] - st[i]
    if i > i0 and st[i] - en[i-1] > 3000: gaps.append(((st[i] - en[i-1]) / 1e3, nm[i-1][:28], nm[i][:28]))
span = en[-1] - t0
print("span %.1f us, kernel busy %.1f us, idle %.1f us" % (span / 1e3, busy / 1e3, (span - busy) / 1e3))
import collections
tot = collections.Counter(); cnt = collections.Counter()
for i in range(i0, len(rows)): tot[nm[i][:40]] += (en[i] - st[i]) / 1e3; cnt[nm[i][:40]] += 1
for k, v in tot.most_common(14): print("  %-42s %4d launches %8.1f us" % (k, cnt[k], v))
print("gaps > 3 us: %d, total %.1f us" % (len(gaps), sum(g[0] for g in gaps)))
for g in sorted(gaps, reverse=True)[:12]: print("   %.1f us  %s -> %s" % g)
