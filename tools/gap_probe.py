"""Scratch: idle gaps between consecutive kernels of the last optimize(5) in a rocprofv3 kernel trace CSV.
  python tools/gap_probe.py <kernel_trace.csv>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last optimize: take the last 400 kernels
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -260:]
gaps = collections.defaultdict(list)
busy = 0
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    busy += (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
    gaps[(a["Kernel_Name"][:40], b["Kernel_Name"][:40])].append(g)
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print("kernels %d span %.1f us busy %.1f us idle %.1f us" % (len(rows), span, busy, span - busy))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%8.1f us total  n=%3d  mean %.2f  | %s -> %s" % (sum(v), len(v), sum(v) / len(v), k[0], k[1]))
