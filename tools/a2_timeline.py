"""Kernel timeline of one batch of speculative a2 trials from a rocprofv3 kernel trace (csv): the launches of every stream between the
linearisation in front of a run of rejections and the one behind it.  usage: python tools/a2_timeline.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("nrs::", "").replace("void ", ""), r["Stream_Id"],
                 int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
main = max(collections.Counter(r[3] for r in rows).items(), key=lambda x: x[1])[0]
shadow = [i for i, r in enumerate(rows) if r[3] != main and "k_nd_level" in r[2] and r[4] >= 200]   # (the large engine of a frame, not its stage-2 engine)
if not shadow:
    sys.exit("no launches outside the main stream: nothing speculative in this trace")
# the batch in the middle of the trace with the most streams
best, best_n = None, 0
for i0 in shadow[len(shadow) // 3::50]:
    j = i0
    while j > 0 and not ("k_finalize<true>" in rows[j][2] and rows[j][3] == main):
        j -= 1
    k = i0
    while k < len(rows) - 1 and not ("k_finalize<true>" in rows[k][2] and rows[k][3] == main):
        k += 1
    n = len(set(r[3] for r in rows[j:k]))
    if n > best_n:
        best, best_n = (j, k), n
j, k = best
t0 = rows[j][0]
names = {main: "main"}
for r in rows[j:k + 1]:
    names.setdefault(r[3], "shadow %d" % (len(names) - 1))
print("start us   dur us   stream     kernel                          workgroups")
for r in rows[j:k + 1]:
    print("%8.1f %8.1f   %-9s  %-30s %d" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, names[r[3]], r[2][:30], r[4]))
