"""Turns the raw rocprofv3 output of tools/profile_r01.sh (gpurun_out/r01/) into the summaries kept
under profiles/: kernel stats csv (copied), PMC summary csv, traffic.json (bytes per launch)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r01")
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
rows = {}
for name, f in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(src, f, "bench_counter_collection.csv"))):
        if r["Counter_Name"] == name:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    # the operator kernel also runs as a convergence-detecting launch that exits before touching the
    # records: only full applications (counter >= half of the largest) enter its average
    for k in list(agg):
        if "k_spmv" in k:
            top = max(agg[k])
            agg[k] = [x for x in agg[k] if x >= 0.5 * top]
    rows[name] = {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}
with open(os.path.join(dst, tag + "_pmc_summary.csv"), "w") as fo:
    fo.write("kernel,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg\n")
    for k in sorted(rows["FETCH_SIZE"], key=lambda k: -rows["FETCH_SIZE"][k][1] * rows["FETCH_SIZE"][k][0]):
        w = rows["WRITE_SIZE"].get(k, (0, 0.0))
        fo.write('"%s",%d,%.1f,%.1f\n' % (k, rows["FETCH_SIZE"][k][0], rows["FETCH_SIZE"][k][1], w[1]))


def kb(d, key):
    v = [x[1] for k, x in d.items() if key in k]
    return v[0] if v else 0.0


f, w = rows["FETCH_SIZE"], rows["WRITE_SIZE"]
upd_f, upd_w = kb(f, "k_pcg_update"), kb(w, "k_pcg_update")
spmv = 2 * kb(f, "k_spmv") * 1024 + kb(w, "k_spmv") * 1024
lin = 2 * (kb(f, "true, true>") + kb(f, "k_reproj<true>")) * 1024 + (kb(w, "true, true>") + kb(w, "k_reproj<true>")) * 1024
json.dump({"C2": {"k_spmv": spmv, "linearize": lin,
                  "note": "bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KB counters x1024); x2 on FETCH_SIZE = gfx950 "
                          "correction of MI355X_MICROARCH.md (HBM section), calibrated on k_pcg_update whose known traffic "
                          "is 192 B/row read + 120 B/row written (counters: %.0f KB fetched, %.0f KB written); %s_pmc_summary.csv"
                          % (upd_f, upd_w, tag)}},
          open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, tag + "_pmc_summary.csv")).read())
# full-launch duration of the operator kernel from the kernel trace of the bench run
tr = os.path.join(src, "stats", "bench_kernel_trace.csv")
if os.path.exists(tr):
    d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(tr)) if "k_spmv" in r["Kernel_Name"]]
    if d:
        full = [x for x in d if x >= 0.5 * max(d)]
        txt = ("k_spmv_f in the bench run: %d launches, %d full applications averaging %.2f us, %d convergence-detecting "
               "early exits averaging %.2f us\n" % (len(d), len(full), sum(full) / len(full) / 1e3, len(d) - len(full),
                                                  (sum(d) - sum(full)) / max(1, len(d) - len(full)) / 1e3))
        open(os.path.join(dst, tag + "_operator_launches.txt"), "w").write(txt)
        print(txt)
print(open(os.path.join(dst, "traffic.json")).read())
print(open(os.path.join(dst, tag + "_bench_kernel_stats.csv")).read()[:1800])
