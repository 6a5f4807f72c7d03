"""gpurun_out/r03 (tools/profile_r03.sh) -> the summaries kept under profiles/: r03_bench_kernel_stats.csv, r03_bench.json,
r03_pmc_summary_{C2,C4}.csv (per-kernel averages of FETCH_SIZE / WRITE_SIZE / the SQ set, full launches only),
r03_kernel_durations_C4.csv, r03_lin_probe.jsonl, traffic.json (bytes per launch, read by bench.py)."""
import collections, csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "r03"), os.path.join(ROOT, "profiles")


def first(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None


f = first("stats/**/*kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(dst, "r03_bench_kernel_stats.csv"))
for name in ("bench.json", "lin_probe.jsonl"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, "r03_" + name))


def counters(tag):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def full_launch_avg(vals):
    top = max(vals)
    v = [x for x in vals if x >= 0.5 * top]            # k_spmv_f also runs as a convergence-detecting early exit; class-1 tile launches are small
    return len(v), sum(v) / len(v)


traffic = {}
for W in ("C2", "C4"):
    rows = collections.defaultdict(dict)
    for tag in ("pmc_fetch_", "pmc_write_", "pmc_sq_"):
        for k, cs in counters(tag + W).items():
            for cname, vals in cs.items():
                rows[k][cname] = full_launch_avg(vals)
    if not rows:
        continue
    cols = ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
            "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
    with open(os.path.join(dst, "r03_pmc_summary_%s.csv" % W), "w") as fo:
        fo.write("kernel,full_launches," + ",".join(c + ("_KB" if "SIZE" in c else "") for c in cols) + "\n")
        for k in sorted(rows, key=lambda k: -rows[k].get("FETCH_SIZE", (0, 0))[1]):
            if not k.startswith(("void nrs", "nrs::")):
                continue
            n = max(v[0] for v in rows[k].values())
            fo.write('"%s",%d,' % (k, n) + ",".join("%.1f" % rows[k].get(c, (0, 0.0))[1] for c in cols) + "\n")

    def kb(key, c):
        v = [rows[k][c][1] for k in rows if key in k and c in rows[k]]
        return max(v) if v else 0.0
    traffic[W] = {"k_spmv": 2 * kb("k_spmv_f", "FETCH_SIZE") * 1024 + kb("k_spmv_f", "WRITE_SIZE") * 1024,
                  "linearize": 2 * max(kb("k_lin_plain", "FETCH_SIZE"), kb("k_reg<2, true, true", "FETCH_SIZE")) * 1024 + max(kb("k_lin_plain", "WRITE_SIZE"), kb("k_reg<2, true, true", "WRITE_SIZE")) * 1024,
                  "calibration_k_pcg_update": {"FETCH_SIZE_KB": kb("k_pcg_update", "FETCH_SIZE"), "WRITE_SIZE_KB": kb("k_pcg_update", "WRITE_SIZE")}}
traffic["note"] = ("bytes per FULL launch = 2*FETCH_SIZE + WRITE_SIZE (KB counters x 1024); the x2 on FETCH_SIZE is the gfx950 correction of "
                   "MI355X_MICROARCH.md (HBM section), which holds for 16-byte-per-lane streaming reads (k_pcg_update: known 192 B/row read, "
                   "120 B/row written -- see calibration_k_pcg_update); the 4- and 8-byte-per-lane streams of k_spmv_f / k_reg are under-counted "
                   "further (their known stream bytes exceed the corrected counters), so these figures are lower bounds")
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
# full-launch durations at C4 from the kernel trace of the FETCH pass
f = first("pmc_fetch_C4/**/*kernel_trace.csv")
if f:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    with open(os.path.join(dst, "r03_kernel_durations_C4.csv"), "w") as fo:
        fo.write("kernel,launches,avg_us,max_us,total_ms\n")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            fo.write('"%s",%d,%.1f,%.1f,%.2f\n' % (k, len(v), sum(v) / len(v), max(v), sum(v) / 1e3))
mf = counters("pmc_mfma_C2")
with open(os.path.join(dst, "r03_mfma_counters.txt"), "w") as fo:
    tot = collections.defaultdict(float)
    for k, cs in mf.items():
        for cname, vals in cs.items():
            tot[cname] += sum(vals)
    fo.write("sum over all kernels of one C2 optimize(2) (rocprofv3 --pmc, tools/profile_r03.sh): %s\n" % dict(tot))
print(open(os.path.join(dst, "traffic.json")).read())
for W in ("C2", "C4"):
    p = os.path.join(dst, "r03_pmc_summary_%s.csv" % W)
    if os.path.exists(p):
        print(open(p).read()[:3000])

# VALU / LDS instructions per wave of the lineariser (full launches): specialised kernel vs the generic one (NRS_NO_PLAIN=1)
with open(os.path.join(dst, "r03_lineariser_instructions_C4.txt"), "w") as fo:
    for tag, label in (("pmc_insts_C4", "k_lin_plain (this round)"), ("pmc_insts_C4_generic", "k_reg<2,true,true> (round 2 kernel, NRS_NO_PLAIN=1)")):
        for k, cs in counters(tag).items():
            if "k_lin_plain" not in k and "k_reg<2, true, true" not in k:
                continue
            v = {c: full_launch_avg(vals)[1] for c, vals in cs.items()}
            if v.get("SQ_WAVES"):
                fo.write("%s: %s\n  per launch: SQ_INSTS_VALU %.4g, SQ_INSTS_LDS %.4g, SQ_WAVES %.4g -> VALU instructions per wave %.0f (x4 waves = per tile %.0f), LDS instructions per wave %.0f\n"
                         % (label, k, v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_LDS", 0), v["SQ_WAVES"], v.get("SQ_INSTS_VALU", 0) / v["SQ_WAVES"],
                            4 * v.get("SQ_INSTS_VALU", 0) / v["SQ_WAVES"], v.get("SQ_INSTS_LDS", 0) / v["SQ_WAVES"]))
