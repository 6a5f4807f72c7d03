"""gpurun_out/r05 (tools/profile_r05.sh) -> the summaries kept under profiles/: r05_bench_kernel_stats.csv, r05_bench.json,
r05_pmc_summary_{C2,C4}.csv (per-kernel averages of FETCH_SIZE / WRITE_SIZE / the SQ set, full launches only),
r05_kernel_durations_C4.csv, r05_lin_probe.jsonl, traffic.json (bytes per launch, read by bench.py)."""
import collections, csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "r05"), os.path.join(ROOT, "profiles")


def first(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None


f = first("stats/**/*kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(dst, "r05_bench_kernel_stats.csv"))
for name in ("bench.json", "lin_probe.jsonl", "small_frames.txt", "nd_crossover.txt", "nd_crossover_dense.txt", "tracked_fps_probe.txt", "nd_phases_1013.txt", "nd_phases_4446.txt",
             "embedded_phases.txt", "embedded_ba_probe.jsonl", "a1_100k.txt", "kernel_regs.txt", "a2_trial_timeline.txt", "shard_pack_probe.txt", "plan_probe.txt", "diag_probe.txt", "nt_probe.jsonl"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, "r05_" + name))


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
    with open(os.path.join(dst, "r05_pmc_summary_%s.csv" % W), "w") as fo:
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
# the direct solver's kernels (tap at 4446 points): bytes per launch, the same correction
nd = {}
for tag, cn in (("pmc_fetch_nd", "FETCH_SIZE"), ("pmc_write_nd", "WRITE_SIZE")):
    for k, cs in counters(tag).items():
        for key in ("k_nd_level", "k_nd_back"):
            if key in k and cn in cs:
                nd.setdefault(key, {})[cn + "_KB_avg_per_launch"] = sum(cs[cn]) / len(cs[cn])
                nd[key]["launches"] = len(cs[cn])
for key, v in nd.items():
    v["bytes_per_launch"] = 2 * 1024 * v.get("FETCH_SIZE_KB_avg_per_launch", 0.0) + 1024 * v.get("WRITE_SIZE_KB_avg_per_launch", 0.0)
if nd:
    traffic["direct_solver_4446_points"] = nd
if len(traffic) > 1:                                          # (the C2 / C4 counter passes ran: PROFILE_C4=1)
    traffic["source"] = "tools/profile_r05.sh (round 5), PROFILE_C4=1"
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
# full-launch durations at C4 from the kernel trace of the FETCH pass
f = first("pmc_fetch_C4/**/*kernel_trace.csv")
if f:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    with open(os.path.join(dst, "r05_kernel_durations_C4.csv"), "w") as fo:
        fo.write("kernel,launches,avg_us,max_us,total_ms\n")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            fo.write('"%s",%d,%.1f,%.1f,%.2f\n' % (k, len(v), sum(v) / len(v), max(v), sum(v) / 1e3))
print(open(os.path.join(dst, "traffic.json")).read())
for W in ("C2", "C4"):
    p = os.path.join(dst, "r05_pmc_summary_%s.csv" % W)
    if os.path.exists(p):
        print(open(p).read()[:3000])

# ---- direct solver (N1): kernel stats of a2 on small frames and of the tap; matrix-core / SQ counters of its kernels
for tag, name in (("nd_a2", "r05_nd_a2_kernel_stats.csv"), ("nd_tap", "r05_nd_tap_kernel_stats.csv"), ("emb_ba", "r05_embedded_ba_kernel_stats.csv")):
    f = first(tag + "/**/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, name))
rows = collections.defaultdict(dict)
for tag in ("pmc_mfma_nd", "pmc_sq_nd"):
    for k, cs in counters(tag).items():
        for cname, vals in cs.items():
            rows[k][cname] = (len(vals), sum(vals))
if rows:
    cols = sorted({c for k in rows for c in rows[k]})
    with open(os.path.join(dst, "r05_pmc_nd.csv"), "w") as fo:
        fo.write("kernel,launches," + ",".join(c + "_sum" for c in cols) + "\n")
        for k in sorted(rows):
            if "nd_" not in k:
                continue
            fo.write('"%s",%d,' % (k, max(v[0] for v in rows[k].values())) + ",".join("%.0f" % rows[k].get(c, (0, 0.0))[1] for c in cols) + "\n")
