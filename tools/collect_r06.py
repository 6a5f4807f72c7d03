"""Copies the summaries of a tools/profile_r06.sh run (gpurun_out/r06) into profiles/ under their r06_* names, regenerates the matrix-core
counter summary and traffic.json's embedded_C2 entry.  usage: python tools/collect_r06.py"""
import collections, csv, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, "gpurun_out/r06"), os.path.join(R, "profiles")
for src, dst in (("bench.json", "r06_bench.json"), ("stats/bench_kernel_stats.csv", "r06_bench_kernel_stats.csv"), ("kft/kft_kernel_stats.csv", "r06_kft_kernel_stats_C2.csv"),
                 ("kft300/kft_kernel_stats.csv", "r06_kft_kernel_stats_300nodes.csv"), ("kft_probe_traced.txt", "r06_kft_probe_traced.txt"), ("kft_crossover.txt", "r06_kft_crossover.txt")):
    shutil.copy(os.path.join(O, src), os.path.join(P, dst))
agg, cnt = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter()
for r in csv.DictReader(open(os.path.join(O, "pmc_mfma_kft/p_counter_collection.csv"))):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CYCLES": cnt[k] += 1
rows = sorted(((k, cnt[k], int(v.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0)), int(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)), int(v.get("SQ_BUSY_CYCLES", 0))) for k, v in agg.items()), key=lambda r: -r[4])
with open(os.path.join(P, "r06_pmc_mfma_kft.csv"), "w") as o:
    o.write("kernel,launches,SQ_INSTS_VALU_MFMA_MOPS_F64,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,mfma_busy_over_busy\n")
    for r in rows[:24]: o.write("%s,%d,%d,%d,%d,%.4f\n" % (r + (r[3] / r[4] if r[4] else 0.0,)))
t = json.load(open(os.path.join(P, "traffic.json")))
n = json.load(open(os.path.join(O, "traffic_embedded_C2.json")))
t["embedded_C2"], t["embedded_C2_source"] = n["embedded_C2"], n["embedded_C2_source"]
json.dump(t, open(os.path.join(P, "traffic.json"), "w"), indent=1)
d = json.loads(open(os.path.join(O, "bench.json")).read().strip().splitlines()[-1])
print("value %.0f (%.3f ms), exact %.1f, roofline %.3f, lin %.3f, hbm regime %.3f / %.3f" % (d["value"], d["ms_per_step"], d["value_exact_trials"], d["roofline"]["frac"], d["roofline_linearize"]["frac"],
      d["roofline_hbm_regime"]["operator"]["frac"], d["roofline_hbm_regime"]["linearize"]["frac"]))
print("5k x 500: %.1f LM it/s (%.2f ms)" % (d["value_5k_x_500"], d["skinned"]["ba_window"]["ms_per_step"]))
for k in ("tracked_fps", "tracked_fps_1k_points", "tracked_fps_flat_knn16_graph", "tracked_fps_5k_x_500"):
    print("%-30s %.1f frames/s, median %.2f ms, p95 %.2f, a2 %.2f ms" % (k, d[k]["value"], d[k]["ms_per_frame_median"], d[k]["ms_per_frame_p95"], d[k]["ms_pose_and_deformation"]))
print("build", d["build"]["sha256_16"])
