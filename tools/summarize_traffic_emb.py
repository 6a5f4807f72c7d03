"""HBM bytes per launch of the embedded C2 window's kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: KB counters):
bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE), the gfx950 correction of MI355X_MICROARCH.md as profiles/traffic.json applies it.
usage: python tools/summarize_traffic_emb.py <fetch dir> <write dir>   -> json on stdout (merged into profiles/traffic.json by hand)"""
import collections
import csv
import glob
import json
import os
import sys

KERNELS = {"k_kft_step": "k_kft_step", "k_spmv_f_skin": "k_spmv_f_skin", "k_pcg_update<true>": "k_pcg_update_skin", "k_kft_gemv": "k_kft_gemv", "k_kft_gct": "k_kft_gct",
           "k_kft_tgt": "k_kft_tgt"}


def collect(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for k, name in KERNELS.items():
                if k in r["Kernel_Name"]:
                    acc[name].append(float(r["Counter_Value"]))
    return acc


fe, wr = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {}
for name in sorted(set(fe) | set(wr)):
    f = sum(fe[name]) / max(1, len(fe[name]))
    w = sum(wr[name]) / max(1, len(wr[name]))
    out[name] = {"FETCH_SIZE_KB_avg_per_launch": f, "WRITE_SIZE_KB_avg_per_launch": w, "launches": len(fe[name]), "bytes_per_launch": 1024.0 * (2 * f + w)}
print(json.dumps({"embedded_C2": out, "embedded_C2_source": "tools/profile_r06.sh (round 6): tools/kft_probe.py 5000 500 20 under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE"}, indent=1))
