"""Per-kernel averages of rocprofv3 --pmc counter_collection csv files: python tools/pmc_summary.py <dir>..."""
import collections, csv, glob, os, sys
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==", f)
        for k in sorted(agg, key=lambda k: -sum(sum(v) for v in agg[k].values())):
            if not any(s in k for s in ("k_reg", "k_spmv", "k_pcg_update", "k_chi", "k_reproj", "k_trial", "k_apply")):
                continue
            print(k[:70], {c: (len(v), round(sum(v) / len(v), 1)) for c, v in sorted(agg[k].items())})
