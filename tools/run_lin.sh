NRS_LIN_DBG=1 timeout 300 python tools/lin_probe.py C4 2>&1 | grep "workload\|phases"
timeout 300 python tools/lin_probe.py C2 C3 2>&1 | grep "workload"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/a2trace -- python $GRAFT_REPO_ROOT/tools/frame_probe.py 5000 2>&1 | grep "points\|^\["
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/a2trace/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last track_deform call: find the last k_pose... just take the final 40% of kernels by count after the last klt kernel
last_klt=max(i for i,r in enumerate(rows) if "klt" in r["Kernel_Name"] or "k_lk" in r["Kernel_Name"] or "pyr" in r["Kernel_Name"])
seg=rows[last_klt+1:]
span=(int(seg[-1]["End_Timestamp"])-int(seg[0]["Start_Timestamp"]))/1e3
by=collections.defaultdict(lambda:[0,0.0])
busy=0
for r in seg:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    by[r["Kernel_Name"][:60]][0]+=1; by[r["Kernel_Name"][:60]][1]+=d; busy+=d
print("segment kernels %d span %.1f us busy %.1f idle %.1f"%(len(seg),span,busy,span-busy))
for k,v in sorted(by.items(),key=lambda kv:-kv[1][1])[:16]: print("%9.1f us n=%5d mean %.2f %s"%(v[1],v[0],v[1]/v[0],k))
gaps=collections.defaultdict(list)
for a,b in zip(seg,seg[1:]):
    g=(int(b["Start_Timestamp"])-int(a["End_Timestamp"]))/1e3
    gaps[(a["Kernel_Name"][:30],b["Kernel_Name"][:30])].append(g)
for k,v in sorted(gaps.items(),key=lambda kv:-sum(kv[1]))[:14]: print("%9.1f us gap n=%5d mean %.2f max %.1f | %s -> %s"%(sum(v),len(v),sum(v)/len(v),max(v),k[0],k[1]))
PY
rm -rf gpurun_out/a2trace
