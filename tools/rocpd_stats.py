"""Per-kernel summary (calls, total/avg/min/max ns, %) from a rocprofv3 rocpd sqlite database."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
scols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
q = "select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (namecol, kd, ks, namecol)
rows = list(db.execute(q))
tot = sum(r[2] for r in rows)
print("%-70s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "%"))
for r in rows:
    print("%-70s %8d %12d %10.0f %10d %10d %6.2f" % (r[0][:70], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
