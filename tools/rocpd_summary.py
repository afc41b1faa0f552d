"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel launches, total / average / min / max duration.
usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name, name)).fetchall()
tot = sum(r[2] for r in rows)
print("%-46s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
for r in rows:
    print("%-46s %8d %14d %12.0f %12d %12d %6.2f%%" % (r[0][:46], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
