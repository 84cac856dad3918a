#!/usr/bin/env python
"""Per-kernel average of every PMC counter found in a set of rocprofv3 rocpd databases.
Usage: tools/pmc_summary.py gpurun_out/<tag> [kernel-substring ...]"""
import glob
import sqlite3
import sys

root = sys.argv[1]
filt = sys.argv[2:] or ["k_loglik"]
for path in sorted(glob.glob(root + "/*/*_results.db")):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" not in tabs:
        rows = cur.execute("select name, count(*), avg(duration)/1e3 from kernels group by name").fetchall()
        for n, c, d in rows:
            if any(f in n for f in filt):
                print("%-14s %-40s calls=%d avg_us=%.1f" % (path.split("/")[-2], n.split("(")[0][-40:], c, d))
        continue
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    q = ("select kernel_name, counter_name, avg(v), count(*) from (select dispatch_id, kernel_name, counter_name, sum(value) as v "
         "from counters_collection group by dispatch_id, kernel_name, counter_name) group by kernel_name, counter_name")
    for n, cn, v, c in cur.execute(q):
        if any(f in n for f in filt):
            print("%-14s %-40s %-24s %.4g  (n=%d)" % (path.split("/")[-2], n.split("(")[0][-40:], cn, v, c))
