#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) result as a small markdown table: per-kernel calls / total / avg
duration, plus VGPR/LDS of the first dispatch.  Usage: tools/rocprof_summary.py results.db [> profiles/xyz.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute(
    "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, "
    "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
    "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc"))
total = sum(r[2] for r in rows) or 1.0
print("| kernel | calls | total us | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | scratch B | grid_x | wg_x |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    name = r[0].split("(")[0].replace("void ", "")
    print("| %s | %d | %.1f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s | %s | %s | %s |" % (
        name, r[1], r[2], r[3], r[4], r[5], 100 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
