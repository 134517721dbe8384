"""Per-GEOMETRY kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite database): one row per (kernel, grid,
workgroup, dynamic LDS) instead of one per kernel name - so that a kernel launched for several stages (the 8-wave
transposed conv serves g_s[2] AND g_s[4]; the conv serves g_a[2] AND g_a[4]) shows each stage's own mean duration.
  python tools/kernel_summary.py results.db out.csv [min_total_us]"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
min_total = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, lds_size, count(*), sum(end - start), min(end - start), max(end - start) "
                   "from kernels group by name, grid_x, grid_y, grid_z, workgroup_x, lds_size order by sum(end - start) desc").fetchall()
total = sum(r[7] for r in rows)
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Grid", "Workgroup", "DynLdsBytes", "Calls", "TotalDurationUs", "AverageUs", "MinUs", "MaxUs", "Percentage"])
    for name, gx, gy, gz, wx, lds, n, tot, mn, mx in rows:
        if tot / 1e3 < min_total:
            continue
        w.writerow([name.replace("licos::", ""), "%dx%dx%d" % (gx, gy, gz), wx, lds, n, round(tot / 1e3, 1), round(tot / n / 1e3, 2),
                    round(mn / 1e3, 2), round(mx / 1e3, 2), round(100.0 * tot / total, 2)])
print("wrote", out, "(%d geometries, %.1f ms of kernels)" % (len(rows), total / 1e6))
