"""Kernel timeline of the LAST `compress` + `decompress` pair in a rocprofv3 --kernel-trace run of tools/hyper_probe.py
(reads the rocpd sqlite database): start / end (ms, relative) and stream of every kernel longer than `min_us`.
  python tools/timeline.py results.db [min_us]"""
import sqlite3
import sys

db = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
con = sqlite3.connect(db)
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select * from kernels order by start").fetchall()
ix = {c: i for i, c in enumerate(cols)}
name_c = "name" if "name" in ix else [c for c in cols if "name" in c][0]
# the last step starts at the third-from-last launch of the first-stage conv... simpler: take the last 2.5 s worth and let the reader cut
t_end = max(r[ix["end"]] for r in rows)
sel = [r for r in rows if (r[ix["end"]] - r[ix["start"]]) / 1e3 >= min_us]
# find the start of the last compress: last occurrence of a gap > 100 ms is unreliable; print the last N
last = sel[-int(sys.argv[3]) if len(sys.argv) > 3 else -80:]
t0 = last[0][ix["start"]]
for r in last:
    nm = r[ix[name_c]]
    nm = nm.split("(")[0].replace("void ", "").replace("licos::", "")[:48]
    q = r[ix["queue_id"]] if "queue_id" in ix else (r[ix["stream_id"]] if "stream_id" in ix else "")
    print("%8.2f %8.2f %7.2f  q%-4s %s" % ((r[ix["start"]] - t0) / 1e6, (r[ix["end"]] - t0) / 1e6, (r[ix["end"]] - r[ix["start"]]) / 1e6, q, nm))
