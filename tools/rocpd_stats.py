#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace) as per-kernel totals -> CSV on stdout."""
import csv
import glob
import sqlite3
import sys

path = sys.argv[1]
dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
rows = {}
for p in dbs:
    cur = sqlite3.connect(p).cursor()
    for name, n, tot, mn, mx in cur.execute("select name, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by name"):
        r = rows.setdefault(name, [0, 0, 1 << 62, 0])
        r[0] += n; r[1] += tot; r[2] = min(r[2], mn); r[3] = max(r[3], mx)
total = sum(r[1] for r in rows.values()) or 1
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
for name, r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    w.writerow([name, r[0], r[1], round(r[1] / r[0], 1), r[2], r[3], round(100.0 * r[1] / total, 3)])
