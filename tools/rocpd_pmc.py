#!/usr/bin/env python3
"""Per-kernel sums of the PMC counters in rocprofv3 rocpd databases -> CSV on stdout."""
import csv
import glob
import sqlite3
import sys

path = sys.argv[1]
dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
acc = {}
for p in dbs:
    con = sqlite3.connect(p)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    if not cols:
        continue
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    q = f"select {name_col}, counter_name, sum(value), count(*) from counters_collection group by {name_col}, counter_name"
    for kn, cn, v, n in cur.execute(q):
        acc.setdefault(kn, {})[cn] = (acc.get(kn, {}).get(cn, (0, 0))[0] + v, n)
w = csv.writer(sys.stdout)
names = sorted({c for k in acc.values() for c in k})
w.writerow(["Kernel", "Dispatches"] + names)
for kn, d in sorted(acc.items()):
    w.writerow([kn[:90], max(n for _, n in d.values())] + [d.get(c, ("", 0))[0] for c in names])
