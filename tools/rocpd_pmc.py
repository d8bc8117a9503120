#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in rocprofv3 rocpd sqlite files.  usage: rocpd_pmc.py <db>..."""
import sqlite3
import sys

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cur = db.cursor()
    try:
        cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
        rows = cur.execute("select * from counters_collection").fetchall()
    except sqlite3.Error as e:
        print(path, "no counters:", e)
        continue
    ni, ci, vi = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name"), cols.index("counter_name"), cols.index("value")
    agg = {}
    for r in rows:
        key = (r[ni][:90], r[ci])
        s = agg.setdefault(key, [0, 0.0])
        s[0] += 1
        s[1] += float(r[vi])
    print("==", path)
    for (k, c), (n, tot) in sorted(agg.items()):
        if "lphq" in k:
            print(f"{c:28s} n={n:4d} avg={tot / n:18.1f}  {k}")
