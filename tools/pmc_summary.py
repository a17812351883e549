#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd sqlite outputs: per-kernel mean duration and mean counter values.
usage: pmc_summary.py <results.db> [...]"""
import sqlite3
import sys
from collections import defaultdict

for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    print("==", db)
    try:
        rows = c.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection "
                         "group by kernel_name, counter_name").fetchall()
    except sqlite3.Error:
        rows = []
    per = defaultdict(dict)
    for k, cn, v, n, d in rows:
        per[k][cn] = (v, n, d)
    for k, cs in per.items():
        print(" kernel:", k[:110])
        for cn, (v, n, d) in sorted(cs.items()):
            print(f"   {cn:34s} {v:16.1f}   (n={n}, mean dur {d/1e3:.2f} us)")
    if not per:
        for k, n, d, mn in c.execute("select name, count(*), avg(duration), min(duration) from kernels group by name"):
            print(f" {k[:100]:100s} n={n} mean {d/1e3:.2f} us min {mn/1e3:.2f} us")
