#!/usr/bin/env python3
"""Per-kernel averages of all counters in a rocprofv3 --pmc database.  usage: pmc_table.py <results.db> [substr]"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
needle = sys.argv[2] if len(sys.argv) > 2 else "k_crc_batch"
rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                 "where kernel_name like ? group by kernel_name, counter_name", (f"%{needle}%",))
tab = defaultdict(dict)
for k, cn, v, n in rows:
    tab[k[:70] + f" x{n}"][cn] = v
for k, d in tab.items():
    print(k)
    for cn in sorted(d):
        print(f"   {cn:24s} {d[cn]:16.0f}")
