#!/usr/bin/env python3
"""Development aid: lists the kernel dispatches of a rocprofv3 --kernel-trace database in launch order.
usage: kt_list.py <results.db> [name-substring]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
sym = [t for t in tabs if "kernel_symbol" in t][0]
needle = sys.argv[2] if len(sys.argv) > 2 else ""
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {sym} s on d.kernel_id=s.id order by d.start")
prev_end = None
for name, st, en in rows:
    if needle in name:
        gap = "" if prev_end is None else f" gap {(st - prev_end) / 1000:8.1f}"
        print(f"{(en - st) / 1000:9.1f} us{gap}  {name[:70]}")
    prev_end = en
