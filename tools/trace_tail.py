#!/usr/bin/env python3
"""Prints the last N kernel dispatches of a rocprofv3 --kernel-trace database with their durations and
the gaps between them.  usage: trace_tail.py <results.db> [N]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = list(c.execute("select name, start, end from kernels order by start"))
prev_end = None
for name, s, e in rows[-n:]:
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:8.1f} us"
    print(f"{(e - s) / 1e3:9.1f} us  {gap:18s} {name[:90]}")
    prev_end = e
