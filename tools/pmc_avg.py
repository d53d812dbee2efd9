#!/usr/bin/env python3
"""Average value of one PMC counter per kernel name from a rocprofv3 --pmc database.
usage: pmc_avg.py <results.db> <COUNTER>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
q = "select kernel_name, avg(value), count(*) from counters_collection where counter_name=? group by kernel_name"
for name, v, n in c.execute(q, (sys.argv[2],)):
    print(f"{v:16.1f}  x{n:<5d} {name[:100]}")
