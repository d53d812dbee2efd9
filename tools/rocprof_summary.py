#!/usr/bin/env python3
"""Turns rocprofv3 result databases (gpurun_out/...) into the small summaries kept under profiles/.

usage: rocprof_summary.py kernel <kt_results.db> <out.csv>
       rocprof_summary.py pmc <fetch_results.db> <write_results.db> <kernel-substring> <out.json>
"""
import json
import sqlite3
import sys


def kernel_stats(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n")
        f.write("kernel,calls,total_us,average_us,percent\n")
        for name, calls, total, avg, pct in rows:
            short = name if len(name) < 120 else name[:117] + "..."
            f.write(f"\"{short}\",{calls},{total:.3f},{avg:.3f},{pct:.2f}\n")
    for r in rows[:3]:
        print(r)


def pmc(fetch_db, write_db, needle, out):
    def avg(db, counter):
        c = sqlite3.connect(db)
        q = ("select avg(value), count(*) from counters_collection where counter_name=? and kernel_name like ?")
        return c.execute(q, (counter, f"%{needle}%")).fetchone()

    f_kb, f_n = avg(fetch_db, "FETCH_SIZE")
    w_kb, w_n = avg(write_db, "WRITE_SIZE")
    # FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies each 128-byte request of a wide
    # coalesced streaming read as 64 bytes, so the read figure is doubled; WRITE_SIZE is exact.
    read_bytes = 2.0 * f_kb * 1024.0
    write_bytes = w_kb * 1024.0
    res = {
        "kernel": needle,
        "fetch_size_kib_avg": f_kb, "fetch_dispatches": f_n,
        "write_size_kib_avg": w_kb, "write_dispatches": w_n,
        "read_bytes_per_launch": read_bytes, "write_bytes_per_launch": write_bytes,
        "bytes_per_launch": read_bytes + write_bytes,
        "correction": "FETCH_SIZE x2 (gfx950 counts 128-B streaming requests as 64 B); WRITE_SIZE as reported; "
                      "separate --pmc passes",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(res)


if __name__ == "__main__":
    if sys.argv[1] == "kernel":
        kernel_stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
