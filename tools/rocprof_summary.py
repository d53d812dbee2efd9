#!/usr/bin/env python3
"""Turns rocprofv3 result databases (gpurun_out/...) into the small summaries kept under profiles/.

usage: rocprof_summary.py kernel <kt_results.db> <out.csv>
       rocprof_summary.py pmc <fetch_results.db> <write_results.db> <kernel-substring> <out.json>
       rocprof_summary.py sq <results.db> <kernel-substring> <out.json>
"""
import datetime
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
        "collected": datetime.date.today().isoformat(),
        "fetch_size_kib_avg": f_kb, "fetch_dispatches": f_n,
        "write_size_kib_avg": w_kb, "write_dispatches": w_n,
        "read_bytes_per_launch": read_bytes, "write_bytes_per_launch": write_bytes,
        "bytes_per_launch": read_bytes + write_bytes,
        "correction": "FETCH_SIZE x2 (gfx950 counts 128-B streaming requests as 64 B); WRITE_SIZE as reported; "
                      "separate --pmc passes",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(res)


def sq(db, needle, out):
    """averages per dispatch of whatever SQ counters the pass collected, and the ratios read off them"""
    c = sqlite3.connect(db)
    rows = c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like ? "
                     "group by counter_name", (f"%{needle}%",)).fetchall()
    v = {name: val for name, val, _ in rows}
    res = {"kernel": needle, "dispatches": max((n for _, _, n in rows), default=0),
           "counters_avg_per_dispatch": {k: v[k] for k in sorted(v)}}
    wc = v.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if k in v:
                res[k.lower() + "_per_wave_cycle"] = round(v[k] / wc, 4)
    if v.get("SQ_LDS_IDX_ACTIVE"):
        res["lds_bank_conflict_cycles_per_lds_cycle"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"], 5)
    res["note"] = ("one rocprofv3 --pmc pass (8 SQ counters); SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count "
                   "quad-cycles summed over waves; SQ_LDS_BANK_CONFLICT = extra LDS cycles, SQ_LDS_IDX_ACTIVE = all LDS cycles")
    json.dump(res, open(out, "w"), indent=1)
    print(res)


if __name__ == "__main__":
    if sys.argv[1] == "kernel":
        kernel_stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "sq":
        sq(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
