#!/usr/bin/env python3
"""Turns gpurun_out/collect/ (tools/collect_profiles.sh) into the tracked files under profiles/ for one round:
rNN_bench.json, rNN_bench_kernel_stats.csv, rNN_bench_pmc_hbm_traffic.json (+ hbm_traffic.json, which bench.py
reads), rNN_bench_8gib_shard.json, rNN_<config>_kernel_stats.csv, rNN_<config>_hbm_traffic.json and
rNN_configs.json: one JSON line per configuration with roofline.traffic filled in from the PMC passes and the
kernel's average duration in the rocprofv3 trace beside the event-timed one.

usage: merge_profiles.py PREFIX   (the prefix the files get, e.g. the collection's label)"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "collect")
dst = os.path.join(ROOT, "profiles")
rnd = sys.argv[1]


def cp(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))


cp("bench.json", f"{rnd}_bench.json")
cp("bench_kernel_stats.csv", f"{rnd}_bench_kernel_stats.csv")
cp("hbm_traffic.json", f"{rnd}_bench_pmc_hbm_traffic.json")
cp("hbm_traffic.json", "hbm_traffic.json")
cp("bench_8gib_shard.json", f"{rnd}_bench_8gib_shard.json")
cp("latency.txt", f"{rnd}_latency.txt")
cp("bench_sq.json", f"{rnd}_bench_sq.json")
cp("frame.txt", f"{rnd}_frame.txt")
cp("log_small.txt", f"{rnd}_log_small.txt")


def trace_avg(path, needle):
    """average duration (us) and calls of the dispatches whose kernel name holds `needle`, weighted over its variants"""
    if not os.path.exists(path):
        return None, None
    tot = calls = 0.0
    for row in csv.reader(l for l in open(path) if not l.startswith("#")):
        if row and row[0] != "kernel" and needle in row[0]:
            tot += float(row[2])
            calls += float(row[1])
    return (round(tot / calls, 2), int(calls)) if calls else (None, None)


out = []
for l in open(os.path.join(src, "configs.jsonl")):
    d = json.loads(l)
    name = d["config"]
    base = name[:-len("_enqueued")] if name.endswith("_enqueued") else name
    key = {"long_64x4096k": "long_64x4m", "long_1x262144k": "long_256m", "long_1x1048576k": "long_1g"}.get(base, base)
    if key.startswith("long_1x"):
        key = "long_single"
    # several shapes in one profiled command, or the waiting and the enqueued calls of one shape (the same kernel on
    # the same buffers; the trace's average is over both): no per-line kernel average for the enqueued line
    shared_run = key == "long_single" or name.endswith("_enqueued")
    needle = d["roofline"]["kernel"].split("<")[0].split(" ")[0]
    if key:
        cp(f"{key}_kernel_stats.csv", f"{rnd}_{key}_kernel_stats.csv")
        cp(f"{key}_hbm_traffic.json", f"{rnd}_{key}_hbm_traffic.json")
        avg, calls = trace_avg(os.path.join(src, f"{key}_kernel_stats.csv"), needle)
        if avg is not None and not shared_run:
            d["roofline"]["rocprofv3_kernel_avg_us"] = avg
            d["roofline"]["rocprofv3_kernel_calls"] = calls
        t = os.path.join(src, f"{key}_hbm_traffic.json")
        if os.path.exists(t) and not shared_run:
            d["roofline"]["traffic"] = json.load(open(t)).get("bytes_per_launch")
        d["evidence"] = [f"profiles/{rnd}_{key}_kernel_stats.csv", f"profiles/{rnd}_{key}_hbm_traffic.json"]
        if os.path.exists(os.path.join(src, f"{key}_sq.json")):
            cp(f"{key}_sq.json", f"{rnd}_{key}_sq.json")
            d["evidence"].append(f"profiles/{rnd}_{key}_sq.json")
    out.append(d)
with open(os.path.join(dst, f"{rnd}_configs.json"), "w") as f:
    for d in out:
        f.write(json.dumps(d) + "\n")
print(f"wrote {len(out)} configuration lines")
