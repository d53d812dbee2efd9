#!/usr/bin/env python3
"""Development aid: how long after a synchronisation the first launches of the benchmark kernel run
below their steady rate (clock ramp), as a function of the number of timed launches."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G  # noqa: E402
import terarkdb_amd.capi as capi  # noqa: E402

eng = capi.Engine(0)
N, B = 131072, 8192
data = G.splitmix_dev(N * B, 301)
stored = torch.zeros(N, dtype=torch.int32, device="cuda")
out = torch.zeros_like(stored)
counts = torch.zeros(3, dtype=torch.int64, device="cuda")
eng.blocks_stamp(data, N, B, B, stored, suffix_byte=0, flags=capi.F_MASK)
torch.cuda.synchronize()


def run(k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        eng.blocks_verify(data, N, B, B, counts, expected=stored, out=out, suffix_byte=0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3


t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.5:
    run(100)
for idle_ms in (0, 0, 1, 5, 20, 0):
    row = []
    for k in (20, 20, 100, 20):
        if idle_ms:
            time.sleep(idle_ms / 1e3)
        row.append(f"{k}x {run(k):6.1f}")
    print(f"after {idle_ms:2d} ms idle: " + "  ".join(row) + " us/launch", flush=True)
