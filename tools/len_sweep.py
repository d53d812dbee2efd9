#!/usr/bin/env python3
"""Development aid: kernel time per round against buffer length on the regular path (uniform aligned
blocks), to separate the per-buffer fixed cost from the per-row cost."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G  # noqa: E402
import terarkdb_amd.capi as capi  # noqa: E402

total = 1 << 30
data = G.splitmix_dev(total, 3)
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = G.engine(lanes)
for blen in (512, 1024, 2048, 4096, 8192, 16384, 32768):
    n = total // blen
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    for _ in range(3):
        eng.blocks_stamp(data, n, blen, blen, out, suffix_byte=0, flags=capi.F_MASK)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        eng.blocks_stamp(data, n, blen, blen, out, suffix_byte=0, flags=capi.F_MASK)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    rounds_per_wave = n / (64 // lanes) / 4096
    print(f"lanes {lanes} len {blen:6d}: {ms*1e3:7.1f} us {total/ms/1e6:6.0f} GB/s  "
          f"{ms*1e3/rounds_per_wave:6.2f} us per wave-round ({blen//(16*lanes)} rows)", flush=True)
eng.close()
