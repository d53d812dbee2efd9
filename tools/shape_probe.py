#!/usr/bin/env python3
"""Development aid: kernel rate for descriptor batches of ~2 KiB buffers in different layouts
(aligned / packed like log records / regular) and lanes-per-buffer settings."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G  # noqa: E402
import terarkdb_amd.capi as capi  # noqa: E402

n = 1_000_000
data = G.splitmix_dev(((n * 2056 + 4096) // 8) * 8, 3)
out = torch.zeros(n, dtype=torch.int32, device="cuda")
shapes = {
    "len 2048 stride 2048 (regular)": (2048, 2048, 0),
    "len 2041 stride 2048 aligned start": (2041, 2048, 0),
    "len 2041 stride 2048 start+13": (2041, 2048, 13),
    "len 2041 stride 2054 packed (log-like)": (2041, 2054, 13),
    "len 2041 stride 2054 shuffled order": (2041, 2054, -1),
}
for lanes in (4, 8):
    eng = G.engine(lanes)
    for name, (ln, stride, start) in shapes.items():
        offs = np.arange(n, dtype=np.uint64) * stride + (13 if start < 0 else start)
        if start < 0:
            # keep runs of 16 consecutive records together (one log block's worth), shuffle the runs
            runs = offs.reshape(-1, 16)
            offs = runs[np.random.default_rng(1).permutation(len(runs))].reshape(-1)
        desc = capi.make_desc_tensor(offs, np.full(n, ln, np.uint32), np.zeros(n, np.uint32), "cuda")
        for _ in range(3):
            eng.batch_stamp(data, desc, n, out, flags=capi.F_MASK)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.batch_stamp(data, desc, n, out, flags=capi.F_MASK)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"lanes {lanes}: {name:42s} {ms*1e3:7.1f} us  {n*ln/ms/1e6:7.0f} GB/s", flush=True)
    eng.close()
