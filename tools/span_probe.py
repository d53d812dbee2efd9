#!/usr/bin/env python3
"""Development aid: the batch CRC kernel alone on a descriptor list of equal spans (log-record shaped: length L at a
stride of L + 7, so every alignment occurs), per lanes-per-buffer geometry."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G  # noqa: E402
import terarkdb_amd.capi as capi  # noqa: E402

total = 1 << 30
data = G.splitmix_dev(total, 3)
shapes = [(2046, 2053), (2048, 2048), (506, 513), (4090, 4097), (8193, 8197), (100, 107)]
for lanes in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,8").split(",")]:
    eng = G.engine(lanes)
    for L, stride in shapes:
        n = (total - L) // stride
        offs = np.arange(n, dtype=np.uint64) * stride
        desc = capi.make_desc_tensor(offs, np.full(n, L, np.uint32), np.zeros(n, np.uint32), "cuda")
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
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
        print(f"lanes {lanes:2d} len {L:5d} stride {stride:5d}: {ms*1e3:7.1f} us {n*L/ms/1e6:6.0f} GB/s", flush=True)
    eng.close()
