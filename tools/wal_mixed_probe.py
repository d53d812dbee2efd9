#!/usr/bin/env python3
"""Development aid: verifies a 256 MiB log of mixed record sizes a few times (run it under
rocprofv3 --kernel-trace to see the per-kernel times of one scan)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import terarkdb_amd.capi as capi  # noqa: E402

OL = O.lib()
tl = O.OracleLog(recycle=False, log_number=123, hw=True)
rnd = O.Random(301)
while tl.written_bytes() < 256 * 1024 * 1024:
    tl.write(O.big_string(b"%d." % tl.written_bytes(), rnd.skewed(17)))
size = tl.written_bytes()
img = np.frombuffer((C.c_uint8 * size).from_address(OL.oracle_sink_data(tl.sink)), dtype=np.uint8).copy()
dev = torch.from_numpy(img).cuda()
nb = (size + 32767) // 32768
blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
counts = torch.zeros(3, dtype=torch.int64, device="cuda")
eng = capi.Engine(0)
for i in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.counts_reset(counts)
    n = eng.wal_verify(dev, size, 123, blocks, counts)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"verify {size} bytes, {n} records: {dt*1e6:.0f} us  {size/dt/1e9:.0f} GB/s", flush=True)
eng.close()
