"""Development aid: end-to-end latency of the smallest host calls (one block verified, one value)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import oracle_lib as O  # noqa: E402
import terarkdb_amd.capi as capi  # noqa: E402

L = capi.lib
eng = capi.Engine(0)
for n_bytes in (4096, 8192, 32768, 65000, 130000, 262144, 1 << 20, 4 << 20):
    block = O.splitmix(n_bytes, 5).tobytes()
    img = np.frombuffer(block + b"\x00" + int(O.mask(O.value(block + b"\x00"))).to_bytes(4, "little"), dtype=np.uint8).copy()
    d = (capi.Desc * 1)()
    d[0].offset, d[0].len, d[0].init = 0, n_bytes + 1, 0
    counts = (C.c_uint64 * 3)()
    ok = (C.c_uint8 * 1)()
    args = (eng.ctx, img.ctypes.data, img.size, d, 1, None, capi.F_EXPECT_AFTER, None, None, ok, counts)
    for _ in range(200):
        assert L.tcrc_host_batch_verify(*args) == 0 and ok[0] == 1
    t0 = time.perf_counter()
    for _ in range(2000):
        L.tcrc_host_batch_verify(*args)
    dt = (time.perf_counter() - t0) / 2000
    t0 = time.perf_counter()
    for _ in range(2000):
        O.value(block + b"\x00", variant="hw")
    dc = (time.perf_counter() - t0) / 2000
    print(f"verify one block of {n_bytes} B from pageable memory: {dt*1e6:.1f} us per call (CPU oracle incl. ctypes: {dc*1e6:.1f} us)", flush=True)
eng.close()
