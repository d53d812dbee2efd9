"""Development aid: device log verification of the db_bench-shaped image, for rocprofv3 kernel traces."""
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
n_rec = int(os.environ.get("MP_RECORDS", "500000"))
import gpu_util as G
t = O.OracleLog(recycle=False, log_number=7, hw=True)
OL.oracle_wal_write_dbbench_records(t.writer, n_rec, 301)
size = t.written_bytes()
crc_bytes = OL.oracle_wal_writer_crc_bytes(t.writer)
img = np.frombuffer((C.c_uint8 * size).from_address(OL.oracle_sink_data(t.sink)), dtype=np.uint8).copy()
dev = torch.from_numpy(img).cuda()
nb = (size + 32767) // 32768
blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
counts = torch.zeros(3, dtype=torch.int64, device="cuda")
for i in range(8):
    if i % 4 == 0:
        lanes = (8, 4)[(i // 4) % 2]
        eng = G.engine(lanes)
        print("lanes per record:", lanes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.counts_reset(counts)
    n = eng.wal_verify(dev, size, 7, blocks, counts)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"verify {size} bytes, {n} records: {dt*1e6:.0f} us  {crc_bytes/dt/1e9:.0f} GB/s", flush=True)
