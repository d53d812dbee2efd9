#!/usr/bin/env python3
"""Development aid: rate of the loads-only probe kernel for different workgroups-per-CU settings and
load policies (the ceiling the CRC kernels are compared with)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import gpu_util as G
    import terarkdb_amd.capi as capi
    eng = G.engine()
    n = 1 << 30
    data = G.splitmix_dev(n, 1)
    scratch = torch.zeros(2 * 1024 * 256 * 16, dtype=torch.int32, device="cuda")
    for w in (1, 2):
        for _ in range(200):
            eng.probe_read(data, n, scratch, w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            eng.probe_read(data, n, scratch, w)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 200
        print(f"nontemporal={os.environ.get('TCRC_NONTEMPORAL', '1')} wgs_per_cu={w}: {ms*1e3:7.1f} us {n/ms/1e6:6.0f} GB/s", flush=True)
else:
    for nt in ("1", "0"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, TCRC_NONTEMPORAL=nt), check=True)
