"""Development aid: cost of unaligned block starts for the batch kernel."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
import terarkdb_amd.capi as capi

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

e = G.engine(8)
for blen, stride, n in ((8192, 8192, 131072), (8192, 8197, 131000), (8192, 8192 + 64, 130000), (8192, 8192 + 16, 130000),
                        (2041, 2047, 500000), (2048, 2048, 524288), (2041, 2048, 524288), (4096, 4096, 262144), (4096, 4101, 262000)):
    dev = G.splitmix_dev(((n * stride + 4096) // 8) * 8)
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    for shift in (0, 3):
        base = dev[shift:]
        ms = timeit(lambda: e.blocks_stamp(base, n, blen, stride, out, suffix_byte=-1))
        print(f"len={blen} stride={stride} base+{shift}: {ms*1e3:.1f} us  {n*blen/ms/1e6:.0f} GB/s", flush=True)
    del dev
