"""Development aid: stamp and verify timings of the 1 GiB / 8 KiB batch (library chosen by TCRC_LIBRARY)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
import terarkdb_amd.capi as capi

def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

n, blen = 131072, 8192
dev = G.splitmix_dev(n * blen)
out = torch.zeros(n, dtype=torch.int32, device="cuda")
stored = torch.zeros(n, dtype=torch.int32, device="cuda")
counts = torch.zeros(3, dtype=torch.int64, device="cuda")
e = G.engine(int(os.environ.get("LANES", "8")))
e.blocks_stamp(dev, n, blen, blen, stored, suffix_byte=0)
for rep in range(3):
    ms = timeit(lambda: e.blocks_stamp(dev, n, blen, blen, out, suffix_byte=0))
    mv = timeit(lambda: e.blocks_verify(dev, n, blen, blen, counts, expected=stored, out=out, suffix_byte=0))
    print(f"{os.environ.get('TCRC_LIBRARY','new')[-16:]} stamp {ms*1e3:.1f} us {n*(blen+1)/ms/1e6:.0f} GB/s | verify {mv*1e3:.1f} us {n*(blen+1)/mv/1e6:.0f} GB/s", flush=True)
