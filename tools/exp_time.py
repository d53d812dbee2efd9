"""Development aid: times the batch kernel on 2046-byte spans at every alignment, without checking results
(for builds with parts of the kernel stubbed out)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
import terarkdb_amd.capi as capi
eng = capi.Engine(0)
total = 1 << 30
data = G.splitmix_dev(total, 3)
for L, stride in ((2046, 2053), (4090, 4101), (1000, 1003)):
    n = (total - L) // stride
    offs = np.arange(n, dtype=np.uint64) * stride
    desc = capi.make_desc_tensor(offs, np.full(n, L, np.uint32), np.zeros(n, np.uint32), "cuda")
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    hint = capi.F_SHORT_SPANS if os.environ.get('EXP_SHORT_HINT') and L < 1400 else 0
    fn = lambda: eng.batch_stamp(data, desc, n, out, flags=capi.F_MASK | hint)
    t_end = time.perf_counter() + 1.5
    while time.perf_counter() < t_end:
        for _ in range(20): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"{os.environ.get('TCRC_LIBRARY','product'):50s} spans of {L}: {us:7.1f} us  {n*L/us/1e3:7.1f} GB/s", flush=True)
