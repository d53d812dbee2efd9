"""Development aid: where the in-place trailer stamp loses against verification on packed variable-size blocks."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G  # noqa: E402
import terarkdb_amd.capi as capi  # noqa: E402

eng = capi.Engine(0)
target = int(os.environ.get("SP_TARGET", "4096"))
rng2 = np.random.default_rng(target)
nblk = (1 << 30) // (target + 133)
lens = (target + rng2.integers(0, 257, nblk)).astype(np.uint32)
offs = np.concatenate([[0], np.cumsum(lens.astype(np.uint64) + 5)[:-1]]).astype(np.uint64)
total = int(offs[-1] + lens[-1] + 5)
image = G.splitmix_dev((total + 7) // 8 * 8, 31)
types = torch.zeros(nblk, dtype=torch.uint8, device="cuda")
out = torch.zeros(nblk, dtype=torch.int32, device="cuda")
d_len = capi.make_desc_tensor(offs, lens, np.zeros(nblk, np.uint32), "cuda")
d_len1 = capi.make_desc_tensor(offs, lens + 1, np.zeros(nblk, np.uint32), "cuda")
vcounts = torch.zeros(3, dtype=torch.int64, device="cuda")
M, SA = capi.F_MASK, capi.F_STORE_AFTER
variants = [
    ("verify, stored value behind the span", lambda: eng.batch_verify(image, d_len1, nblk, vcounts, flags=capi.F_EXPECT_AFTER)),
    ("stamp to an array, type byte inside the span", lambda: eng.batch_stamp(image, d_len1, nblk, out, flags=M)),
    ("stamp to an array, type bytes from an array", lambda: eng.batch_stamp(image, d_len, nblk, out, suffix=types, flags=M)),
    ("stamp in place (4 bytes), type byte inside the span", lambda: eng.batch_stamp(image, d_len1, nblk, None, flags=M | SA)),
    ("stamp in place (type + 4 bytes), type bytes from an array", lambda: eng.batch_stamp(image, d_len, nblk, None, suffix=types, flags=M | SA)),
    ("stamp in place AND to an array, type byte inside the span", lambda: eng.batch_stamp(image, d_len1, nblk, out, flags=M | SA)),
    ("stamp in place AND to an array, type bytes from an array", lambda: eng.batch_stamp(image, d_len, nblk, out, suffix=types, flags=M | SA)),
]
for rep in range(2):
    for name, fn in variants:
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print(f"{name:62s} {ms*1e3:7.1f} us {float(lens.sum() + nblk) / ms / 1e6:7.0f} GB/s", flush=True)
