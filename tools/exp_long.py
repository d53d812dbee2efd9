"""Development aid: wall time of single long-buffer calls over buffer size and segment length (values checked once
against the CPU oracle; the oracle is not timed)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import terarkdb_amd.capi as capi

eng = capi.Engine(0)
S = O.splitmix(int(os.environ.get("EXP_IMAGE_MIB", "64")) << 20, 3)
dev = torch.from_numpy(S).cuda()
out = torch.zeros(64, dtype=torch.int32, device="cuda")
sizes = [int(x) for x in os.environ.get("EXP_SIZES", "262144,1048576,4194304,16777216").split(",")]
segs = [int(x) for x in os.environ.get("EXP_SEGS", "0,512,1024,2048,4096").split(",")]
nbuf = int(os.environ.get("EXP_NBUF", "1"))
for sz in sizes:
    want = O.value(S[:sz].tobytes())
    row = []
    for seg in segs:
        fn = lambda: eng.extend_long(dev, [i * sz for i in range(nbuf)], [sz] * nbuf, [0] * nbuf, out, segment_len=seg)
        fn()
        assert int(out[:1].cpu().numpy().view(np.uint32)[0]) == want, (sz, seg)
        t_end = time.perf_counter() + 0.7
        while time.perf_counter() < t_end:
            for _ in range(50): fn()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(200): fn()
            best = min(best, (time.perf_counter() - t0) / 200)
        row.append(f"seg {seg:5d}: {best * 1e6:6.2f} us")
    print(f"{sz >> 10:6d} KiB  " + "  ".join(row), flush=True)
