"""Development aid: scan time of logs of small records (the walk of a block is a chain of dependent loads)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import terarkdb_amd.capi as capi
eng = capi.Engine(0)
filler = O.splitmix(1 << 16, 9).tobytes()
for name, lo, hi, total in (("0..40 B", 0, 40, 8 << 20), ("100..160 B", 100, 160, 64 << 20), ("400..600 B", 400, 600, 64 << 20),
                            ("900..1100 B", 900, 1100, 64 << 20)):
    rng = np.random.default_rng(1)
    t = O.OracleLog(recycle=False, log_number=9)
    acc = 0
    while acc < total:
        n = int(rng.integers(lo, hi + 1))
        t.write(filler[:n])
        acc += n + 7
    img = np.frombuffer(t.contents(), dtype=np.uint8).copy()
    t.close()
    nb = (img.size + 32767) // 32768
    want = np.zeros((nb, 4), np.uint32)
    O.lib().oracle_wal_scan_blocks(img.ctypes.data, img.size, 9, want.ctypes.data)
    dev = torch.from_numpy(img).cuda()
    blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    n = eng.wal_verify(dev, img.size, 9, blocks, counts)
    assert (blocks.cpu().numpy().view(np.uint32) == want).all()
    for _ in range(20):
        eng.wal_verify(dev, img.size, 9, blocks, counts, want_records=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.wal_verify(dev, img.size, 9, blocks, counts, want_records=False)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    per_block = want[:, 0].max()
    print(f"{name:12s} {img.size/1e6:7.1f} MB {n:8d} records, at most {per_block} per block: {us:8.1f} us = {img.size/us/1e3:7.1f} GB/s, "
          f"{us/per_block*1e3:6.0f} ns per record of the fullest block", flush=True)
