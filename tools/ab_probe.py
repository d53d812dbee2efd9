"""Development aid: the log scan's two kernels (TCRC_WAL_SPLIT=0: one control wave; default: walker and placer
apart) timed on the same box, in the same process, on the same images - boxes of the pool differ by a few per cent,
so only such a ratio says whether a change helped."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_lib as O
import terarkdb_amd.capi as capi

def engines():
    os.environ["TCRC_WAL_SPLIT"] = "0"
    a = capi.Engine(0)
    os.environ["TCRC_WAL_SPLIT"] = "1"
    b = capi.Engine(0)
    return a, b

def small_image(lo, hi, total):
    rng = np.random.default_rng(1)
    filler = O.splitmix(1 << 16, 9).tobytes()
    t = O.OracleLog(recycle=False, log_number=9)
    acc = 0
    while acc < total:
        n = int(rng.integers(lo, hi + 1))
        t.write(filler[:n]); acc += n + 7
    img = np.frombuffer(t.contents(), dtype=np.uint8).copy(); t.close()
    return img, 9

def timed(eng, dev, size, lognum, blocks, counts, reps):
    for _ in range(10):
        eng.wal_verify(dev, size, lognum, blocks, counts, want_records=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.wal_verify(dev, size, lognum, blocks, counts, want_records=False)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def main():
    import measure_configs as MC
    fused, split = engines()
    which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["s40", "s128", "s512", "s1k", "mixed", "2k"]
    shapes = {"s40": lambda: small_image(0, 40, 8 << 20), "s128": lambda: small_image(100, 160, 64 << 20),
              "s512": lambda: small_image(400, 600, 64 << 20), "s1k": lambda: small_image(900, 1100, 64 << 20),
              "mixed": lambda: MC.log_image("mixed")[:2], "2k": lambda: MC.log_image("2k")[:2]}
    # settle the clocks
    x = torch.zeros(1 << 28, dtype=torch.uint8, device="cuda")
    for _ in range(300):
        x.add_(1)
    torch.cuda.synchronize()
    for name in which:
        img, lognum = shapes[name]()
        nb = (img.size + 32767) // 32768
        want = np.zeros((nb, 4), np.uint32)
        O.lib().oracle_wal_scan_blocks(img.ctypes.data, img.size, lognum, want.ctypes.data)
        dev = torch.from_numpy(img).cuda()
        blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
        counts = torch.zeros(3, dtype=torch.int64, device="cuda")
        for eng in (fused, split):
            blocks.zero_(); eng.counts_reset(counts)
            eng.wal_verify(dev, img.size, lognum, blocks, counts)
            assert (blocks.cpu().numpy().view(np.uint32) == want).all(), name
        reps = 20 if img.size < (1 << 30) else 10
        tf, ts = [], []
        for _ in range(3):
            tf.append(timed(fused, dev, img.size, lognum, blocks, counts, reps))
            ts.append(timed(split, dev, img.size, lognum, blocks, counts, reps))
        print(f"{name:6s} {img.size/1e6:8.1f} MB  fused {min(tf):8.1f} us   split {min(ts):8.1f} us   split/fused {min(ts)/min(tf):.3f}", flush=True)
        del dev
    fused.close(); split.close()

if __name__ == "__main__":
    main()
