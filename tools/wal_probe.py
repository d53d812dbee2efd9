"""Development aid: device-resident log verification timings for the config-1 log (2 KiB records) and a mixed log."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import terarkdb_amd.capi as capi
OL = O.lib()
eng = capi.Engine(0)
n_rec = int(os.environ.get("WP_RECORDS", "1000000"))


def timed(fn, reps=7):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def run(img, lognum, label, covered):
    dev = torch.from_numpy(img).cuda()
    nb = (img.size + 32767) // 32768
    blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    def v():
        eng.counts_reset(counts)
        return eng.wal_verify(dev, img.size, lognum, blocks, counts)
    n = v(); torch.cuda.synchronize()
    want = np.zeros((nb, 4), np.uint32)
    OL.oracle_wal_scan_blocks(img.ctypes.data, img.size, lognum, want.ctypes.data)
    ok = (blocks.cpu().numpy().view(np.uint32) == want).all()
    t = timed(v)
    def enq():
        eng.wal_verify(dev, img.size, lognum, blocks, counts, want_records=False)
    # settle: the clocks come up only after about a second of sustained work (the headline bench does the same)
    t_end = time.perf_counter() + 2.0
    while time.perf_counter() < t_end:
        for _ in range(50): enq()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): enq()
    e1.record(); torch.cuda.synchronize()
    tp = e0.elapsed_time(e1) / 50 * 1e-3
    print(f"{label}: back-to-back {tp*1e6:.1f} us per scan, {covered/tp/1e9:.0f} GB/s ({100*covered/tp/8e12:.1f}% of peak)")
    print(f"{label}: {img.size} B, {n} records, {t*1e6:.1f} us, {covered/t/1e9:.0f} GB/s ({100*covered/t/8e12:.1f}% of peak) ok={ok}", flush=True)


t = O.OracleLog(recycle=False, log_number=7, hw=True)
OL.oracle_wal_write_dbbench_records(t.writer, n_rec, 301)
size = t.written_bytes(); crc_bytes = int(OL.oracle_wal_writer_crc_bytes(t.writer))
img = np.frombuffer((C.c_uint8 * size).from_address(OL.oracle_sink_data(t.sink)), dtype=np.uint8).copy()
run(img, 7, "config1 (2 KiB records)", crc_bytes)
t.close()
for recycle in (0, 1):
    tl = O.OracleLog(recycle=bool(recycle), log_number=123)
    rnd = O.Random(301)
    while tl.written_bytes() < 256 * 1024 * 1024:
        tl.write(O.big_string(b"%d." % tl.written_bytes(), rnd.skewed(17)))
    s3 = tl.written_bytes()
    img3 = np.frombuffer((C.c_uint8 * s3).from_address(OL.oracle_sink_data(tl.sink)), dtype=np.uint8).copy()
    run(img3, 123, f"config3 mixed (recycle={recycle})", s3)
    tl.close()
# small records: 100-byte payloads
tl = O.OracleLog(recycle=False, log_number=9)
msg = bytes(range(100))
for _ in range(600000): tl.write(msg)
s4 = tl.written_bytes()
img4 = np.frombuffer((C.c_uint8 * s4).from_address(OL.oracle_sink_data(tl.sink)), dtype=np.uint8).copy()
run(img4, 9, "100-byte records", s4)
