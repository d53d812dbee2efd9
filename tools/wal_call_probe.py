"""Development aid: what one log-scan call costs from the host's side - the time the call itself takes to enqueue,
a call to completion (with the record count asked for), and the time per scan of a queue of scans - for mixed-size
logs of several sizes (library chosen by TCRC_LIBRARY)."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import terarkdb_amd.capi as capi
OL = O.lib()
eng = capi.Engine(0)
name = os.path.basename(os.environ.get("TCRC_LIBRARY", "product"))
for mib in (1, 8, 64, 256):
    tl = O.OracleLog(recycle=False, log_number=123)
    rnd = O.Random(301)
    while tl.written_bytes() < mib << 20:
        tl.write(O.big_string(b"%d." % tl.written_bytes(), rnd.skewed(17)))
    size = tl.written_bytes()
    img = np.frombuffer((C.c_uint8 * size).from_address(OL.oracle_sink_data(tl.sink)), dtype=np.uint8).copy()
    tl.close()
    dev = torch.from_numpy(img).cuda()
    nb = (size + 32767) // 32768
    blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    enq = lambda: eng.wal_verify(dev, size, 123, blocks, counts, want_records=False)
    full = lambda: eng.wal_verify(dev, size, 123, blocks, counts, want_records=True)
    t_end = time.perf_counter() + 1.0
    while time.perf_counter() < t_end:
        for _ in range(20): enq()
        torch.cuda.synchronize()
    # the call alone: a few at a time so that the queue never fills
    t_call = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4): enq()
        t_call.append((time.perf_counter() - t0) / 4)
    torch.cuda.synchronize()
    t_full = []
    for _ in range(50):
        t0 = time.perf_counter()
        full()
        t_full.append(time.perf_counter() - t0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): enq()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:18s} {mib:4d} MiB: enqueue {np.median(t_call)*1e6:6.1f} us, call to completion {np.median(t_full)*1e6:6.1f} us, "
          f"queued {e0.elapsed_time(e1)/50*1e3:6.1f} us per scan", flush=True)
