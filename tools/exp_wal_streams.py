"""Development aid: scans of the mixed-size log queued on one stream against the same number dealt to two and four
streams (the tail of one scan then overlaps the head of the next)."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import terarkdb_amd.capi as capi
OL = O.lib()
eng = capi.Engine(0)
tl = O.OracleLog(recycle=False, log_number=123)
rnd = O.Random(301)
while tl.written_bytes() < 256 * 1024 * 1024:
    tl.write(O.big_string(b"%d." % tl.written_bytes(), rnd.skewed(17)))
size = tl.written_bytes()
img = np.frombuffer((C.c_uint8 * size).from_address(OL.oracle_sink_data(tl.sink)), dtype=np.uint8).copy()
nb = (size + 32767) // 32768
for nstreams in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    devs = [torch.from_numpy(img).cuda() for _ in range(nstreams)]
    blocks = [torch.zeros((nb, 4), dtype=torch.int32, device="cuda") for _ in range(nstreams)]
    counts = [torch.zeros(3, dtype=torch.int64, device="cuda") for _ in range(nstreams)]
    def run(k):
        for i in range(k):
            s = i % nstreams
            eng.wal_verify(devs[s], size, 123, blocks[s], counts[s], want_records=False, stream=streams[s].cuda_stream)
    t_end = time.perf_counter() + 1.0
    while time.perf_counter() < t_end:
        run(40)
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(200)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bad = sum(int(c.cpu()[0]) for c in counts)
    print(f"{nstreams} stream(s): {dt / 200 * 1e6:7.1f} us per scan, mismatches {bad}", flush=True)
