"""Development aid: times the log scan on the mixed-size log without checking results (for builds with parts stubbed out)."""
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
dev = torch.from_numpy(img).cuda()
nb = (size + 32767) // 32768
blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
counts = torch.zeros(3, dtype=torch.int64, device="cuda")
fn = lambda: eng.wal_verify(dev, size, 123, blocks, counts, want_records=False)
t_end = time.perf_counter() + 1.5
while time.perf_counter() < t_end:
    for _ in range(20): fn()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fn()
e1.record(); torch.cuda.synchronize()
print(f"{os.path.basename(os.environ.get('TCRC_LIBRARY','product')):40s} mixed log: {e0.elapsed_time(e1)/50*1e3:7.1f} us", flush=True)
