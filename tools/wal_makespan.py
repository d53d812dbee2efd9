"""Development aid: the log-scan kernel's makespan (first workgroup's start to last workgroup's end, from the
tcrc_debug_wal_trace hook) over a number of single launches of the mixed-size log, beside the time per launch
of a queue of launches between two events."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import terarkdb_amd.capi as capi
OL = O.lib()
eng = capi.Engine(0)
capi.lib.tcrc_debug_wal_trace.argtypes = [C.c_void_p, C.c_void_p]
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
trace = torch.zeros(256 * 32, dtype=torch.int64, device="cuda")
fn = lambda: eng.wal_verify(dev, size, 123, blocks, counts, want_records=False)
for _ in range(400): fn()
torch.cuda.synchronize()
spans, walks, meds = [], [], []
for rep in range(20):
    for _ in range(20): fn()
    trace.zero_()
    capi.lib.tcrc_debug_wal_trace(eng.ctx, C.c_void_p(trace.data_ptr()))
    fn()
    torch.cuda.synchronize()
    capi.lib.tcrc_debug_wal_trace(eng.ctx, None)
    t = trace.cpu().numpy().reshape(256, 32).astype(np.float64)
    spans.append((t[:, 31].max() - t[:, 0].min()) / 100.0)
    walks.append((t[:, 2].max() - t[:, 0].min()) / 100.0)
    meds.append((np.median(t[:, 31]) - t[:, 0].min()) / 100.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fn()
e1.record(); torch.cuda.synchronize()
name = os.path.basename(os.environ.get("TCRC_LIBRARY", "product"))
print(f"{name:20s} makespan min/med/max {min(spans):.1f} {np.median(spans):.1f} {max(spans):.1f} us; median workgroup end {np.median(meds):.1f}; "
      f"last walk over {np.median(walks):.1f}; queued launches {e0.elapsed_time(e1) / 50 * 1e3:.1f} us each", flush=True)
