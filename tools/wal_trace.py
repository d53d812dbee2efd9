"""Development aid: phase timeline of the log-scan kernel's workgroups (tcrc_debug_wal_trace)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import terarkdb_amd.capi as capi
OL = O.lib()
eng = capi.Engine(0)
capi.lib.tcrc_debug_wal_trace.argtypes = [C.c_void_p, C.c_void_p]
which = sys.argv[1] if len(sys.argv) > 1 else "mixed"
if which == "mixed":
    tl = O.OracleLog(recycle=False, log_number=123)
    rnd = O.Random(301)
    while tl.written_bytes() < 256 * 1024 * 1024:
        tl.write(O.big_string(b"%d." % tl.written_bytes(), rnd.skewed(17)))
    lognum = 123
else:
    tl = O.OracleLog(recycle=False, log_number=7, hw=True)
    OL.oracle_wal_write_dbbench_records(tl.writer, 1000000, 301)
    lognum = 7
size = tl.written_bytes()
img = np.frombuffer((C.c_uint8 * size).from_address(OL.oracle_sink_data(tl.sink)), dtype=np.uint8).copy()
dev = torch.from_numpy(img).cuda()
nb = (size + 32767) // 32768
blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
counts = torch.zeros(3, dtype=torch.int64, device="cuda")
trace = torch.zeros(256 * 32, dtype=torch.int64, device="cuda")
for _ in range(200):
    eng.wal_verify(dev, size, lognum, blocks, counts, want_records=False)
torch.cuda.synchronize()
capi.lib.tcrc_debug_wal_trace(eng.ctx, C.c_void_p(trace.data_ptr()))
eng.wal_verify(dev, size, lognum, blocks, counts, want_records=False)
torch.cuda.synchronize()
capi.lib.tcrc_debug_wal_trace(eng.ctx, None)
t = trace.cpu().numpy().reshape(256, 32).astype(np.float64)
t0 = t[:, 0].min()
us = np.where(t > 0, (t - t0) / 100.0, np.nan)
names = {0: "start", 1: "tables in place", 2: "walk over", 3: "all placed", 31: "end"}
print("event: min / median / max over workgroups (us since first start)")
for k in (0, 1, 2, 3, 31):
    col = us[:, k]
    if np.isnan(col).all():
        continue
    print(f"{names.get(k, str(k)):18s} {np.nanmin(col):7.1f} {np.nanmedian(col):7.1f} {np.nanmax(col):7.1f}   ({int((~np.isnan(col)).sum())} wgs)")
raw = trace.cpu().numpy().reshape(256, 32).astype(np.float64)
for k, nm in ((4, "walker's own code, us"), (5, "placer's own code, us")):
    if raw[:, k].any():
        v = (raw[:, k] - 1) / 100.0
        print(f"{nm:24s} min/median/max {v.min():7.1f} {np.median(v):7.1f} {v.max():7.1f}   sum over workgroups {v.sum():9.0f}")
for k, nm in ((6, "walker visits"), (7, "placer visits")):
    if raw[:, k].any():
        v = raw[:, k] - 1
        print(f"{nm:24s} min/median/max {v.min():7.0f} {np.median(v):7.0f} {v.max():7.0f}")
if len(sys.argv) > 2:
    np.save(sys.argv[2], us)
    order = np.argsort(us[:, 31])
    print("workgroups by end time: wg, tables, walk over, end")
    for w in list(order[:5]) + list(order[-12:]):
        print(f"  {w:4d} {us[w,1]:6.1f} {us[w,2]:6.1f} {us[w,31]:6.1f}")
    print("correlation(walk over, end) =", np.corrcoef(us[:, 2], us[:, 31])[0, 1])
    print("end - walk over: min/median/max", np.min(us[:, 31] - us[:, 2]), np.median(us[:, 31] - us[:, 2]), np.max(us[:, 31] - us[:, 2]))
