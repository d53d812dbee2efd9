"""Development aid: GPU clock and power (rocm-smi) while a kernel runs back to back: the headline block kernel
and the log scan."""
import ctypes as C, os, subprocess, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import terarkdb_amd.capi as capi
import gpu_util as G
OL = O.lib()
eng = capi.Engine(0)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        import json
        d = json.loads(out)
        card = d[sorted(d)[0]]
        keep = {k: v for k, v in card.items() if "sclk" in k or "mclk" in k or "ower" in k}
        return keep
    except Exception as e:  # noqa
        return {"error": str(e)[:80]}


def under_load(name, fn, seconds=3.0):
    stop = False
    samples = []
    def sampler():
        while not stop:
            samples.append(smi()); time.sleep(0.3)
    th = threading.Thread(target=sampler); th.start()
    t_end = time.perf_counter() + seconds
    times = []
    while time.perf_counter() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 50 * 1e3)
    stop = True; th.join()
    print(name, "us per launch over time:", [round(t, 1) for t in times[:3]], "...", [round(t, 1) for t in times[-3:]])
    for s in samples[:2] + samples[-2:]:
        print("   ", s)


print("idle", smi())
n, blen = 131072, 8192
data = G.splitmix_dev(n * blen, 301)
out = torch.zeros(n, dtype=torch.int32, device="cuda")
under_load("blocks 8 KiB", lambda: eng.blocks_stamp(data, n, blen, blen, out, suffix_byte=0, flags=capi.F_MASK))
del data
t = O.OracleLog(recycle=False, log_number=7, hw=True)
OL.oracle_wal_write_dbbench_records(t.writer, 500000, 301)
size = t.written_bytes()
img = np.frombuffer((C.c_uint8 * size).from_address(OL.oracle_sink_data(t.sink)), dtype=np.uint8).copy()
dev = torch.from_numpy(img).cuda()
blocks = torch.zeros(((size + 32767) // 32768, 4), dtype=torch.int32, device="cuda")
counts = torch.zeros(3, dtype=torch.int64, device="cuda")
eng.counts_reset(counts)
under_load("log scan 1 GB", lambda: eng.wal_verify(dev, size, 7, blocks, counts, want_records=False))
