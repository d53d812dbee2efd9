"""Development aid: 64 x 4 MiB through tcrc_dev_extend_long_enqueue with the segment length forced."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import terarkdb_amd.capi as capi
eng = capi.Engine(0)
L = capi.lib
S = O.splitmix(256 << 20, 301)
dS = torch.from_numpy(S).cuda()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
n, sz = 64, 4 << 20
d = (capi.Desc * n)()
for i in range(n):
    d[i].offset, d[i].len, d[i].init = i * sz, sz, 0
out = torch.zeros(n, dtype=torch.int32, device="cuda")
want = O.batch_extend(S, np.arange(n, dtype=np.uint64) * sz, np.full(n, sz, np.uint32))
for seg in (0, 2048, 4096, 8192, 16384, 32768):
    run = lambda: capi.check(L.tcrc_dev_extend_long_enqueue(eng.ctx, C.c_void_p(dS.data_ptr()), d, n, seg, C.c_void_p(out.data_ptr()), stream))
    out.zero_(); run(); torch.cuda.synchronize()
    assert (out.cpu().numpy().view(np.uint32) == want).all(), seg
    for _ in range(3000):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print(f"segment_len {seg:6d}: {us:7.2f} us per call = {n*sz/us/1e3:7.1f} GB/s = {n*sz/us/1e3/8000:.3f} of peak", flush=True)
eng.close()
