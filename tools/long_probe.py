"""Development aid: long-buffer call timings."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import terarkdb_amd.capi as capi
L = capi.lib
eng = capi.Engine(0)
S = O.splitmix(64 * 4 * 1024 * 1024, 301)
dS = torch.from_numpy(S).cuda()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for sz in (256 * 1024, 1024 * 1024, 4 * 1024 * 1024):
    for n in (1, 64):
        d = (capi.Desc * n)()
        for i in range(n):
            d[i].offset, d[i].len, d[i].init = i * sz, sz, 0
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
        def run():
            capi.check(L.tcrc_dev_extend_long(eng.ctx, C.c_void_p(dS.data_ptr()), d, n, 0, C.c_void_p(out.data_ptr()), stream))
        for _ in range(3): run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        want = O.batch_extend(S, np.arange(n, dtype=np.uint64) * sz, np.full(n, sz, np.uint32))
        ok = (out.cpu().numpy().view(np.uint32) == want).all()
        print(f"{n} x {sz}: {dt*1e6:.1f} us  {n*sz/dt/1e9:.1f} GB/s  ok={ok}", flush=True)
