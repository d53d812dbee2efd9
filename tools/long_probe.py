"""Development aid: long-buffer call timings (host call to completion), per segment length."""
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
segs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
for sz in (256 * 1024, 1024 * 1024, 4 * 1024 * 1024):
    for n in (1, 16, 64):
        d = (capi.Desc * n)()
        for i in range(n):
            d[i].offset, d[i].len, d[i].init = i * sz, sz, 0
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
        want = O.batch_extend(S, np.arange(n, dtype=np.uint64) * sz, np.full(n, sz, np.uint32))
        for seg in segs:
            for fn, name in ((L.tcrc_dev_extend_long, "sync"), (L.tcrc_dev_extend_long_noalloc, "noalloc")):
                if name == "noalloc" and n > 16:
                    continue
                def run():
                    capi.check(fn(eng.ctx, C.c_void_p(dS.data_ptr()), d, n, seg, C.c_void_p(out.data_ptr()), stream))
                    if name == "noalloc":
                        torch.cuda.synchronize()
                for _ in range(5): run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50): run()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 50
                ok = (out.cpu().numpy().view(np.uint32) == want).all()
                print(f"{n:3d} x {sz:8d} seg {seg:6d} {name:8s}: {dt*1e6:7.1f} us  {n*sz/dt/1e9:8.1f} GB/s  ok={ok}", flush=True)
