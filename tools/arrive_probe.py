"""Development aid: 64 x 4 MiB, 1 x 256 MiB and 64 x 1 MiB through tcrc_dev_extend_long_enqueue, then the batch
kernel over the same 256 MiB.  TCRC_LIBRARY names another build of the library to time; CHECK=0 skips the comparison
with the oracle (for builds that leave results out on purpose)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import terarkdb_amd.capi as capi
check = os.environ.get("CHECK", "1") == "1"
eng = capi.Engine(0)
L = capi.lib
S = O.splitmix(256 << 20, 301)
dS = torch.from_numpy(S).cuda()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for n, sz in ((64, 4 << 20), (1, 256 << 20), (64, 1 << 20)):
    d = (capi.Desc * n)()
    for i in range(n):
        d[i].offset, d[i].len, d[i].init = i * sz, sz, 0
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    run = lambda: capi.check(L.tcrc_dev_extend_long_enqueue(eng.ctx, C.c_void_p(dS.data_ptr()), d, n, 0, C.c_void_p(out.data_ptr()), stream))
    if check:
        want = O.batch_extend(S, np.arange(n, dtype=np.uint64) * sz, np.full(n, sz, np.uint32))
        for _ in range(3):
            out.zero_(); run(); torch.cuda.synchronize()
            assert (out.cpu().numpy().view(np.uint32) == want).all(), (n, sz)
    for _ in range(3000):
        run()
    torch.cuda.synchronize()
    best = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            run()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 200 * 1e3)
    us = sorted(best)[2]
    print(f"{os.environ.get('TCRC_LIBRARY','default')[-16:]:>16} {n:4d} x {sz>>20:4d} MiB: median {us:7.2f} us  min {min(best):7.2f} = {n*sz/us/1e3/8000:.3f} of peak", flush=True)
    if check:
        out.zero_(); run(); torch.cuda.synchronize()
        assert (out.cpu().numpy().view(np.uint32) == want).all(), (n, sz)
eng.close()
# the batch kernel over the same 256 MiB as 32768 blocks of 8 KiB: what a launch of this size costs without any folding
eng = capi.Engine(0)
N, B = 32768, 8192
stored = torch.zeros(N, dtype=torch.int32, device="cuda")
out = torch.zeros_like(stored)
eng.blocks_stamp(dS, N, B, B, stored, suffix_byte=-1, flags=0)
torch.cuda.synchronize()
for _ in range(3000):
    eng.blocks_stamp(dS, N, B, B, out, suffix_byte=-1, flags=0)
torch.cuda.synchronize()
best = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        eng.blocks_stamp(dS, N, B, B, out, suffix_byte=-1, flags=0)
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 200 * 1e3)
us = sorted(best)[2]
print(f"{'blocks':>16} {N} x 8 KiB: median {us:7.2f} us  min {min(best):7.2f} = {N*B/us/1e3/8000:.3f} of peak", flush=True)
eng.close()
