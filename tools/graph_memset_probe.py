#!/usr/bin/env python3
"""Development aid: the no-allocation long-buffer call recorded into a HIP graph and replayed, with its
accumulator lines cleared by hipMemsetAsync (TCRC_DEBUG_NOALLOC_MEMSET=1, the form that gave wrong results) or by
the small kernel the library uses (default).  Prints per replay whether the outputs are right; run it under
`rocprofv3 --kernel-trace` and list the dispatches with tools/kt_list.py to see where the fill lands on replay.

usage: graph_memset_probe.py [memset|kernel] [replays]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
mode = sys.argv[1] if len(sys.argv) > 1 else "memset"
replays = int(sys.argv[2]) if len(sys.argv) > 2 else 3
os.environ["TCRC_DEBUG_NOALLOC_MEMSET"] = "1" if mode == "memset" else "0"

import numpy as np  # noqa: E402
import torch  # noqa: E402
import oracle_lib as O  # noqa: E402
import terarkdb_amd.capi as capi  # noqa: E402

eng = capi.Engine(0)
host = O.splitmix(4 << 20, 77)
dev = torch.from_numpy(host.copy()).cuda()
offs, lens, inits = [5, 1 << 20, 3 << 20], [(1 << 20) - 9, 2 << 20, 70001], [0, 0xdeadbeef, 7]
want = O.batch_extend(host, np.array(offs, np.uint64), np.array(lens, np.uint32), np.array(inits, np.uint32))
out = torch.zeros(3, dtype=torch.int32, device="cuda")
eng.extend_long(dev, offs, lens, inits, out, noalloc=True)
torch.cuda.synchronize()
print("plain call right:", bool((out.cpu().numpy().view(np.uint32) == want).all()), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    eng.extend_long(dev, offs, lens, inits, out, noalloc=True, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for rep in range(replays):
    out.zero_()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    print(f"mode {mode} replay {rep}: right={bool((got == want).all())} got={[hex(int(x)) for x in got]}", flush=True)
    lines = np.zeros(2048, dtype=np.uint32)
    capi.lib.tcrc_debug_static_lines(eng.ctx, C.c_void_p(lines.ctypes.data), C.c_size_t(lines.size))
    nz = np.nonzero(lines)[0]
    print(f"    accumulator lines afterwards: {nz.size} non-zero words of {lines.size}; first: "
          f"{[(int(i), hex(int(lines[i]))) for i in nz[:12]]}", flush=True)
eng.close()
