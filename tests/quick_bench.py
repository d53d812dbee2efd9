"""Ad-hoc timing of the uniform-block kernel (development aid, not the judged bench)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import gpu_util as G
import terarkdb_amd.capi as capi


def main():
    n, blen = 131072, 8192
    dev = G.splitmix_dev(n * blen)
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    for lanes in (8, 16, 64):
        e = G.engine(lanes)
        for _ in range(3):
            e.blocks_stamp(dev, n, blen, blen, out, suffix_byte=0)
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        iters = 20
        for _ in range(iters):
            e.blocks_stamp(dev, n, blen, blen, out, suffix_byte=0)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / iters
        x = int(np.bitwise_xor.reduce(out.cpu().numpy().view(np.uint32)))
        print(f"lanes={lanes} {ms:.3f} ms/launch  {n * (blen + 1) / ms / 1e6:.1f} GB/s  xor={x:08x} (want 0744a84e)", flush=True)
        e.close()


if __name__ == "__main__":
    main()
