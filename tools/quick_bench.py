"""Ad-hoc timing of the uniform-block kernel (development aid, not the judged bench)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import gpu_util as G
import terarkdb_amd.capi as capi


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


def main():
    n, blen = 131072, 8192
    dev = G.splitmix_dev(n * blen)
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    scratch = torch.zeros(2 * 1024 * 256, dtype=torch.int32, device="cuda")
    lanes_list = [int(x) for x in os.environ.get("QB_LANES", "8,16").split(",")]
    for rep in range(2):
      for rmap in ("0",):
        for nt in ("1",):
            os.environ["TCRC_NONTEMPORAL"] = nt
            e = G.engine(8)
            for wpc in (1, 2):
                ms = timeit(lambda: e.probe_read(dev, n * blen, scratch, wpc))
                print(f"nt={nt} probe_read wgs/cu={wpc}: {ms:.4f} ms  {n * blen / ms / 1e6:.1f} GB/s", flush=True)
            e.close()
            for lanes in lanes_list:
                e = G.engine(lanes)
                ms = timeit(lambda: e.blocks_stamp(dev, n, blen, blen, out, suffix_byte=0))
                x = int(np.bitwise_xor.reduce(out.cpu().numpy().view(np.uint32)))
                print(f"nt={nt} lanes={lanes} {ms:.4f} ms/launch  {n * (blen + 1) / ms / 1e6:.1f} GB/s  "
                      f"xor={x:08x} (want 0744a84e)", flush=True)
                e.close()


if __name__ == "__main__":
    main()
