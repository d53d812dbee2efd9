"""Helpers shared by the GPU parity tests (device side only; the checker is oracle_lib)."""
import os

import numpy as np
import torch

import oracle_lib as O


def engine(lanes=None):
    """Engine on cuda:0; `lanes` picks the lanes-per-buffer geometry (8, 16 or 64)."""
    import terarkdb_amd.capi as capi
    if lanes is None:
        os.environ.pop("TCRC_LANES_PER_BUFFER", None)
    else:
        os.environ["TCRC_LANES_PER_BUFFER"] = str(lanes)
    try:
        return capi.Engine(0)
    finally:
        os.environ.pop("TCRC_LANES_PER_BUFFER", None)


def to_dev(a):
    a = np.ascontiguousarray(a)
    if not a.flags.writeable:
        a = a.copy()  # torch refuses to wrap read-only memory quietly
    return torch.from_numpy(a).cuda()


def splitmix_dev(nbytes, state=301):
    """The splitmix64 stream generated directly on the GPU (same bytes as oracle_lib.splitmix)."""
    assert nbytes % 8 == 0
    n = nbytes // 8
    idx = torch.arange(1, n + 1, dtype=torch.int64, device="cuda")
    golden = -7046029254386353131  # 0x9e3779b97f4a7c15 as int64
    z = idx * golden + state

    def lsr(x, k):  # logical shift right on int64
        return (x >> k) & ((1 << (64 - k)) - 1)

    z = (z ^ lsr(z, 30)) * -4658895280553007687  # 0xbf58476d1ce4e5b9
    z = (z ^ lsr(z, 27)) * -7723592293110705685  # 0x94d049bb133111eb
    z = z ^ lsr(z, 31)
    return z.view(torch.uint8)


def oracle_desc_batch(buf, offs, lens, inits, suffix=None, masked=False):
    return O.batch_extend(buf, offs, lens, inits, suffix, masked)
