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
    from terarkdb_amd.synth import splitmix_dev as gen
    return gen(nbytes, state)


def oracle_desc_batch(buf, offs, lens, inits, suffix=None, masked=False):
    return O.batch_extend(buf, offs, lens, inits, suffix, masked)
