"""Drivers for the whole-file table verification: through the engine (GPU) or, for the CPU tests of
the host logic, through the callback seam with the oracle doing the arithmetic."""
import ctypes as C

import numpy as np

import oracle_lib as O
import terarkdb_amd.capi as capi


@capi.CHECK_FN
def _oracle_check(user, image, image_bytes, spans, n, computed, ok):
    arr = np.frombuffer((C.c_uint8 * image_bytes).from_address(image), dtype=np.uint8)
    offs = np.array([spans[i].offset for i in range(n)], dtype=np.uint64)
    lens = np.array([spans[i].len for i in range(n)], dtype=np.uint32)
    got = O.batch_extend(arr, offs, lens, None, None, True)
    for i in range(n):
        computed[i] = int(got[i])
        p = int(offs[i]) + int(lens[i])
        stored = int(arr[p]) | (int(arr[p + 1]) << 8) | (int(arr[p + 2]) << 16) | (int(arr[p + 3]) << 24)
        ok[i] = 1 if stored == int(got[i]) else 0
    return 0


def _call(fn, head, path, extent_bytes=0, read_threads=0, index_type_hint=0):
    stats = (C.c_uint64 * 7)()
    msg = C.create_string_buffer(1024)
    rc = fn(*head, path.encode(), extent_bytes, read_threads, index_type_hint, stats, msg, 1024)
    return rc, msg.value.decode(), [int(x) for x in stats]


def verify_with_oracle(path, **kw):
    return _call(capi.lib.tcrc_shim_sst_verify_cb, (_oracle_check, None), path, **kw)


def verify_with_engine(eng):
    def run(path, **kw):
        return _call(capi.lib.tcrc_shim_sst_verify, (eng.ctx,), path, **kw)
    return run
