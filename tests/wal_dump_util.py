"""Drivers for dump_wal: through the engine, or through the callback seam with the oracle scanning."""
import ctypes as C

import terarkdb_amd.capi as capi


def _call(fn, head, path, header, values):
    cap = 64 << 20
    out = C.create_string_buffer(cap)
    err = C.create_string_buffer(1 << 16)
    on, en = C.c_size_t(0), C.c_size_t(0)
    rc = fn(*head, path.encode(), int(header), int(values), out, cap, C.byref(on), err, 1 << 16, C.byref(en))
    assert on.value < cap
    return rc, out.value.decode(), err.value.decode()


def dump_with_oracle(path, header, values):
    import engine_log
    return _call(capi.lib.tcrc_shim_dump_wal_cb, (engine_log._oracle_scan, None), path, header, values)


def dump_with_engine(eng):
    def run(path, header, values):
        return _call(capi.lib.tcrc_shim_dump_wal, (eng.ctx,), path, header, values)
    return run
