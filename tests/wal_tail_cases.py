"""Tailing a log file that is still being written (db/log_test.cc:743-813, RetriableLogTest): the record's
bytes reach the file in two parts, the reader (retry_after_eof) is already waiting inside ReadRecord when
the second part arrives, and must return the whole record.  `open_reader(path, log_number)` returns a
reader handle of the shim; the writer side is the CPU oracle."""
import ctypes as C
import threading
import time

import oracle_lib as O
import terarkdb_amd.capi as capi


def _read(reader):
    rec, n = C.c_void_p(), C.c_size_t()
    ok = capi.lib.tcrc_shim_wal_reader_read_record(reader, 0, C.byref(rec), C.byref(n))
    return C.string_at(rec, n.value) if ok else "Read error"


def _tail(open_reader, tmp, recycle, delta_of_header, payload=b"foo", earlier=()):
    hs = 11 if recycle else 7
    t = O.OracleLog(recycle=recycle, log_number=123)
    for e in earlier:
        t.write(e)
    old = t.written_bytes()
    t.write(payload)
    img = t.contents()
    t.close()
    delta = hs + delta_of_header
    path = "%s/log_%d_%d_%d" % (tmp, int(recycle), delta_of_header, len(earlier))
    with open(path, "wb") as f:
        f.write(img[:old + delta])       # part 1: ends inside the header (or just past it)
        f.flush()
        reader = open_reader(path, 123)
        assert reader
        got = []
        for e in earlier:                 # complete records are there already
            assert _read(reader) == e
        th = threading.Thread(target=lambda: got.append(_read(reader)))
        th.start()
        time.sleep(0.3)                   # the reader is polling at the end of the file by now
        assert th.is_alive() and not got
        f.write(img[old + delta:])       # part 2
        f.flush()
        th.join(timeout=20)
        assert not th.is_alive(), "reader did not pick up the rest of the record"
    assert got == [payload]
    assert capi.lib.tcrc_shim_wal_reader_dropped_bytes(reader) == 0
    capi.lib.tcrc_shim_wal_reader_free(reader)


def partial_header(open_reader, tmp, recycle):  # :743-777
    _tail(open_reader, tmp, recycle, -1)


def full_header(open_reader, tmp, recycle):  # :779-813
    _tail(open_reader, tmp, recycle, +1)


def later_in_the_file(open_reader, tmp, recycle):
    """The same with earlier records in the file and a payload that spans blocks."""
    earlier = [b"a" * 10, b"b" * 40000, b""]
    _tail(open_reader, tmp, recycle, -1, payload=b"q" * 70000, earlier=earlier)
    _tail(open_reader, tmp, recycle, +1 + 35000, payload=b"r" * 70000, earlier=earlier)


ALL = [partial_header, full_header, later_in_the_file]
