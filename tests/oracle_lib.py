"""ctypes view of oracle/_ref/liboracle.so (the CPU checker; test infrastructure only)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "_ref", "liboracle.so")
_lib = None

u8p = C.POINTER(C.c_uint8)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_SO)
        sz, u32, u64, vp, i = C.c_size_t, C.c_uint32, C.c_uint64, C.c_void_p, C.c_int
        sig = {
            "oracle_crc32c_extend_bitwise": (u32, [u32, vp, sz]),
            "oracle_crc32c_extend": (u32, [u32, vp, sz]),
            "oracle_crc32c_extend_hw": (u32, [u32, vp, sz]),
            "oracle_crc32c_value": (u32, [vp, sz]),
            "oracle_crc32c_mask": (u32, [u32]),
            "oracle_crc32c_unmask": (u32, [u32]),
            "oracle_crc32c_hw_available": (i, []),
            "oracle_sst_trailer_crc": (u32, [vp, sz, C.c_uint8, i]),
            "oracle_sst_trailer_batch": (None, [vp, sz, sz, sz, C.c_uint8, i, vp]),
            "oracle_sst_check_block": (i, [vp, sz, C.POINTER(u32), C.POINTER(u32)]),
            "oracle_batch_extend": (None, [vp, vp, vp, vp, vp, i, sz, vp]),
            "oracle_fill_splitmix64": (None, [vp, sz, u64]),
            "oracle_fill_fnv_chain": (None, [vp, sz]),
            "oracle_sink_new": (vp, []),
            "oracle_sink_free": (None, [vp]),
            "oracle_sink_data": (vp, [vp]),
            "oracle_sink_size": (sz, [vp]),
            "oracle_sink_drop": (None, [vp, sz]),
            "oracle_sink_rewind_for_overwrite": (None, [vp]),
            "oracle_sink_append_raw": (None, [vp, C.c_char_p, sz]),
            "oracle_wal_writer_new": (vp, [vp, u64, i]),
            "oracle_wal_writer_add_records_fixed": (i, [vp, vp, sz, sz]),
            "oracle_wal_writer_free": (None, [vp]),
            "oracle_wal_writer_use_hw": (None, [vp, i]),
            "oracle_wal_writer_type_crc": (u32, [vp, i]),
            "oracle_wal_writer_physical_records": (u64, [vp]),
            "oracle_wal_writer_crc_bytes": (u64, [vp]),
            "oracle_wal_writer_add_record": (i, [vp, vp, sz]),
            "oracle_source_new": (vp, [vp]),
            "oracle_source_free": (None, [vp]),
            "oracle_source_force_error": (None, [vp, sz]),
            "oracle_source_force_eof": (None, [vp, sz]),
            "oracle_source_clear_partial": (None, [vp]),
            "oracle_source_read_after_partial": (i, [vp]),
            "oracle_wal_reader_new": (vp, [vp, i, u64, i]),
            "oracle_wal_reader_free": (None, [vp]),
            "oracle_wal_reader_use_hw": (None, [vp, i]),
            "oracle_wal_reader_dropped_bytes": (sz, [vp]),
            "oracle_wal_reader_report": (C.c_char_p, [vp]),
            "oracle_wal_reader_is_eof": (i, [vp]),
            "oracle_wal_reader_last_record_offset": (u64, [vp]),
            "oracle_wal_reader_physical_records": (u64, [vp]),
            "oracle_wal_reader_crc_bytes": (u64, [vp]),
            "oracle_wal_reader_unmark_eof": (None, [vp]),
            "oracle_wal_reader_read_record": (i, [vp, i, C.POINTER(vp), C.POINTER(sz)]),
            "oracle_wal_list_physical": (sz, [vp, sz, u64, vp, sz]),
            "oracle_wal_scan_blocks": (None, [vp, sz, u64, vp]),
            "oracle_wal_write_dbbench_records": (sz, [vp, sz, u32]),
            "oracle_wal_read_all": (sz, [vp, i, C.POINTER(u64)]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if isinstance(a, np.ndarray) else a


def extend(init, data, variant="table"):
    """Extend(init, data) with the chosen oracle variant: table | bitwise | hw."""
    if isinstance(data, (bytes, bytearray)):
        arr = np.frombuffer(bytes(data), dtype=np.uint8)
    else:
        arr = np.ascontiguousarray(data, dtype=np.uint8)
    fn = {"table": lib().oracle_crc32c_extend, "bitwise": lib().oracle_crc32c_extend_bitwise,
          "hw": lib().oracle_crc32c_extend_hw}[variant]
    return fn(init, _ptr(arr) if arr.size else None, arr.size)


def value(data, variant="table"):
    return extend(0, data, variant)


def mask(c):
    return lib().oracle_crc32c_mask(c)


def unmask(c):
    return lib().oracle_crc32c_unmask(c)


def splitmix(nbytes, state=301):
    assert nbytes % 8 == 0
    a = np.empty(nbytes, dtype=np.uint8)
    lib().oracle_fill_splitmix64(_ptr(a), nbytes, state)
    return a


def fnv_chain(nbytes=4 * 1024 * 1024):
    a = np.empty(nbytes, dtype=np.uint8)
    lib().oracle_fill_fnv_chain(_ptr(a), nbytes)
    return a


def sst_trailer_batch(buf, block_len, stride, n_blocks, type_byte=0, hw=False):
    out = np.empty(n_blocks, dtype=np.uint32)
    lib().oracle_sst_trailer_batch(_ptr(buf), block_len, stride, n_blocks, type_byte, int(hw), _ptr(out))
    return out


def batch_extend(buf, off, length, init=None, suffix=None, masked=False):
    off = np.ascontiguousarray(off, dtype=np.uint64)
    length = np.ascontiguousarray(length, dtype=np.uint32)
    n = off.size
    out = np.empty(n, dtype=np.uint32)
    init_a = None if init is None else np.ascontiguousarray(init, dtype=np.uint32)
    suf_a = None if suffix is None else np.ascontiguousarray(suffix, dtype=np.uint8)
    lib().oracle_batch_extend(_ptr(buf), _ptr(off), _ptr(length), None if init_a is None else _ptr(init_a),
                              None if suf_a is None else _ptr(suf_a), int(masked), n, _ptr(out))
    return out


class Random:
    """util/random.h:22-75 restated (Park-Miller, seed 301 in the reference's log test)."""

    def __init__(self, s):
        self.seed = (s & 2147483647) or 1

    def next(self):
        product = self.seed * 16807
        self.seed = (product >> 31) + (product & 2147483647)
        if self.seed > 2147483647:
            self.seed -= 2147483647
        return self.seed

    def uniform(self, n):
        return self.next() % n

    def skewed(self, max_log):
        return self.uniform(1 << self.uniform(max_log + 1))


def big_string(partial, n):
    """db/log_test.cc:26-33"""
    if n == 0:
        return b""
    reps = (n + len(partial) - 1) // len(partial)
    return (partial * reps)[:n]


def number_string(i):
    return b"%d." % i


class OracleLog:
    """Writer + reader pair over one in-memory file, the fixture shape of db/log_test.cc:47-260."""

    def __init__(self, recycle=False, log_number=123, checksum=True, retry_after_eof=False, hw=False):
        L = lib()
        self.recycle = bool(recycle)
        self.header_size = 11 if recycle else 7
        self.sink = L.oracle_sink_new()
        self.writer = L.oracle_wal_writer_new(self.sink, log_number, int(bool(recycle)))
        self.source = L.oracle_source_new(self.sink)
        self.reader = L.oracle_wal_reader_new(self.source, int(checksum), log_number, int(retry_after_eof))
        if hw:
            L.oracle_wal_writer_use_hw(self.writer, 1)
            L.oracle_wal_reader_use_hw(self.reader, 1)

    def close(self):
        L = lib()
        L.oracle_wal_reader_free(self.reader)
        L.oracle_source_free(self.source)
        L.oracle_wal_writer_free(self.writer)
        L.oracle_sink_free(self.sink)

    def write(self, msg):
        buf = (C.c_uint8 * max(1, len(msg))).from_buffer_copy(msg if msg else b"\0")
        lib().oracle_wal_writer_add_record(self.writer, buf, len(msg))

    def written_bytes(self):
        return lib().oracle_sink_size(self.sink)

    def contents(self):
        n = self.written_bytes()
        return C.string_at(lib().oracle_sink_data(self.sink), n) if n else b""

    def _bytes(self):
        n = self.written_bytes()
        return (C.c_uint8 * n).from_address(lib().oracle_sink_data(self.sink))

    def read(self, mode=0):
        rec = C.c_void_p()
        n = C.c_size_t()
        ok = lib().oracle_wal_reader_read_record(self.reader, mode, C.byref(rec), C.byref(n))
        if not ok:
            return "EOF"
        return C.string_at(rec, n.value) if n.value else b""

    def increment_byte(self, offset, delta):
        b = self._bytes()
        b[offset] = (b[offset] + delta) & 0xff

    def set_byte(self, offset, v):
        self._bytes()[offset] = v & 0xff

    def shrink_size(self, n):
        lib().oracle_sink_drop(self.sink, n)

    def fix_checksum(self, header_offset, length, recyclable):
        hs = 11 if recyclable else 7
        img = self.contents()
        crc = mask(value(img[header_offset + 6: header_offset + hs + length]))
        b = self._bytes()
        for k in range(4):
            b[header_offset + k] = (crc >> (8 * k)) & 0xff

    def force_error(self, position=0):
        lib().oracle_source_force_error(self.source, position)

    def force_eof(self, position=0):
        lib().oracle_source_force_eof(self.source, position)

    def unmark_eof(self):
        lib().oracle_source_clear_partial(self.source)
        lib().oracle_wal_reader_unmark_eof(self.reader)

    def is_eof(self):
        return bool(lib().oracle_wal_reader_is_eof(self.reader))

    def dropped_bytes(self):
        return lib().oracle_wal_reader_dropped_bytes(self.reader)

    def report_message(self):
        return lib().oracle_wal_reader_report(self.reader).decode()

    def match_error(self, msg):
        m = self.report_message()
        return "OK" if msg in m else m

    def overwriting_writer(self):
        """A second writer over the same file from offset 0 (db/log_test.cc:632-635)."""
        L = lib()
        L.oracle_sink_rewind_for_overwrite(self.sink)
        outer = self

        class _W:
            def __init__(self):
                self.w = L.oracle_wal_writer_new(outer.sink, 123, 1)

            def write(self, msg):
                buf = (C.c_uint8 * max(1, len(msg))).from_buffer_copy(msg if msg else b"\0")
                L.oracle_wal_writer_add_record(self.w, buf, len(msg))

        return _W()


def list_physical(img, log_number):
    arr = np.frombuffer(img, dtype=np.uint8)
    cap = max(16, len(img) // 7 + 1)
    cap = min(cap, 4_000_000)
    out = np.zeros(cap * 4, dtype=np.uint64)
    n = lib().oracle_wal_list_physical(_ptr(arr), arr.size, log_number, _ptr(out), cap)
    return out[: 4 * min(n, cap)].reshape(-1, 4), n


def wal_scan_blocks(img, log_number):
    """Per 32 KiB block: (n_records, first_bad, end_offset, end_code); see oracle/wal_oracle.c."""
    arr = np.frombuffer(img, dtype=np.uint8) if not isinstance(img, np.ndarray) else img
    nb = (arr.size + 32767) // 32768
    out = np.zeros((nb, 4), dtype=np.uint32)
    if nb:
        lib().oracle_wal_scan_blocks(_ptr(arr), arr.size, log_number, _ptr(out))
    return out


def write_edge_records(t, recycle):
    """Appends to the OracleLog t records placed to produce every trailer length 0..10 and the fragment edge
    sizes (db/log_test.cc:293-351)."""
    hs = 11 if recycle else 7
    for sz in (0, 1, 32768 - hs, 32768 - hs + 1, 65536, 100000, 32761, 32762):
        t.write(big_string(b"e%d." % sz, sz))
    for trailer in range(0, 11):
        # pad the current block so that exactly `trailer` bytes stay free after a record
        used = t.written_bytes() % 32768
        room = 32768 - used
        body = room - hs - trailer
        if body < 0:
            t.write(big_string(b"p", room + 100))
            used = t.written_bytes() % 32768
            body = 32768 - used - hs - trailer
        t.write(big_string(b"t%d." % trailer, body))
        t.write(b"after-trailer")


def build_wal_image(recycle, log_number=123, n_random=500, seed=301, edge=True):
    """A log image like db/log_test.cc's RandomRead (Random(seed).Skewed(17) sizes) plus
    records placed to produce every trailer length 0..10 and the fragment edge sizes."""
    t = OracleLog(recycle=recycle, log_number=log_number)
    rnd = Random(seed)
    if edge:
        write_edge_records(t, recycle)
    for i in range(n_random):
        t.write(big_string(number_string(i), rnd.skewed(17)))
    img = t.contents()
    t.close()
    return img
