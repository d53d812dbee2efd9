"""The writer/reader fixture of db/log_test.cc over the engine's own log::Writer / log::Reader
(through include/tcrc_shim.h).  Same interface as oracle_lib.OracleLog, so wal_scenarios.py
runs over both.

ctx given  -> checksums come from the GPU (the shipped configuration).
ctx = None -> the two checksum seams are served by callbacks into the CPU oracle, which lets
              the host-side framing/reassembly/policy logic be tested on a machine without a GPU.
"""
import ctypes as C

import numpy as np

import oracle_lib as O
import terarkdb_amd.capi as capi

L = capi.lib


@capi.STAMP_FN
def _oracle_stamp(user, base, base_bytes, spans, n, out):
    for i in range(n):
        d = spans[i]
        out[i] = O.lib().oracle_crc32c_mask(O.lib().oracle_crc32c_extend(d.init, base + d.offset, d.len))
    return 0


@capi.SCAN_FN
def _oracle_scan(user, img, n, log_number, blocks):
    nb = (n + 32767) // 32768
    tmp = np.zeros((nb, 4), dtype=np.uint32)
    O.lib().oracle_wal_scan_blocks(img, n, log_number, tmp.ctypes.data)
    for b in range(nb):
        blocks[b].n_records, blocks[b].first_bad, blocks[b].end_offset, blocks[b].end_code = (int(x) for x in tmp[b])
    return 0


class EngineLog:
    def __init__(self, recycle=False, log_number=123, checksum=True, retry_after_eof=False, ctx=None,
                 manual_flush=False, bulk_image=False):
        self.recycle = bool(recycle)
        self.ctx = ctx
        self.log_number = log_number
        self.manual_flush = manual_flush
        self.bulk_image = bulk_image
        self.file = L.tcrc_shim_memfile_new()
        self.writer = self._new_writer(overwrite=False)
        if ctx is not None:
            self.reader = L.tcrc_shim_wal_reader_new(ctx, self.file, int(checksum), log_number, int(retry_after_eof))
        else:
            self.reader = L.tcrc_shim_wal_reader_new_cb(_oracle_scan, None, self.file, int(checksum), log_number,
                                                        int(retry_after_eof))
        self._attached = False
        self._extra_writers = []

    def _new_writer(self, overwrite):
        if self.ctx is not None:
            return L.tcrc_shim_wal_writer_new(self.ctx, self.file, self.log_number, int(self.recycle),
                                              int(self.manual_flush), int(overwrite))
        return L.tcrc_shim_wal_writer_new_cb(_oracle_stamp, None, self.file, self.log_number, int(self.recycle),
                                             int(self.manual_flush), int(overwrite))

    def close(self):
        for w in self._extra_writers:
            L.tcrc_shim_wal_writer_free(w)
        L.tcrc_shim_wal_reader_free(self.reader)
        L.tcrc_shim_wal_writer_free(self.writer)
        L.tcrc_shim_memfile_free(self.file)

    # -- writer side
    def write(self, msg, writer=None):
        buf = (C.c_uint8 * max(1, len(msg))).from_buffer_copy(msg if msg else b"\0")
        rc = L.tcrc_shim_wal_writer_add_record(writer or self.writer, buf, len(msg))
        assert rc == 0, L.tcrc_shim_wal_writer_status(writer or self.writer)

    def flush(self):
        assert L.tcrc_shim_wal_writer_write_buffer(self.writer) == 0

    def written_bytes(self):
        return L.tcrc_shim_memfile_size(self.file)

    def contents(self):
        n = self.written_bytes()
        return C.string_at(L.tcrc_shim_memfile_data(self.file), n) if n else b""

    def _bytes(self):
        n = self.written_bytes()
        return (C.c_uint8 * n).from_address(L.tcrc_shim_memfile_data(self.file))

    def increment_byte(self, offset, delta):
        b = self._bytes()
        b[offset] = (b[offset] + delta) & 0xff

    def set_byte(self, offset, v):
        self._bytes()[offset] = v & 0xff

    def shrink_size(self, n):
        L.tcrc_shim_memfile_drop(self.file, n)

    def fix_checksum(self, header_offset, length, recyclable):
        # db/log_test.cc:211-218 (a test helper: the CPU oracle recomputes the stored value)
        hs = 11 if recyclable else 7
        img = self.contents()
        crc = O.mask(O.value(img[header_offset + 6: header_offset + hs + length]))
        b = self._bytes()
        for k in range(4):
            b[header_offset + k] = (crc >> (8 * k)) & 0xff

    def overwriting_writer(self):
        w = self._new_writer(overwrite=True)
        self._extra_writers.append(w)
        outer = self

        class _W:
            def write(self, msg):
                outer.write(msg, writer=w)

        return _W()

    # -- reader side
    def read(self, mode=0):
        if self.bulk_image and not self._attached:
            # recovery shape: the whole file is verified in one pass before the first record is read
            assert L.tcrc_shim_wal_reader_attach_image(self.reader, None, 0) == 0
            self._attached = True
        rec = C.c_void_p()
        n = C.c_size_t()
        ok = L.tcrc_shim_wal_reader_read_record(self.reader, mode, C.byref(rec), C.byref(n))
        if not ok:
            return "EOF"
        return C.string_at(rec, n.value) if n.value else b""

    def force_error(self, position=0):
        L.tcrc_shim_wal_reader_force_error(self.reader, position)

    def force_eof(self, position=0):
        L.tcrc_shim_wal_reader_force_eof(self.reader, position)

    def unmark_eof(self):
        L.tcrc_shim_wal_reader_unmark_eof(self.reader)

    def is_eof(self):
        return bool(L.tcrc_shim_wal_reader_is_eof(self.reader))

    def dropped_bytes(self):
        return L.tcrc_shim_wal_reader_dropped_bytes(self.reader)

    def report_message(self):
        return L.tcrc_shim_wal_reader_report(self.reader).decode()

    def match_error(self, msg):
        m = self.report_message()
        return "OK" if msg in m else m

    def verifier_calls(self):
        return L.tcrc_shim_wal_reader_verifier_calls(self.reader)
