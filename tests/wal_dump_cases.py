"""`dump_wal` scenarios shared by the CPU (callback seam) and GPU tests.  `dump(path, header, values)`
returns (rc, out_text, err_text).  The expected rows are built here from the write batches themselves
(row format: tools/ldb_cmd.cc:1873-1965,2012-2034) and from the CPU oracle reader for what survives
a corruption."""
import ctypes as C
import os
import struct

import oracle_lib as O

HEADER = "Sequence,Count,ByteSize,Physical Offset,Key(s)"


def varint(v):
    out = bytearray()
    while v >= 128:
        out.append((v & 127) | 128)
        v >>= 7
    out.append(v)
    return bytes(out)


def lp(b):
    return varint(len(b)) + b


def hx(b):
    return "0x" + b.hex().upper()


class Batch:
    """A write batch representation (db/write_batch.cc:10-34) and the text dump_wal prints for it."""

    def __init__(self, seq):
        self.seq, self.count, self.body, self.text, self.text_v = seq, 0, b"", "", ""

    def _op(self, body, text, text_v=None, counted=True):
        self.body += body
        self.text += text
        self.text_v += text if text_v is None else text_v
        self.count += 1 if counted else 0
        return self

    def put(self, k, v, cf=0):
        tag = bytes([1]) if cf == 0 else bytes([5]) + varint(cf)
        return self._op(tag + lp(k) + lp(v), "PUT(%d) : %s " % (cf, hx(k)), "PUT(%d) : %s : %s " % (cf, hx(k), hx(v)))

    def merge(self, k, v, cf=0):
        tag = bytes([2]) if cf == 0 else bytes([6]) + varint(cf)
        return self._op(tag + lp(k) + lp(v), "MERGE(%d) : %s " % (cf, hx(k)),
                        "MERGE(%d) : %s : %s " % (cf, hx(k), hx(v)))

    def delete(self, k, cf=0):
        tag = bytes([0]) if cf == 0 else bytes([4]) + varint(cf)
        return self._op(tag + lp(k), "DELETE(%d) : %s " % (cf, hx(k)))

    def single_delete(self, k, cf=0):
        tag = bytes([7]) if cf == 0 else bytes([8]) + varint(cf)
        return self._op(tag + lp(k), "SINGLE_DELETE(%d) : %s " % (cf, hx(k)))

    def delete_range(self, a, b, cf=0):
        tag = bytes([15]) if cf == 0 else bytes([14]) + varint(cf)
        return self._op(tag + lp(a) + lp(b), "DELETE_RANGE(%d) : %s %s " % (cf, hx(a), hx(b)))

    def log_data(self, blob):
        return self._op(bytes([3]) + lp(blob), "", counted=False)

    def noop(self):
        return self._op(bytes([13]), "NOOP ", counted=False)

    def prepare(self, xid):
        self._op(bytes([9]), "BEGIN_PREPARE(false) ", counted=False)
        return self

    def end_prepare(self, xid):
        return self._op(bytes([10]) + lp(xid), "END_PREPARE(%s) " % hx(xid), counted=False)

    def commit(self, xid):
        return self._op(bytes([11]) + lp(xid), "COMMIT(%s) " % hx(xid), counted=False)

    def rollback(self, xid):
        return self._op(bytes([12]) + lp(xid), "ROLLBACK(%s) " % hx(xid), counted=False)

    def rep(self):
        return struct.pack("<QI", self.seq, self.count) + self.body

    def row(self, offset, values):
        r = self.rep()
        return "%d,%d,%d,%d,%s\n" % (self.seq, self.count, len(r), offset, self.text_v if values else self.text)


def _offsets_by_oracle(img, log_number, mode=0):
    """(record bytes, LastRecordOffset) for everything the CPU oracle reader returns from img."""
    L = O.lib()
    t = O.OracleLog(log_number=log_number)
    L.oracle_wal_writer_free(t.writer)
    L.oracle_sink_append_raw(t.sink, bytes(img), len(img))
    t.writer = L.oracle_wal_writer_new(t.sink, log_number, 0)
    got = []
    while True:
        r = t.read(mode)
        if r == "EOF":
            break
        got.append((r, L.oracle_wal_reader_last_record_offset(t.reader)))
    dropped, report = t.dropped_bytes(), t.report_message()
    t.close()
    return got, dropped, report


def _batches():
    out = []
    seq = 100
    big = bytes(range(256)) * 300  # a value that spans several 32 KiB blocks
    for i in range(40):
        b = Batch(seq)
        if i % 8 == 0:
            b.put(b"k%d" % i, b"v" * (i + 1)).delete(b"gone").merge(b"m", b"+1", cf=3)
        elif i % 8 == 1:
            b.put(b"big%d" % i, big, cf=2).single_delete(b"once").delete_range(b"a", b"c")
        elif i % 8 == 2:
            b.prepare(b"xid1").put(b"p", b"q").end_prepare(b"xid1")
        elif i % 8 == 3:
            b.commit(b"xid1").noop().log_data(b"blob").rollback(b"xid2")
        elif i % 8 == 4:
            b.delete(b"d", cf=7).single_delete(b"s", cf=7).delete_range(b"x", b"z", cf=7)
        else:
            b.put(b"key%04d" % i, os.urandom(0) + b"value%d" % i)
        seq += max(1, b.count)
        out.append(b)
    return out


def _write_log(path, batches, log_number, recycle=False):
    t = O.OracleLog(recycle=recycle, log_number=log_number)
    for b in batches:
        t.write(b.rep())
    img = t.contents()
    t.close()
    open(path, "wb").write(img)
    return img


def clean_log(dump, tmp):
    for recycle, name, num in ((False, "000007.log", 7), (True, "000123.log", 123)):
        path = os.path.join(tmp, name)
        batches = _batches()
        img = _write_log(path, batches, num, recycle)
        recs, dropped, _ = _offsets_by_oracle(img, num)
        assert [r for r, _ in recs] == [b.rep() for b in batches] and dropped == 0
        for header in (False, True):
            for values in (False, True):
                rc, out, err = dump(path, header, values)
                want = ""
                if header:
                    want += HEADER + (" : value " if values else "") + "\n"
                want += "".join(b.row(off, values) for b, (_, off) in zip(batches, recs))
                assert rc == 0 and err == ""
                assert out == want


def corrupted_log(dump, tmp):
    path = os.path.join(tmp, "000009.log")
    batches = _batches()
    img = bytearray(_write_log(path, batches, 9))
    img[7 + 20] ^= 0x10            # inside the first record's payload
    img[32768 * 2 + 100] ^= 0x01   # inside the big value that crosses blocks
    open(path, "wb").write(img)
    recs, dropped, report = _offsets_by_oracle(bytes(img), 9)
    assert dropped > 0 and len(recs) < len(batches)
    by_rep = {b.rep(): b for b in batches}
    rc, out, err = dump(path, True, False)
    want = HEADER + "\n" + "".join(by_rep[r].row(off, False) for r, off in recs)
    assert rc == 0
    assert out == want
    lines = err.splitlines()
    assert lines and all(l.startswith("Corruption detected in log file Corruption: ") for l in lines)
    assert "checksum mismatch" in err
    # a record that is too small to be a write batch is reported, not printed
    path2 = os.path.join(tmp, "000010.log")
    t = O.OracleLog(log_number=10)
    t.write(b"tiny")
    t.write(batches[5].rep())
    open(path2, "wb").write(t.contents())
    t.close()
    rc, out, err = dump(path2, False, False)
    assert rc == 0 and out == batches[5].row(11, False)
    assert err == "Corruption detected in log file Corruption: log record too small\n"


def manifest_named_file(dump, tmp):
    """The reference's own MANIFEST-CORRECT parses as a log; its records are version edits, so only the
    row prefix (sequence, count, size, offset read as a write batch header) is checked."""
    here = os.path.dirname(os.path.abspath(__file__))
    src = open(os.path.join(here, "golden", "reference_MANIFEST-CORRECT.bin"), "rb").read()
    path = os.path.join(tmp, "MANIFEST-000005")
    open(path, "wb").write(src)
    recs, dropped, _ = _offsets_by_oracle(src, 5)
    rc, out, err = dump(path, False, False)
    assert rc == 0 and dropped == 0
    rows = out.splitlines()
    small = [r for r, _ in recs if len(r) < 12]
    assert len(rows) == len(recs) - len(small)
    k = 0
    for r, off in recs:
        if len(r) < 12:
            continue
        seq, cnt = struct.unpack_from("<QI", r)
        assert rows[k].startswith("%d,%d,%d,%d," % (seq, cnt, len(r), off))
        k += 1
    assert err.count("log record too small") == len(small)
    # missing file
    rc, out, err = dump(os.path.join(tmp, "nope.log"), False, False)
    assert rc == -1 and "Failed to open WAL file" in err


ALL = [clean_log, corrupted_log, manifest_named_file]
