"""The scenarios of the reference's db/log_test.cc, written once against a small
fixture interface so the same cases run over the CPU oracle and over the device
engine's writer/reader.  `mk(recycle, **kw)` returns an object with write/read/...
like OracleLog in oracle_lib.py.  Line numbers refer to db/log_test.cc."""
from oracle_lib import Random, big_string, number_string

K_BLOCK = 32768
ABSOLUTE = 1  # WALRecoveryMode::kAbsoluteConsistency


def hs(recycle):
    return 11 if recycle else 7


def random_skewed_string(i, rnd):
    return big_string(number_string(i), rnd.skewed(17))


def empty(mk, recycle):  # :268
    t = mk(recycle)
    assert t.read() == "EOF"


def read_write(mk, recycle):  # :270
    t = mk(recycle)
    for m in (b"foo", b"bar", b"", b"xxxx"):
        t.write(m)
    for m in (b"foo", b"bar", b"", b"xxxx"):
        assert t.read() == m
    assert t.read() == "EOF"
    assert t.read() == "EOF"


def many_blocks(mk, recycle, n=100000):  # :283
    t = mk(recycle)
    for i in range(n):
        t.write(number_string(i))
    for i in range(n):
        assert t.read() == number_string(i)
    assert t.read() == "EOF"


def fragmentation(mk, recycle):  # :293
    t = mk(recycle)
    t.write(b"small")
    t.write(big_string(b"medium", 50000))
    t.write(big_string(b"large", 100000))
    assert t.read() == b"small"
    assert t.read() == big_string(b"medium", 50000)
    assert t.read() == big_string(b"large", 100000)
    assert t.read() == "EOF"


def marginal_trailer(mk, recycle):  # :303
    t = mk(recycle)
    n = K_BLOCK - 2 * hs(recycle)
    t.write(big_string(b"foo", n))
    assert t.written_bytes() == K_BLOCK - hs(recycle)
    t.write(b"")
    t.write(b"bar")
    assert t.read() == big_string(b"foo", n)
    assert t.read() == b""
    assert t.read() == b"bar"
    assert t.read() == "EOF"


def marginal_trailer2(mk, recycle):  # :317
    t = mk(recycle)
    n = K_BLOCK - 2 * hs(recycle)
    t.write(big_string(b"foo", n))
    assert t.written_bytes() == K_BLOCK - hs(recycle)
    t.write(b"bar")
    assert t.read() == big_string(b"foo", n)
    assert t.read() == b"bar"
    assert t.read() == "EOF"
    assert t.dropped_bytes() == 0
    assert t.report_message() == ""


def short_trailer(mk, recycle):  # :331
    t = mk(recycle)
    n = K_BLOCK - 2 * hs(recycle) + 4
    t.write(big_string(b"foo", n))
    assert t.written_bytes() == K_BLOCK - hs(recycle) + 4
    t.write(b"")
    t.write(b"bar")
    assert t.read() == big_string(b"foo", n)
    assert t.read() == b""
    assert t.read() == b"bar"
    assert t.read() == "EOF"


def aligned_eof(mk, recycle):  # :344
    t = mk(recycle)
    n = K_BLOCK - 2 * hs(recycle) + 4
    t.write(big_string(b"foo", n))
    assert t.written_bytes() == K_BLOCK - hs(recycle) + 4
    assert t.read() == big_string(b"foo", n)
    assert t.read() == "EOF"


def random_read(mk, recycle, n=500):  # :353
    t = mk(recycle)
    w = Random(301)
    for i in range(n):
        t.write(random_skewed_string(i, w))
    r = Random(301)
    for i in range(n):
        assert t.read() == random_skewed_string(i, r)
    assert t.read() == "EOF"
    return t


def read_error(mk, recycle):  # :368
    t = mk(recycle)
    t.write(b"foo")
    t.force_error()
    assert t.read() == "EOF"
    assert t.dropped_bytes() == K_BLOCK
    assert t.match_error("read error") == "OK"


def bad_record_type(mk, recycle):  # :376
    t = mk(recycle)
    t.write(b"foo")
    t.increment_byte(6, 100)
    t.fix_checksum(0, 3, False)
    assert t.read() == "EOF"
    assert t.dropped_bytes() == 3
    assert t.match_error("unknown record type") == "OK"


def truncated_trailing_record_is_ignored(mk, recycle):  # :386
    t = mk(recycle)
    t.write(b"foo")
    t.shrink_size(4)
    assert t.read() == "EOF"
    assert t.dropped_bytes() == 0
    assert t.report_message() == ""


def truncated_trailing_record_is_not_ignored(mk, recycle):  # :395
    t = mk(recycle)
    t.write(b"foo")
    t.shrink_size(4)
    assert t.read(ABSOLUTE) == "EOF"
    assert t.dropped_bytes() > 0
    assert t.match_error("Corruption: truncated header") == "OK"


def bad_length(mk, recycle):  # :404
    t = mk(recycle)
    payload = K_BLOCK - hs(recycle)
    t.write(big_string(b"bar", payload))
    t.write(b"foo")
    t.increment_byte(4, 1)
    if not recycle:
        assert t.read() == b"foo"
        assert t.dropped_bytes() == K_BLOCK
        assert t.match_error("bad record length") == "OK"
    else:
        assert t.read() == "EOF"


def bad_length_at_end_is_ignored(mk, recycle):  # :420
    t = mk(recycle)
    t.write(b"foo")
    t.shrink_size(1)
    assert t.read() == "EOF"
    assert t.dropped_bytes() == 0
    assert t.report_message() == ""


def bad_length_at_end_is_not_ignored(mk, recycle):  # :428
    t = mk(recycle)
    t.write(b"foo")
    t.shrink_size(1)
    assert t.read(ABSOLUTE) == "EOF"
    assert t.dropped_bytes() > 0
    assert t.match_error("Corruption: truncated header") == "OK"


def checksum_mismatch(mk, recycle):  # :436
    t = mk(recycle)
    t.write(b"foooooo")
    t.increment_byte(0, 14)
    assert t.read() == "EOF"
    if not recycle:
        assert t.dropped_bytes() == 14
        assert t.match_error("checksum mismatch") == "OK"
    else:
        assert t.dropped_bytes() == 0
        assert t.report_message() == ""


def unexpected_middle_type(mk, recycle):  # :449
    t = mk(recycle)
    t.write(b"foo")
    t.set_byte(6, 7 if recycle else 3)
    t.fix_checksum(0, 3, bool(recycle))
    assert t.read() == "EOF"
    assert t.dropped_bytes() == 3
    assert t.match_error("missing start") == "OK"


def unexpected_last_type(mk, recycle):  # :459
    t = mk(recycle)
    t.write(b"foo")
    t.set_byte(6, 8 if recycle else 4)
    t.fix_checksum(0, 3, bool(recycle))
    assert t.read() == "EOF"
    assert t.dropped_bytes() == 3
    assert t.match_error("missing start") == "OK"


def unexpected_full_type(mk, recycle):  # :468
    t = mk(recycle)
    t.write(b"foo")
    t.write(b"bar")
    t.set_byte(6, 6 if recycle else 2)
    t.fix_checksum(0, 3, bool(recycle))
    assert t.read() == b"bar"
    assert t.read() == "EOF"
    assert t.dropped_bytes() == 3
    assert t.match_error("partial record without end") == "OK"


def unexpected_first_type(mk, recycle):  # :479
    t = mk(recycle)
    t.write(b"foo")
    t.write(big_string(b"bar", 100000))
    t.set_byte(6, 6 if recycle else 2)
    t.fix_checksum(0, 3, bool(recycle))
    assert t.read() == big_string(b"bar", 100000)
    assert t.read() == "EOF"
    assert t.dropped_bytes() == 3
    assert t.match_error("partial record without end") == "OK"


def missing_last_is_ignored(mk, recycle):  # :490
    t = mk(recycle)
    t.write(big_string(b"bar", K_BLOCK))
    t.shrink_size(14)
    assert t.read() == "EOF"
    assert t.report_message() == ""
    assert t.dropped_bytes() == 0


def missing_last_is_not_ignored(mk, recycle):  # :499
    t = mk(recycle)
    t.write(big_string(b"bar", K_BLOCK))
    t.shrink_size(14)
    assert t.read(ABSOLUTE) == "EOF"
    assert t.dropped_bytes() > 0
    assert t.match_error("Corruption: error reading trailing data") == "OK"


def partial_last_is_ignored(mk, recycle):  # :508
    t = mk(recycle)
    t.write(big_string(b"bar", K_BLOCK))
    t.shrink_size(1)
    assert t.read() == "EOF"
    assert t.report_message() == ""
    assert t.dropped_bytes() == 0


def partial_last_is_not_ignored(mk, recycle):  # :517
    t = mk(recycle)
    t.write(big_string(b"bar", K_BLOCK))
    t.shrink_size(1)
    assert t.read(ABSOLUTE) == "EOF"
    assert t.dropped_bytes() > 0
    assert t.match_error("Corruption: truncated headerCorruption: error reading trailing data") == "OK"


def error_joins_records(mk, recycle):  # :527
    t = mk(recycle)
    t.write(big_string(b"foo", K_BLOCK))
    t.write(big_string(b"bar", K_BLOCK))
    t.write(b"correct")
    for off in range(K_BLOCK, 2 * K_BLOCK):
        t.set_byte(off, ord("x"))
    if not recycle:
        assert t.read() == b"correct"
        assert t.read() == "EOF"
        d = t.dropped_bytes()
        assert 2 * K_BLOCK <= d <= 2 * K_BLOCK + 100
    else:
        assert t.read() == "EOF"


def clear_eof_single_block(mk, recycle):  # :554
    t = mk(recycle)
    t.write(b"foo")
    t.write(b"bar")
    t.force_eof(3 + hs(recycle) + 2)
    assert t.read() == b"foo"
    t.unmark_eof()
    assert t.read() == b"bar"
    assert t.is_eof()
    assert t.read() == "EOF"
    t.write(b"xxx")
    t.unmark_eof()
    assert t.read() == b"xxx"
    assert t.is_eof()


def clear_eof_multi_block(mk, recycle):  # :570
    t = mk(recycle)
    nfull = 5
    n = (K_BLOCK - hs(recycle)) * nfull + 25
    t.write(big_string(b"foo", n))
    t.write(big_string(b"bar", n))
    t.force_eof(n + nfull * hs(recycle) + hs(recycle) + 3)
    assert t.read() == big_string(b"foo", n)
    assert t.is_eof()
    t.unmark_eof()
    assert t.read() == big_string(b"bar", n)
    assert t.is_eof()
    t.write(big_string(b"xxx", n))
    t.unmark_eof()
    assert t.read() == big_string(b"xxx", n)
    assert t.is_eof()


def clear_eof_error(mk, recycle):  # :588
    t = mk(recycle)
    t.write(b"foo")
    t.write(b"bar")
    t.unmark_eof()
    assert t.read() == b"foo"
    assert t.is_eof()
    t.write(b"xxx")
    t.force_error(0)
    t.unmark_eof()
    assert t.read() == b"bar"
    assert t.read() == "EOF"


def clear_eof_error2(mk, recycle):  # :606
    t = mk(recycle)
    t.write(b"foo")
    t.write(b"bar")
    t.unmark_eof()
    assert t.read() == b"foo"
    t.write(b"xxx")
    t.force_error(3)
    t.unmark_eof()
    assert t.read() == b"bar"
    assert t.read() == "EOF"
    assert t.dropped_bytes() == 3
    assert t.match_error("read error") == "OK"


def recycle_case(mk, recycle):  # :620
    if not recycle:
        return
    t = mk(recycle)
    for m in (b"foo", b"bar", b"baz", b"bif", b"blitz"):
        t.write(m)
    while t.written_bytes() < K_BLOCK * 2:
        t.write(b"xxxxxxxxxxxxxxxx")
    w2 = t.overwriting_writer()
    w2.write(b"foooo")
    w2.write(b"bar")
    assert t.written_bytes() >= K_BLOCK * 2
    assert t.read() == b"foooo"
    assert t.read() == b"bar"
    assert t.read() == "EOF"


ALL = [empty, read_write, fragmentation, marginal_trailer, marginal_trailer2, short_trailer, aligned_eof,
       read_error, bad_record_type, truncated_trailing_record_is_ignored,
       truncated_trailing_record_is_not_ignored, bad_length, bad_length_at_end_is_ignored,
       bad_length_at_end_is_not_ignored, checksum_mismatch, unexpected_middle_type, unexpected_last_type,
       unexpected_full_type, unexpected_first_type, missing_last_is_ignored, missing_last_is_not_ignored,
       partial_last_is_ignored, partial_last_is_not_ignored, error_joins_records, clear_eof_single_block,
       clear_eof_multi_block, clear_eof_error, clear_eof_error2, recycle_case]
