"""Pins the CPU oracle's log writer/reader: replays db/log_test.cc's scenarios over it and
checks the byte strings and CRCs captured from a build of the reference."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import wal_scenarios as W

HERE = os.path.dirname(os.path.abspath(__file__))
CAP = json.load(open(os.path.join(HERE, "golden", "reference_captured.json")))


def mk(recycle, **kw):
    return O.OracleLog(recycle=recycle, **kw)


@pytest.mark.parametrize("recycle", [0, 2])  # INSTANTIATE_TEST_CASE_P(bool, LogTest, Values(0, 2))
@pytest.mark.parametrize("scenario", W.ALL, ids=lambda f: f.__name__)
def test_log_scenarios(scenario, recycle):
    scenario(mk, recycle)


@pytest.mark.parametrize("recycle", [0, 2])
def test_many_blocks(recycle):
    W.many_blocks(mk, recycle, n=20000)


@pytest.mark.parametrize("recycle", [0, 2])
def test_random_read_and_captured_image_facts(recycle):
    t = W.random_read(mk, recycle)
    img = t.contents()
    cap = CAP["wal_random_read"]["recyclable" if recycle else "legacy"]
    assert len(img) == cap["file_size"]
    recs, n = O.list_physical(img, 123)
    assert n == cap["physical_records"]
    types = recs[:, 1].astype(int)
    base = 4 if recycle else 0
    assert int((types == base + 1).sum()) == cap["full"]
    assert int((types == base + 2).sum()) == cap["first"]
    assert int((types == base + 3).sum()) == cap["middle"]
    assert int((types == base + 4).sum()) == cap["last"]
    assert int(np.bitwise_xor.reduce(recs[:, 3].astype(np.uint32))) == int(cap["xor_of_stored_masked"], 16)
    assert O.value(img) == int(cap["value_of_whole_file"], 16)
    assert O.value(img, "hw") == int(cap["value_of_whole_file"], 16)


@pytest.mark.parametrize("kind", ["legacy", "recyclable"])
def test_captured_small_wal(kind):
    cap = CAP["wal_small"]
    t = mk(kind == "recyclable", log_number=cap["log_number"])
    for r in cap["records"]:
        t.write(b"a" * 40000 if r == "a*40000" else r.encode())
    img = t.contents()
    assert len(img) == cap[kind]["file_size"]
    assert img[:32].hex() == cap[kind]["first32_hex"]
    recs, n = O.list_physical(img, cap["log_number"])
    assert n == len(cap[kind]["physical"])
    for got, want in zip(recs, cap[kind]["physical"]):
        assert [int(got[0]), int(got[1]), int(got[2])] == want[:3]
        assert int(got[3]) == int(want[3], 16)


def test_type_crc_seeds():
    t = mk(0)
    for i, h in enumerate(CAP["wal_type_crc_0_to_8"]):
        assert O.lib().oracle_wal_writer_type_crc(t.writer, i) == int(h, 16)


def test_hw_variant_writes_identical_bytes():
    a, b = mk(0), mk(0, hw=True)
    rnd = O.Random(301)
    for i in range(200):
        s = W.random_skewed_string(i, rnd)
        a.write(s)
        b.write(s)
    assert a.contents() == b.contents()
    while True:
        x, y = a.read(), b.read()
        assert x == y
        if x == "EOF":
            break


def test_dbbench_shape_log_facts_scaled():
    """The CPU-reference configuration at 1/20 scale: same record shape, and the captured
    full-size facts follow from the same geometry (16 records per 32 KiB block, one split per block)."""
    import ctypes as C
    cap = CAP["wal_db_bench_shape"]
    t = mk(0, log_number=7)
    n = 50000
    O.lib().oracle_wal_write_dbbench_records(t.writer, n, 301)
    phys = O.lib().oracle_wal_writer_physical_records(t.writer)
    crc_bytes = O.lib().oracle_wal_writer_crc_bytes(t.writer)
    size = t.written_bytes()
    # the file is n * 2040 payload bytes + 7 per physical record + the zero trailers of blocks that end
    # with fewer than 7 free bytes
    pad = size - (n * cap["record_size"] + 7 * phys)
    assert 0 <= pad < 7 * (size // 32768 + 1)
    assert crc_bytes == n * cap["record_size"] + phys
    # the same two relations pin the captured full-size numbers
    cap_pad = cap["file_size"] - (cap["logical_records"] * cap["record_size"] + 7 * cap["physical_records"])
    assert 0 <= cap_pad < 7 * (cap["file_size"] // 32768 + 1)
    assert cap["crc_covered_bytes"] == cap["logical_records"] * cap["record_size"] + cap["physical_records"]
    assert cap["full"] + cap["first"] + cap["last"] == cap["physical_records"]
    # ratio of split records matches: one FIRST/LAST pair per 32 KiB block
    assert abs((phys - n) / n - cap["first"] / cap["logical_records"]) < 1e-3
    total = C.c_uint64()
    assert O.lib().oracle_wal_read_all(t.reader, 0, C.byref(total)) == n
    assert total.value == n * cap["record_size"]


@pytest.mark.parametrize("name", ["reference_MANIFEST-CORRECT.bin", "reference_MANIFEST-WRONG.bin"])
def test_reference_manifest_files(name):
    """Log files written by real TerarkDB builds: all checksums verify, and re-writing the records
    they hold gives the same bytes."""
    import manifest_cases as M
    img = M.load(name)
    recs, n = O.list_physical(img, 0)
    assert n == M.FILES[name][0]
    blocks = O.wal_scan_blocks(img, 0)
    assert (blocks[:, 1] == 0xffffffff).all() and int(blocks[:, 0].sum()) == n

    def put(t, b):
        O.lib().oracle_sink_append_raw(t.sink, b, len(b))
    M.roundtrip(mk, name, put)
