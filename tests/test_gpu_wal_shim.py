"""The reference's log_test scenarios through the engine's log::Writer / log::Reader with the
checksums computed on the GPU, plus file-level parity with the CPU oracle's writer/reader."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import wal_scenarios as W

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CAP = json.load(open(os.path.join(HERE, "golden", "reference_captured.json")))


@pytest.fixture(scope="module")
def eng():
    import gpu_util as G
    e = G.engine()
    yield e
    e.close()


def factory(eng, **fixed):
    from engine_log import EngineLog

    def mk(recycle, **kw):
        kw.update(fixed)
        return EngineLog(recycle=recycle, ctx=eng.ctx, **kw)
    return mk


@pytest.mark.parametrize("recycle", [0, 2])
@pytest.mark.parametrize("scenario", W.ALL, ids=lambda f: f.__name__)
def test_log_scenarios_block_by_block(eng, scenario, recycle):
    scenario(factory(eng), recycle)


@pytest.mark.parametrize("recycle", [0, 2])
@pytest.mark.parametrize("scenario", W.ALL, ids=lambda f: f.__name__)
def test_log_scenarios_whole_file_verdicts(eng, scenario, recycle):
    scenario(factory(eng, bulk_image=True), recycle)


@pytest.mark.parametrize("recycle", [0, 2])
def test_many_blocks_group_commit(eng, recycle):
    """db/log_test.cc:283 with the writer in manual-flush mode: 100000 records framed on the host,
    stamped by one device submission at WriteBuffer(), read back with one device verification."""
    from engine_log import EngineLog
    t = EngineLog(recycle=recycle, ctx=eng.ctx, manual_flush=True, bulk_image=True)
    for i in range(100000):
        t.write(O.number_string(i))
    t.flush()
    for i in range(100000):
        assert t.read() == O.number_string(i)
    assert t.read() == "EOF"
    assert t.verifier_calls() == 0


@pytest.mark.parametrize("recycle", [0, 2])
def test_random_read_bytes_match_oracle_and_capture(eng, recycle):
    t = W.random_read(factory(eng, bulk_image=True), recycle)
    img = t.contents()
    cap = CAP["wal_random_read"]["recyclable" if recycle else "legacy"]
    assert len(img) == cap["file_size"]
    assert O.value(img) == int(cap["value_of_whole_file"], 16)
    recs, n = O.list_physical(img, 123)
    assert n == cap["physical_records"]
    assert int(np.bitwise_xor.reduce(recs[:, 3].astype(np.uint32))) == int(cap["xor_of_stored_masked"], 16)


@pytest.mark.parametrize("kind", ["legacy", "recyclable"])
def test_captured_small_wal(eng, kind):
    from engine_log import EngineLog
    cap = CAP["wal_small"]
    t = EngineLog(recycle=(kind == "recyclable"), log_number=cap["log_number"], ctx=eng.ctx)
    for r in cap["records"]:
        t.write(b"a" * 40000 if r == "a*40000" else r.encode())
    img = t.contents()
    assert len(img) == cap[kind]["file_size"]
    assert img[:32].hex() == cap[kind]["first32_hex"]
    recs, n = O.list_physical(img, cap["log_number"])
    for got, want in zip(recs, cap[kind]["physical"]):
        assert [int(got[0]), int(got[1]), int(got[2])] == want[:3]
        assert int(got[3]) == int(want[3], 16)


@pytest.mark.parametrize("recycle", [0, 2])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_corrupted_image_reads_like_the_cpu_reader(eng, recycle, mode):
    """Seeded bit flips all over an image: the engine's reader (whole-file device verdicts) must return the
    same records, drop the same byte counts and report the same messages as the CPU oracle's reader, in
    every WALRecoveryMode."""
    from engine_log import EngineLog
    import terarkdb_amd.capi as capi
    if recycle and mode == 3:
        pytest.skip("kSkipAnyCorruptedRecords never leaves an old-record header in the reference (it loops)")
    img = bytearray(O.build_wal_image(bool(recycle), n_random=200))
    rng = np.random.default_rng(17 + mode)
    for _ in range(25):
        pos = int(rng.integers(0, len(img)))
        img[pos] ^= 1 << int(rng.integers(0, 8))
    img = bytes(img)

    import ctypes as C
    b = EngineLog(recycle=recycle, ctx=eng.ctx, bulk_image=True)
    capi.lib.tcrc_shim_memfile_append(b.file, img, len(img))
    # oracle side: rebuild reader over a sink filled with the same bytes
    fill = O.lib().oracle_sink_new()
    O.lib().oracle_sink_append_raw(fill, img, len(img))
    src = O.lib().oracle_source_new(fill)
    rd = O.lib().oracle_wal_reader_new(src, 1, 123, 0)
    try:
        while True:
            rec = C.c_void_p()
            n = C.c_size_t()
            ok = O.lib().oracle_wal_reader_read_record(rd, mode, C.byref(rec), C.byref(n))
            want = (C.string_at(rec, n.value) if n.value else b"") if ok else "EOF"
            got = b.read(mode)
            assert got == want
            if not ok:
                break
        assert b.dropped_bytes() == O.lib().oracle_wal_reader_dropped_bytes(rd)
        assert b.report_message() == O.lib().oracle_wal_reader_report(rd).decode()
    finally:
        O.lib().oracle_wal_reader_free(rd)
        O.lib().oracle_source_free(src)
        O.lib().oracle_sink_free(fill)


@pytest.mark.parametrize("name", ["reference_MANIFEST-CORRECT.bin", "reference_MANIFEST-WRONG.bin"])
def test_reference_manifest_files_on_device(eng, name):
    """Real TerarkDB log files: the device verifies every record, the engine's reader returns every logical
    record, and the engine's writer (device-stamped CRCs) reproduces the files byte for byte."""
    import manifest_cases as M
    import terarkdb_amd.capi as capi
    import gpu_util as G
    import torch
    img = M.load(name)
    arr = np.frombuffer(img, dtype=np.uint8).copy()
    nb = (arr.size + 32767) // 32768
    blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    n = eng.wal_verify(G.to_dev(arr), arr.size, 0, blocks, counts)
    torch.cuda.synchronize()
    assert n == M.FILES[name][0]
    assert int(counts.cpu().numpy().view(np.uint64)[capi.CNT_MISMATCH]) == 0
    assert (blocks.cpu().numpy().view(np.uint32) == O.wal_scan_blocks(img, 0)).all()

    def put(t, b):
        capi.lib.tcrc_shim_memfile_append(t.file, b, len(b))
    M.roundtrip(factory(eng), name, put)
    M.roundtrip(factory(eng, bulk_image=True), name, put)


@pytest.mark.parametrize("case_name", ["clean_log", "corrupted_log", "manifest_named_file"])
def test_dump_wal_on_device(case_name, tmp_path):
    import gpu_util as G
    import wal_dump_cases
    import wal_dump_util
    e = G.engine()
    try:
        getattr(wal_dump_cases, case_name)(wal_dump_util.dump_with_engine(e), str(tmp_path))
    finally:
        e.close()


@pytest.mark.parametrize("recycle", [False, True])
@pytest.mark.parametrize("case_name", ["partial_header", "full_header", "later_in_the_file"])
def test_tailing_reader_on_device(case_name, recycle, tmp_path):
    import gpu_util as G
    import terarkdb_amd.capi as capi
    import wal_tail_cases
    e = G.engine()
    try:
        def open_reader(path, log_number):
            return capi.lib.tcrc_shim_wal_reader_open(e.ctx, path.encode(), 1, log_number, 1)
        getattr(wal_tail_cases, case_name)(open_reader, str(tmp_path), recycle)
    finally:
        e.close()
