"""A log checked piece by piece (a file streamed through staging buffers) must give what one pass over
the whole image gives: the same per-block results, counters and record count - which the CPU oracle's
reader rules pin.  Only the file's real end may report a truncated header or record."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu
BLOCK = 32768
END_TRAILER, END_SHORT_AT_EOF, END_BAD_LEN, END_BAD_LEN_AT_EOF = 1, 2, 3, 4  # TCRC_WAL_END_*


@pytest.fixture(scope="module")
def eng():
    import gpu_util as G
    e = G.engine()
    yield e
    e.close()


def _damaged_image(recycle, seed):
    img = bytearray(O.build_wal_image(recycle, n_random=400))
    recs, n = O.list_physical(bytes(img), 123)
    rng = np.random.default_rng(seed)
    hs = 11 if recycle else 7
    for ri in rng.choice(n, size=25, replace=False):
        off, typ, ln, _ = (int(x) for x in recs[ri])
        if ln:
            img[off + hs + int(rng.integers(0, ln))] ^= 0x10
    return bytes(img)


def _device_parts(eng, img, log_number, cuts):
    """cuts: block indices where a new part starts."""
    import gpu_util as G
    import terarkdb_amd.capi as capi
    arr = np.frombuffer(img, dtype=np.uint8).copy()
    dev = G.to_dev(arr)
    nb = (arr.size + BLOCK - 1) // BLOCK
    blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(capi.CNT_WORDS, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    edges = [0] + list(cuts) + [nb]
    total = 0
    for b0, b1 in zip(edges[:-1], edges[1:]):
        if b0 == b1:
            continue
        lo, hi = b0 * BLOCK, min(b1 * BLOCK, arr.size)
        n = C.c_uint64(0)
        capi.check(capi.lib.tcrc_dev_wal_verify_part(
            eng.ctx, C.c_void_p(dev.data_ptr() + lo), hi - lo, log_number, b0, int(b1 < nb),
            C.c_void_p(blocks.data_ptr() + 16 * b0), C.c_void_p(counts.data_ptr()), C.byref(n), eng._stream(None)))
        total += n.value
    torch.cuda.synchronize()
    return blocks.cpu().numpy().view(np.uint32), counts.cpu().numpy().view(np.uint64), total


@pytest.mark.parametrize("recycle", [0, 1])
def test_device_parts_equal_the_whole_image(eng, recycle):
    from test_gpu_wal import device_scan
    img = _damaged_image(recycle, 77 + recycle)
    want = O.wal_scan_blocks(img, 123)
    whole, whole_counts, whole_n = device_scan(eng, img, 123)
    assert (whole == want).all()
    nb = want.shape[0]
    for cuts in ([1], [nb // 2], [1, 2, 3, nb - 1], list(range(1, nb))):
        got, counts, n = _device_parts(eng, img, 123, cuts)
        assert (got == want).all(), cuts
        assert counts.tolist() == whole_counts.tolist(), cuts
        assert n == whole_n


def test_a_block_tail_inside_the_file_is_no_truncation(eng):
    """A block that ends in a trailer too short for a header (db/log_writer.cc:61-73) or in a record
    length that runs past the block is a truncated header / record only in the file's last block
    (db/log_reader.cc:330-346,357-372)."""
    import terarkdb_amd.capi as capi
    t = O.OracleLog(recycle=False, log_number=9)
    t.write(b"a" * (BLOCK - 7 - 3))  # leaves a 3-byte trailer in block 0
    t.write(b"b" * 100)
    t.write(b"c" * (2 * BLOCK))
    img = bytearray(t.contents())
    t.close()
    assert len(img) > 2 * BLOCK
    # block 1: make the first record's length run past the block
    img[BLOCK + 4:BLOCK + 6] = (0xffff).to_bytes(2, "little")
    img = bytes(img)
    want = O.wal_scan_blocks(img, 9)
    assert want[0, 3] == END_TRAILER and want[1, 3] == END_BAD_LEN
    nb = want.shape[0]
    got, counts, n = _device_parts(eng, img, 9, list(range(1, nb)))
    assert (got == want).all()
    # the same bytes as a file that ends after block 0 / block 1 do report the truncation
    got0, _, _ = _device_parts(eng, img[:BLOCK], 9, [])
    assert got0[0, 3] == END_SHORT_AT_EOF and (got0 == O.wal_scan_blocks(img[:BLOCK], 9)).all()
    got1, _, _ = _device_parts(eng, img[:2 * BLOCK], 9, [1])
    assert got1[0, 3] == END_TRAILER and got1[1, 3] == END_BAD_LEN_AT_EOF
    assert (got1 == O.wal_scan_blocks(img[:2 * BLOCK], 9)).all()


def test_part_arguments(eng):
    import gpu_util as G
    import terarkdb_amd.capi as capi
    dev = G.to_dev(np.zeros(3 * BLOCK, dtype=np.uint8))
    blocks = torch.zeros((3, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    n = C.c_uint64(0)
    rc = capi.lib.tcrc_dev_wal_verify_part(eng.ctx, C.c_void_p(dev.data_ptr()), BLOCK + 5, 1, 0, 1,
                                           C.c_void_p(blocks.data_ptr()), C.c_void_p(counts.data_ptr()), C.byref(n),
                                           eng._stream(None))
    assert rc == capi.ERR_INVALID
    hb = np.zeros(BLOCK + 5, dtype=np.uint8)
    hblocks = np.zeros((2, 4), dtype=np.uint32)
    rc = capi.lib.tcrc_host_wal_verify_part(eng.ctx, hb.ctypes.data, BLOCK + 5, 1, 0, 1, hblocks.ctypes.data, None,
                                            C.byref(n))
    assert rc == capi.ERR_INVALID


@pytest.mark.parametrize("recycle", [0, 1])
def test_file_streamed_in_small_pieces(eng, recycle, tmp_path):
    """tcrc_shim_wal_verify_file with pieces of one to a few blocks against the oracle's verdicts."""
    import terarkdb_amd.capi as capi
    img = _damaged_image(recycle, 5 + recycle)
    want = O.wal_scan_blocks(img, 123)
    path = tmp_path / "000123.log"
    path.write_bytes(img)
    nb = want.shape[0]
    n_bad_blocks = int((want[:, 1] != 0xffffffff).sum())
    assert n_bad_blocks > 5
    first_bad = int(np.nonzero(want[:, 1] != 0xffffffff)[0][0])
    results = []
    for extent, threads in [(0, 0), (BLOCK, 1), (3 * BLOCK, 2), (5 * BLOCK + 17, 8), (1 << 30, 3)]:
        blocks = np.zeros((nb, 4), dtype=np.uint32)
        report = (C.c_uint64 * 5)()
        msg = C.create_string_buffer(256)
        rc = capi.lib.tcrc_shim_wal_verify_file(eng.ctx, str(path).encode(), 123, extent, threads, blocks.ctypes.data, nb,
                                                report, msg, 256)
        assert rc == 0, msg.value
        assert (blocks == want).all(), extent
        assert report[0] == len(img) and report[1] == int(want[:, 0].sum())
        assert report[3] == first_bad and report[4] == int(want[-1, 3])
        results.append(report[2])
    assert len(set(results)) == 1 and results[0] >= n_bad_blocks
    # a missing file
    msg = C.create_string_buffer(256)
    assert capi.lib.tcrc_shim_wal_verify_file(eng.ctx, str(tmp_path / "nope.log").encode(), 1, 0, 0, None, 0, None, msg,
                                              256) == -1
    assert msg.value.startswith(b"IO error")
    # an empty one
    (tmp_path / "000001.log").write_bytes(b"")
    report = (C.c_uint64 * 5)()
    assert capi.lib.tcrc_shim_wal_verify_file(eng.ctx, str(tmp_path / "000001.log").encode(), 1, 0, 0, None, 0, report,
                                              msg, 256) == 0
    assert list(report)[:3] == [0, 0, 0]
