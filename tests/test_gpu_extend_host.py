"""GPU parity for the scalar-shaped host entry points (tcrc_extend / tcrc_value) and the whole-file
checksum helpers built on them, against the CPU oracle and the reference's known answers."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "crc32c_known_answers.json")))


@pytest.fixture(scope="module")
def eng():
    import gpu_util as G
    e = G.engine()
    yield e
    e.close()


def test_value_known_answers(eng):
    # util/crc32c_test.cc:67-94 (RFC 3720 B.4) and the ASCII vectors
    assert eng.value(b"123456789") == 0xe3069283
    assert eng.value(b"") == 0
    assert eng.extend(0x1234, b"") == 0x1234
    assert eng.value(bytes(32)) == 0x8a9136aa
    assert eng.value(b"\xff" * 32) == 0x62a8ab43
    assert eng.value(bytes(range(32))) == 0x46dd794e
    assert eng.value(bytes(range(31, -1, -1))) == 0x113fdb5c
    # Extend(Value(a), b) == Value(a + b)   (util/crc32c_test.cc:114-117)
    assert eng.extend(eng.value(b"hello "), b"world") == eng.value(b"hello world")


def test_folly_table_over_fnv_buffer(eng):
    # util/crc32c_test.cc:31-65,96-111: one-shot and split through Extend
    buf = O.fnv_chain()
    for off, ln, not_crc in KAT["fnv_chain"]:
        piece = buf[off:off + ln]
        want = ~not_crc & 0xffffffff
        assert eng.value(piece) == want, (off, ln)
        half = ln // 3
        assert eng.extend(eng.value(piece[:half]), piece[half:]) == want, (off, ln)
    for v in KAT["rfc3720"]:
        assert eng.value(bytes.fromhex(v["hex"])) == int(v["crc"], 16), v["name"]


def test_random_lengths_and_inits(eng):
    rng = np.random.default_rng(3)
    big = O.splitmix(9 << 20, 99)
    for ln in [1, 2, 3, 4, 7, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 1000, 2047, 2048, 2049, 4095, 8192, 8193,
               65535, 65536, 65537, 262144 - 7, 262144 + 1, (1 << 20) + 7, (4 << 20) - 1, (8 << 20) + 13]:
        off = int(rng.integers(0, 64))
        init = int(rng.integers(0, 2**32))
        piece = big[off:off + ln]
        assert eng.extend(init, piece) == O.extend(init, piece.tobytes()), ln


def test_more_than_one_piece(eng):
    # above 256 MiB the call chains pieces through init
    n = (256 << 20) + (40 << 20) + 5
    a = O.splitmix((n + 7) // 8 * 8, 5)[:n]
    assert eng.extend(77, a) == O.extend(77, a, variant="hw")


def _file_checksum(eng, path, size_limit=0, buffer_size=0):
    import terarkdb_amd.capi as capi
    crc, size = C.c_uint32(0), C.c_uint64(0)
    msg = C.create_string_buffer(512)
    rc = capi.lib.tcrc_shim_file_checksum(eng.ctx, path.encode(), size_limit, buffer_size, C.byref(crc), C.byref(size),
                                          msg, 512)
    return rc, crc.value, size.value, msg.value.decode()


def test_file_checksum_and_copy(eng, tmp_path):
    import terarkdb_amd.capi as capi
    n = (23 << 20) + 13
    data = O.splitmix((n + 7) // 8 * 8, 41)[:n]
    src = str(tmp_path / "000012.sst")
    data.tofile(src)
    want = O.value(data, variant="hw")
    for bs in (0, 1 << 20, 5 << 20, (5 << 20) + 3, 64 << 20):
        assert _file_checksum(eng, src, 0, bs) == (0, want, n, "OK")
    for limit in (1, 4096, (5 << 20) + 1, n - 1, n, n + 100):
        k = min(limit, n)
        assert _file_checksum(eng, src, limit, 1 << 20) == (0, O.value(data[:k], variant="hw"), k, "OK")
    rc, _, _, msg = _file_checksum(eng, str(tmp_path / "missing"))
    assert rc == -1 and "missing" in msg and msg.startswith("IO error")
    empty = str(tmp_path / "empty")
    open(empty, "wb").close()
    assert _file_checksum(eng, empty) == (0, 0, 0, "OK")

    dst = str(tmp_path / "copy.sst")
    crc, size = C.c_uint32(0), C.c_uint64(0)
    msg = C.create_string_buffer(512)
    rc = capi.lib.tcrc_shim_copy_file(eng.ctx, src.encode(), dst.encode(), 0, 3 << 20, 1, C.byref(crc), C.byref(size),
                                      msg, 512)
    assert (rc, crc.value, size.value) == (0, want, n)
    assert (np.fromfile(dst, dtype=np.uint8) == data).all()
    limit = (7 << 20) + 5
    rc = capi.lib.tcrc_shim_copy_file(eng.ctx, src.encode(), dst.encode(), limit, 2 << 20, 0, C.byref(crc),
                                      C.byref(size), msg, 512)
    assert (rc, crc.value, size.value) == (0, O.value(data[:limit], variant="hw"), limit)
    assert os.path.getsize(dst) == limit


def test_back_to_back_small_calls_return_their_own_results(eng):
    # a small call returns when its kernel has written the completion word, not after a stream
    # synchronisation: thousands of calls in a row, each result checked, sizes changing from call to call
    rng = np.random.default_rng(77)
    pool = rng.integers(0, 256, 1 << 16, dtype=np.uint8).tobytes()
    for i in range(3000):
        n = int(rng.integers(1, 9000))
        o = int(rng.integers(0, len(pool) - n))
        init = int(rng.integers(0, 1 << 32))
        assert eng.extend(init, pool[o:o + n]) == O.extend(init, pool[o:o + n]), (i, n, o)


def test_back_to_back_small_batches_of_changing_size(eng):
    # the same for descriptor batches that span one to many workgroups
    import terarkdb_amd.capi as capi
    rng = np.random.default_rng(78)
    image = rng.integers(0, 256, 60000, dtype=np.uint8)  # short enough to go through the zero-copy block
    for i in range(400):
        n = int(rng.integers(1, 700))
        lens = rng.integers(0, 1400, n).astype(np.uint32)
        offs = rng.integers(0, image.size - 1400, n).astype(np.uint64)
        inits = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        d = (capi.Desc * n)()
        for k in range(n):
            d[k].offset, d[k].len, d[k].init = int(offs[k]), int(lens[k]), int(inits[k])
        out = (C.c_uint32 * n)()
        assert capi.lib.tcrc_host_batch_stamp(eng.ctx, image.ctypes.data, image.size, d, n, None, capi.F_MASK, out) == 0
        want = O.batch_extend(image, offs, lens, inits, None, True)
        assert (np.frombuffer(out, dtype=np.uint32) == want).all(), i
