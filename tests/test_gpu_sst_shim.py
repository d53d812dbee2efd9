"""SST block trailers through the engine's WriteRawBlocks / CheckBlockChecksums, against the CPU oracle
and the reference's Corruption text (table/block_fetcher.cc:64-69)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import gpu_util as G
    e = G.engine()
    yield e
    e.close()


def make_blocks(seed, n):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(1, 9000, size=n).astype(np.uint64)
    sizes[:6] = [0, 1, 4096, 8192, 8193, 16]
    types = rng.integers(0, 8, size=n).astype(np.uint8)  # CompressionType bytes
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    base = rng.integers(0, 256, size=int(sizes.sum()) + 8, dtype=np.uint8)
    return base, offs, sizes, types


def test_trailers_equal_the_builder_rule(eng):
    import terarkdb_amd.capi as capi
    base, offs, sizes, types = make_blocks(3, 400)
    n = len(sizes)
    out = np.zeros(5 * n, dtype=np.uint8)
    rc = capi.lib.tcrc_shim_block_trailers(eng.ctx, base.ctypes.data, offs.ctypes.data, sizes.ctypes.data,
                                           types.ctypes.data, n, 1, out.ctypes.data)
    assert rc == 0
    for i in range(n):
        blk = base[int(offs[i]): int(offs[i] + sizes[i])]
        want = O.lib().oracle_sst_trailer_crc(blk.ctypes.data if blk.size else None, blk.size, int(types[i]), 0)
        tr = out[5 * i: 5 * i + 5]
        assert tr[0] == types[i]
        assert int.from_bytes(bytes(tr[1:]), "little") == want, i
    # kNoChecksum writes a zero checksum field; the xxHash kinds are declined
    rc = capi.lib.tcrc_shim_block_trailers(eng.ctx, base.ctypes.data, offs.ctypes.data, sizes.ctypes.data,
                                           types.ctypes.data, n, 0, out.ctypes.data)
    assert rc == 0 and (out.reshape(n, 5)[:, 1:] == 0).all()
    for kind in (2, 3):
        assert capi.lib.tcrc_shim_block_trailers(eng.ctx, base.ctypes.data, offs.ctypes.data, sizes.ctypes.data,
                                                 types.ctypes.data, n, kind, out.ctypes.data) == -2


def test_write_then_check_and_corruption_message(eng):
    import terarkdb_amd.capi as capi
    L = capi.lib
    base, offs, sizes, types = make_blocks(4, 300)
    n = len(sizes)
    f = L.tcrc_shim_memfile_new()
    L.tcrc_shim_memfile_append(f, b"HEAD" * 4, 16)  # the table does not start at offset 0
    hoffs = np.zeros(n, dtype=np.uint64)
    assert L.tcrc_shim_write_raw_blocks(eng.ctx, f, base.ctypes.data, offs.ctypes.data, sizes.ctypes.data,
                                        types.ctypes.data, n, 1, hoffs.ctypes.data) == 0
    size = L.tcrc_shim_memfile_size(f)
    assert size == 16 + int(sizes.sum()) + 5 * n
    assert int(hoffs[0]) == 16 and int(hoffs[1]) == 16 + int(sizes[0]) + 5
    img = np.frombuffer(C.string_at(L.tcrc_shim_memfile_data(f), size), dtype=np.uint8).copy()
    ok = np.zeros(n, dtype=np.uint8)
    msg = C.create_string_buffer(512)
    rc = L.tcrc_shim_check_blocks(eng.ctx, img.ctypes.data, img.size, 0, hoffs.ctypes.data, sizes.ctypes.data, n, 1,
                                  b"000007.sst", ok.ctypes.data, msg, 512)
    assert rc == 0 and ok.all() and msg.value == b""
    # every block also passes the CPU oracle's CheckBlockChecksum
    for i in range(n):
        assert O.lib().oracle_sst_check_block(img.ctypes.data + int(hoffs[i]), int(sizes[i]), None, None) == 1
    # flip one bit in two blocks (the corruption_test recipe) and compare the first message
    for b in (200, 77):
        img[int(hoffs[b]) + int(sizes[b]) // 2] ^= 0x80
    rc = L.tcrc_shim_check_blocks(eng.ctx, img.ctypes.data, img.size, 0, hoffs.ctypes.data, sizes.ctypes.data, n, 1,
                                  b"000007.sst", ok.ctypes.data, msg, 512)
    assert rc == 1
    assert sorted(np.nonzero(ok == 0)[0].tolist()) == [77, 200]
    actual, stored = C.c_uint32(), C.c_uint32()
    assert O.lib().oracle_sst_check_block(img.ctypes.data + int(hoffs[77]), int(sizes[77]), C.byref(actual),
                                          C.byref(stored)) == 0
    want = (f"Corruption: block checksum mismatch: expected {actual.value}, got {stored.value}  in 000007.sst "
            f"offset {int(hoffs[77])} size {int(sizes[77])}")
    assert msg.value.decode() == want
    # an image that starts mid-file
    start = int(hoffs[100])
    sub = img[start:].copy()
    rc = L.tcrc_shim_check_blocks(eng.ctx, sub.ctypes.data, sub.size, start, hoffs[100:].ctypes.data,
                                  sizes[100:].ctypes.data, n - 100, 1, b"x", ok.ctypes.data, msg, 512)
    assert rc == 1 and sorted(np.nonzero(ok[: n - 100] == 0)[0].tolist()) == [100]
    L.tcrc_shim_memfile_free(f)
