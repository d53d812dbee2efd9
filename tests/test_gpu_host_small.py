"""Small host-pointer calls (one fetched block, a handful of records): the engine serves them without
copy commands, reading pageable input through a bounce area and pinned input where it lies.  Values,
verdicts and counters are checked against the CPU oracle for both kinds of memory, including spans
that start in the middle of a pinned allocation."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import gpu_util as G
    e = G.engine()
    yield e
    e.close()


def _descs(offsets, lens, inits):
    import terarkdb_amd.capi as capi
    d = (capi.Desc * len(offsets))()
    for i, (o, l, c) in enumerate(zip(offsets, lens, inits)):
        d[i].offset, d[i].len, d[i].init = int(o), int(l), int(c)
    return d


def _pinned(eng, nbytes):
    import terarkdb_amd.capi as capi
    p = capi.lib.tcrc_host_alloc(eng.ctx, nbytes)
    assert p
    return p, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))


def _layout(rng, n, total):
    lens = rng.integers(0, 700, n).astype(np.uint32)
    lens[:4] = [0, 1, 15, 16]
    offs = np.sort(rng.integers(0, total - 800, n)).astype(np.uint64)
    inits = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    return offs, lens, inits


@pytest.mark.parametrize("memory", ["pageable", "pinned", "pinned_interior"])
def test_small_stamp_and_verify(eng, memory):
    import terarkdb_amd.capi as capi
    L = capi.lib
    rng = np.random.default_rng(11)
    total = 60000 if memory == "pageable" else 900000  # pinned input is read in place up to 1 MiB
    data = O.splitmix(total + 8 - total % 8, 17)[:total]
    shift = 0
    handle = None
    if memory == "pageable":
        buf = data.copy()
    else:
        shift = 4099 if memory == "pinned_interior" else 0
        handle, whole = _pinned(eng, total + shift)
        buf = whole[shift:]
        buf[:] = data
    n = 300
    offs, lens, inits = _layout(rng, n, total)
    d = _descs(offs, lens, inits)
    sfx = rng.integers(0, 256, n).astype(np.uint8)
    want = O.batch_extend(data, offs, lens, inits, sfx, True)
    out = (C.c_uint32 * n)()
    assert L.tcrc_host_batch_stamp(eng.ctx, buf.ctypes.data, total, d, n, sfx.ctypes.data, capi.F_MASK, out) == 0
    assert (np.frombuffer(out, dtype=np.uint32) == want).all()

    # verify against given values, some of them wrong
    expected = want.copy()
    bad = sorted(rng.choice(n, 7, replace=False).tolist())
    expected[bad] ^= 0x40
    ok = np.full(n, 9, dtype=np.uint8)
    got = np.zeros(n, dtype=np.uint32)
    counts = (C.c_uint64 * 3)()
    assert L.tcrc_host_batch_verify(eng.ctx, buf.ctypes.data, total, d, n, sfx.ctypes.data, 0, expected.ctypes.data,
                                    got.ctypes.data, ok.ctypes.data, counts) == 0
    assert list(counts) == [7, bad[0], n]
    assert (got == want).all()
    assert sorted(np.nonzero(ok == 0)[0].tolist()) == bad and set(ok.tolist()) == {0, 1}
    # clean
    assert L.tcrc_host_batch_verify(eng.ctx, buf.ctypes.data, total, d, n, sfx.ctypes.data, 0, want.ctypes.data,
                                    None, None, counts) == 0
    assert list(counts) == [0, 2**64 - 1, n]
    if handle:
        L.tcrc_host_free(eng.ctx, handle)


@pytest.mark.parametrize("memory", ["pageable", "pinned_interior"])
def test_one_fetched_block_with_its_trailer(eng, memory):
    """table/block_fetcher.cc:36-71 for a single block as a read path meets it: contents, type byte and
    the stored value in one host buffer."""
    import terarkdb_amd.capi as capi
    L = capi.lib
    n_bytes = 4096 + 37
    block = O.splitmix(n_bytes + 8 - n_bytes % 8, 5)[:n_bytes].tobytes()
    img = block + b"\x00" + int(O.mask(O.value(block + b"\x00"))).to_bytes(4, "little")
    handle = None
    if memory == "pageable":
        buf = np.frombuffer(img, dtype=np.uint8).copy()
    else:
        handle, whole = _pinned(eng, 1 << 16)
        buf = whole[777:777 + len(img)]
        buf[:] = np.frombuffer(img, dtype=np.uint8)
    d = _descs([0], [n_bytes + 1], [0])
    counts = (C.c_uint64 * 3)()
    got = (C.c_uint32 * 1)()
    ok = (C.c_uint8 * 1)()
    args = (eng.ctx, buf.ctypes.data, len(img), d, 1, None, capi.F_EXPECT_AFTER, None, got, ok, counts)
    assert L.tcrc_host_batch_verify(*args) == 0
    assert list(counts) == [0, 2**64 - 1, 1] and ok[0] == 1 and got[0] == O.mask(O.value(block + b"\x00"))
    buf[1000] ^= 0x80  # db/corruption_test.cc:154-188 flips bits like this
    assert L.tcrc_host_batch_verify(*args) == 0
    assert list(counts) == [1, 0, 1] and ok[0] == 0
    if handle:
        L.tcrc_host_free(eng.ctx, handle)


def test_small_pinned_scalar_calls(eng):
    """tcrc_extend on short inputs that start anywhere inside a pinned allocation."""
    handle, whole = _pinned(eng, 1 << 15)
    whole[:] = O.splitmix(1 << 15, 23)
    for off, ln in [(0, 1), (1, 15), (3, 16), (13, 4096), (4097, 9000), (255, 16384), (16383, 16384)]:
        piece = whole[off:off + ln]
        assert eng.extend(0xdeadbeef, piece) == O.extend(0xdeadbeef, piece.tobytes()), (off, ln)
    import terarkdb_amd.capi as capi
    capi.lib.tcrc_host_free(eng.ctx, handle)


def test_descriptor_limits(eng):
    """Host forms reject descriptors they can see; the device forms cannot look, so the kernel leaves a
    descriptor above 1 GiB alone and a verify counts it as a mismatch."""
    import gpu_util as G
    import terarkdb_amd.capi as capi
    L = capi.lib
    hb = np.zeros(4096, dtype=np.uint8)
    ho = (C.c_uint32 * 2)()
    hc = (C.c_uint64 * 3)()
    d = _descs([2**64 - 1], [2], [0])  # offset + len wraps around
    assert L.tcrc_host_batch_stamp(eng.ctx, hb.ctypes.data, 4096, d, 1, None, 1, ho) == capi.ERR_INVALID
    assert L.tcrc_host_batch_verify(eng.ctx, hb.ctypes.data, 4096, d, 1, None, capi.F_EXPECT_AFTER, None, None, None,
                                    hc) == capi.ERR_INVALID
    d = _descs([4093], [0], [0])  # the stored value would lie outside
    assert L.tcrc_host_batch_verify(eng.ctx, hb.ctypes.data, 4096, d, 1, None, capi.F_EXPECT_AFTER, None, None, None,
                                    hc) == capi.ERR_INVALID
    d = _descs([0], [(1 << 30) + 1], [0])
    assert L.tcrc_host_batch_stamp(eng.ctx, hb.ctypes.data, 1 << 31, d, 1, None, 1, ho) == capi.ERR_TOO_LARGE

    host = O.splitmix(1 << 16, 31)
    dev = G.to_dev(host)
    offs, lens, inits = [0, 100, 4096], [5000, 0xfffffff0, 4096], [0, 0, 7]
    desc = capi.make_desc_tensor(offs, lens, inits, "cuda")
    out = torch.zeros(3, dtype=torch.int32, device="cuda")
    ok = torch.full((3,), 9, dtype=torch.uint8, device="cuda")
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    want = O.batch_extend(host, np.array([0, 4096], dtype=np.uint64), np.array([5000, 4096], dtype=np.uint32),
                          np.array([0, 7], dtype=np.uint32), None, True)
    expected = torch.from_numpy(np.array([want[0], O.mask(0), want[1]], dtype=np.uint32).view(np.int32)).cuda()  # mask(0): an empty span's value
    eng.batch_verify(dev, desc, 3, counts, expected=expected, out=out, ok=ok)
    torch.cuda.synchronize()
    assert ok.cpu().tolist() == [1, 0, 1]
    assert counts.cpu().tolist() == [1, 1, 3]
    got = out.cpu().numpy().view(np.uint32)
    assert got[0] == want[0] and got[2] == want[1]


def test_few_buffers_some_of_them_long(eng):
    """A small batch that holds long buffers (a table's filter or index block among its meta blocks) takes
    another route through the engine; values, verdicts and counters must not depend on it."""
    import terarkdb_amd.capi as capi
    L = capi.lib
    rng = np.random.default_rng(21)
    lens = [300000, 0, 17, (1 << 20) + 3, 4097, 81920, 81919]
    total = sum(lens) + 5 * len(lens) + 64
    img = O.splitmix(total + 8 - total % 8, 77)[:total].copy()
    offs, pos = [], 13
    for ln in lens:
        offs.append(pos)
        pos += ln + 5
    inits = rng.integers(0, 2**32, len(lens), dtype=np.uint64).astype(np.uint32)
    d = _descs(offs, lens, inits)
    n = len(lens)
    for masked in (False, True):
        want = O.batch_extend(img, np.array(offs, dtype=np.uint64), np.array(lens, dtype=np.uint32), inits, None, masked)
        out = (C.c_uint32 * n)()
        assert L.tcrc_host_batch_stamp(eng.ctx, img.ctypes.data, total, d, n, None, capi.F_MASK if masked else 0, out) == 0
        assert (np.frombuffer(out, dtype=np.uint32) == want).all()
    # trailers in the image: [span][stored masked value]
    d0 = _descs(offs, lens, [0] * n)
    want = O.batch_extend(img, np.array(offs, dtype=np.uint64), np.array(lens, dtype=np.uint32), None, None, True)
    for o, ln, w in zip(offs, lens, want):
        img[o + ln:o + ln + 4] = np.frombuffer(int(w).to_bytes(4, "little"), dtype=np.uint8)
    counts = (C.c_uint64 * 3)()
    ok = np.full(n, 9, dtype=np.uint8)
    got = np.zeros(n, dtype=np.uint32)
    args = (eng.ctx, img.ctypes.data, total, d0, n, None, capi.F_EXPECT_AFTER, None, got.ctypes.data, ok.ctypes.data, counts)
    assert L.tcrc_host_batch_verify(*args) == 0
    assert list(counts) == [0, 2**64 - 1, n] and ok.tolist() == [1] * n and (got == want).all()
    img[offs[3] + 999999] ^= 1   # inside the 1 MiB buffer
    img[offs[5] + lens[5] + 2] ^= 4  # a stored value
    assert L.tcrc_host_batch_verify(*args) == 0
    assert list(counts) == [2, 3, n] and ok.tolist() == [1, 1, 1, 0, 1, 0, 1]
    # given values instead of in-image ones; unknown flag bits are refused on this route too
    expected = want.copy()
    expected[0] ^= 1
    assert L.tcrc_host_batch_verify(eng.ctx, img.ctypes.data, total, d0, n, None, 0, expected.ctypes.data, None, None,
                                    counts) == 0
    assert list(counts) == [2, 0, n]
    assert L.tcrc_host_batch_verify(eng.ctx, img.ctypes.data, total, d0, n, None, capi.F_STORE_AFTER,
                                    expected.ctypes.data, None, None, counts) == capi.ERR_INVALID
    assert L.tcrc_host_batch_verify(eng.ctx, img.ctypes.data, total, d0, n, None, 0, None, None, None,
                                    counts) == capi.ERR_INVALID


def test_short_span_hint_with_one_long_buffer(eng):
    """TCRC_F_SHORT_SPANS is a hint: a table's short data blocks plus one large filter or index block (which sends
    the call down the long-buffer route) must give the same values and verdicts with and without it."""
    import terarkdb_amd.capi as capi
    L = capi.lib
    lens = [700, 1100, 90000, 333, 0, 1408, 262144 + 7]
    total = sum(lens) + 5 * len(lens) + 64
    img = O.splitmix(total + 8 - total % 8, 78)[:total].copy()
    offs, pos = [], 7
    for ln in lens:
        offs.append(pos)
        pos += ln + 5
    n = len(lens)
    d = _descs(offs, lens, [0] * n)
    want = O.batch_extend(img, np.array(offs, dtype=np.uint64), np.array(lens, dtype=np.uint32), None, None, True)
    for hint in (0, capi.F_SHORT_SPANS):
        out = (C.c_uint32 * n)()
        assert L.tcrc_host_batch_stamp(eng.ctx, img.ctypes.data, total, d, n, None, capi.F_MASK | hint, out) == 0, \
            capi.lib.tcrc_last_error().decode()
        assert (np.frombuffer(out, dtype=np.uint32) == want).all()
    for o, ln, w in zip(offs, lens, want):
        img[o + ln:o + ln + 4] = np.frombuffer(int(w).to_bytes(4, "little"), dtype=np.uint8)
    img[offs[2] + 4242] ^= 0x10
    for hint in (0, capi.F_SHORT_SPANS):
        counts = (C.c_uint64 * 3)()
        ok = np.full(n, 9, dtype=np.uint8)
        got = np.zeros(n, dtype=np.uint32)
        rc = L.tcrc_host_batch_verify(eng.ctx, img.ctypes.data, total, d, n, None, capi.F_EXPECT_AFTER | hint, None,
                                      got.ctypes.data, ok.ctypes.data, counts)
        assert rc == 0, capi.lib.tcrc_last_error().decode()
        assert list(counts) == [1, 2, n] and ok.tolist() == [1, 1, 0, 1, 1, 1, 1]
        assert (np.delete(got, 2) == np.delete(want, 2)).all()
