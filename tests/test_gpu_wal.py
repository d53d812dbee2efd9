"""GPU parity for write-ahead-log images: the device scan against the CPU oracle's reader rules."""
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


def device_scan(eng, img, log_number):
    import gpu_util as G
    import terarkdb_amd.capi as capi
    arr = np.frombuffer(img, dtype=np.uint8).copy() if not isinstance(img, np.ndarray) else img
    dev = G.to_dev(arr)
    nb = (arr.size + 32767) // 32768
    blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(capi.CNT_WORDS, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    nrec = eng.wal_verify(dev, arr.size, log_number, blocks, counts)
    torch.cuda.synchronize()
    return blocks.cpu().numpy().view(np.uint32), counts.cpu().numpy().view(np.uint64), nrec


@pytest.mark.parametrize("recycle", [0, 1])
def test_clean_image_all_records_verify(eng, recycle):
    import terarkdb_amd.capi as capi
    img = O.build_wal_image(recycle)
    want = O.wal_scan_blocks(img, 123)
    got, counts, nrec = device_scan(eng, img, 123)
    assert (got == want).all()
    assert int(counts[capi.CNT_MISMATCH]) == 0
    assert nrec == int(want[:, 0].sum()) == int(counts[capi.CNT_CHECKED])
    recs, n = O.list_physical(img, 123)
    assert n == nrec


@pytest.mark.parametrize("recycle", [0, 1])
def test_bit_flips_give_the_cpu_readers_verdicts(eng, recycle):
    """Seeded single-bit flips in payload, type byte, crc bytes and length bytes."""
    img = bytearray(O.build_wal_image(recycle, n_random=300))
    recs, n = O.list_physical(bytes(img), 123)
    rng = np.random.default_rng(5 + recycle)
    hs = 11 if recycle else 7
    chosen = rng.choice(n, size=60, replace=False)
    for k, ri in enumerate(chosen):
        off, typ, ln, _ = (int(x) for x in recs[ri])
        kind = k % 5
        if kind == 0 and ln > 0:
            pos = off + hs + int(rng.integers(0, ln))      # payload
        elif kind == 1:
            pos = off + 6                                  # type byte
        elif kind == 2:
            pos = off + int(rng.integers(0, 4))            # stored crc
        elif kind == 3:
            pos = off + 4 + int(rng.integers(0, 2))        # length bytes
        else:
            pos = off + (7 if recycle else 6)              # log number / type
        img[pos] ^= 1 << int(rng.integers(0, 8))
    img = bytes(img)
    want = O.wal_scan_blocks(img, 123)
    got, counts, nrec = device_scan(eng, img, 123)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, (bad[:5], got[bad[:5]], want[bad[:5]])
    assert (want[:, 1] != 0xffffffff).sum() > 10  # the flips really produced checksum failures


def test_wrong_log_number_and_truncations(eng):
    img = O.build_wal_image(1, n_random=50)
    # another log number: every recyclable record is an "old record"
    want = O.wal_scan_blocks(img, 124)
    got, counts, nrec = device_scan(eng, img, 124)
    assert (got == want).all() and nrec == 0
    for cut in (1, 3, 6, 7, 8, 11, 12, 500, 32768, 32769, 40000):
        part = img[: len(img) - cut]
        want = O.wal_scan_blocks(part, 123)
        got, counts, nrec = device_scan(eng, part, 123)
        assert (got == want).all(), cut
    # preallocated (zeroed) tail, as produced by fallocate-style writers
    padded = img + bytes(70000)
    want = O.wal_scan_blocks(padded, 123)
    got, counts, nrec = device_scan(eng, padded, 123)
    assert (got == want).all()


def test_many_tiny_records(eng):
    t = O.OracleLog(recycle=False, log_number=9)
    for i in range(30000):
        t.write(b"" if i % 3 == 0 else O.number_string(i))
    img = t.contents()
    t.close()
    want = O.wal_scan_blocks(img, 9)
    got, counts, nrec = device_scan(eng, img, 9)
    assert (got == want).all() and nrec == int(want[:, 0].sum()) and nrec >= 30000


REC = np.dtype([("offset", "<u8"), ("length", "<u4"), ("type", "u1"), ("status", "u1"), ("header_size", "<u2")])


@pytest.mark.parametrize("recycle", [False, True])
def test_record_list_matches_oracle(recycle):
    """The (offset, length, type, verdict) list in file order, device and host forms."""
    import ctypes as C
    import gpu_util as G
    import terarkdb_amd.capi as capi
    eng = G.engine()
    img = bytearray(O.build_wal_image(recycle, n_random=300, seed=11))
    rng = np.random.default_rng(21)
    for pos in rng.integers(0, len(img), 40):
        img[int(pos)] ^= 1 << int(rng.integers(0, 8))
    img = bytes(img)
    want_blocks = O.wal_scan_blocks(img, 123)
    accepted, n_acc = O.list_physical(img, 123)
    dev = G.to_dev(np.frombuffer(img, dtype=np.uint8).copy())
    nb = (len(img) + 32767) // 32768
    blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    total = eng.wal_verify(dev, len(img), 123, blocks, counts)
    assert eng.wal_list(dev, len(img), 123, blocks, None, 0) == total  # sizing call
    recs = torch.zeros(total * 16, dtype=torch.uint8, device="cuda")
    assert eng.wal_list(dev, len(img), 123, blocks, recs, total) == total
    torch.cuda.synchronize()
    r = recs.cpu().numpy().view(REC)
    assert total == int(want_blocks[:, 0].sum())
    assert (np.diff(r["offset"].astype(np.int64)) > 0).all()  # file order
    ok = r[r["status"] == 1]
    assert len(ok) == n_acc
    assert (ok["offset"] == accepted[:, 0]).all() and (ok["type"] == accepted[:, 1]).all()
    assert (ok["length"] == accepted[:, 2]).all()
    assert set(np.unique(ok["header_size"]).tolist()) <= {7, 11}
    # per block: first_bad marks the one BAD record, everything after it in the block is SKIPPED
    blk_of = (r["offset"] // 32768).astype(np.int64)
    for b in np.nonzero(want_blocks[:, 1] != 0xffffffff)[0][:50]:
        st = r["status"][blk_of == b]
        fb = int(want_blocks[b, 1])
        assert (st[:fb] == 1).all() and st[fb] == 0 and (st[fb + 1:] == 2).all()
    # a smaller capacity returns the leading records and the full count
    few = torch.zeros(10 * 16, dtype=torch.uint8, device="cuda")
    assert eng.wal_list(dev, len(img), 123, blocks, few, 10) == total
    torch.cuda.synchronize()
    assert (few.cpu().numpy().view(REC) == r[:10]).all()
    # host form
    arr = np.frombuffer(img, dtype=np.uint8)
    hrec = np.zeros(total, dtype=REC)
    hblocks = np.zeros((nb, 4), dtype=np.uint32)
    cnt = C.c_uint64(0)
    capi.check(capi.lib.tcrc_host_wal_list(eng.ctx, arr.ctypes.data, len(img), 123, hblocks.ctypes.data,
                                           hrec.ctypes.data, total, C.byref(cnt)))
    assert cnt.value == total and (hrec == r).all() and (hblocks == want_blocks).all()
    eng.close()


def _oracle_image(records, recycle, log_number, prefix=b""):
    """Bytes the CPU oracle writer produces for the records (after first writing `prefix` records)."""
    t = O.OracleLog(recycle=recycle, log_number=log_number)
    for r in prefix:
        t.write(r)
    start = t.written_bytes()
    for r in records:
        t.write(r)
    img = t.contents()
    t.close()
    return img[start:], start % 32768


@pytest.mark.parametrize("recycle", [False, True])
def test_device_framing_continues_after_an_image_that_ends_at_a_block_boundary(recycle):
    """A call whose image fills its last block to the byte hands back block offset 0 (the reference's writer
    starts the next block at its next record, db/log_writer.cc:58-80), and the next call takes that offset."""
    import gpu_util as G
    eng = G.engine()
    hs = 11 if recycle else 7
    src = O.splitmix(1 << 17, 5)
    first = [src[:32768 - hs].tobytes()]             # one FULL record, exactly one block
    rest = [src[100:100 + 70000].tobytes(), b"", src[7:9].tobytes()]
    dsrc = G.to_dev(src)
    want1, _ = _oracle_image(first, recycle, 9)
    want2, start = _oracle_image(rest, recycle, 9, prefix=first)
    assert len(want1) == 32768 and start == 0
    dst = torch.zeros(len(want1) + len(want2) + 8, dtype=torch.uint8, device="cuda")
    w1, bo1, p1 = eng.wal_frame(dsrc, [0], [32768 - hs], 9, recycle, dst)
    assert (w1, bo1, p1) == (32768, 0, 1)
    w2, bo2, _ = eng.wal_frame(dsrc, [100, 0, 7], [70000, 0, 2], 9, recycle, dst[w1:], block_offset=bo1)
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    assert got[:w1].tobytes() == want1 and got[w1:w1 + w2].tobytes() == want2
    assert bo2 == (w1 + w2) % 32768
    eng.close()


@pytest.mark.parametrize("recycle", [False, True])
def test_device_framing_is_byte_identical_to_the_cpu_writer(recycle):
    import ctypes as C
    import gpu_util as G
    import terarkdb_amd.capi as capi
    eng = G.engine()
    rnd = O.Random(301)
    hs = 11 if recycle else 7
    lens = [0, 1, 3, 32768 - hs, 32768 - hs + 1, 65536, 100000, 17, 32761, 32762, 5, 2040, 2040, 2040]
    lens += [rnd.skewed(17) for _ in range(300)]
    # trailers of every length 0..10 at the end of a block
    records = []
    src = O.splitmix((sum(lens) + 64 + 7) // 8 * 8, 77)
    offs, pos = [], 3  # records start at odd source offsets
    for ln in lens:
        offs.append(pos)
        records.append(src[pos:pos + ln].tobytes())
        pos += ln
    want, _ = _oracle_image(records, recycle, 123)
    dsrc = G.to_dev(src)
    arr = (capi.Desc * len(lens))()
    for i, (o, ln) in enumerate(zip(offs, lens)):
        arr[i].offset, arr[i].len = o, ln
    size = capi.lib.tcrc_wal_framed_size(arr, len(lens), int(recycle), 0)
    assert size == len(want)
    dst = torch.full((size + 100,), 0xAB, dtype=torch.uint8, device="cuda")
    written, bo, phys = eng.wal_frame(dsrc, offs, lens, 123, recycle, dst)
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    assert written == size and bo == size % 32768
    assert got[:size].tobytes() == want
    assert (got[size:] == 0xAB).all()  # nothing written past the image
    # the image verifies clean on the device and lists as many physical records as were framed
    nb = (size + 32767) // 32768
    blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    assert eng.wal_verify(dst, size, 123, blocks, counts) == phys
    assert int(counts.cpu()[0]) == 0
    # continuing an existing log: frame the second half starting at the block offset the first half left
    half = len(lens) // 2
    first_img, _ = _oracle_image(records[:half], recycle, 123)
    rest_img, start_bo = _oracle_image(records[half:], recycle, 123, prefix=records[:half])
    w1, bo1, _ = eng.wal_frame(dsrc, offs[:half], lens[:half], 123, recycle, dst)
    assert (w1, bo1) == (len(first_img), len(first_img) % 32768) and bo1 == start_bo
    dst2 = torch.zeros(len(rest_img) + 8, dtype=torch.uint8, device="cuda")
    w2, bo2, _ = eng.wal_frame(dsrc, offs[half:], lens[half:], 123, recycle, dst2, block_offset=bo1)
    torch.cuda.synchronize()
    assert w2 == len(rest_img) and dst2.cpu().numpy()[:w2].tobytes() == rest_img
    assert bo2 == (len(first_img) + len(rest_img)) % 32768
    # too small a destination is refused before anything runs
    with pytest.raises(capi.TcrcError):
        eng.wal_frame(dsrc, offs, lens, 123, recycle, dst[:size - 1])
    eng.close()


def test_device_framing_of_many_small_records():
    """More physical records than one round per lane of every wave's groups (4096 waves x 8 groups x 8 lanes):
    the header checksums are stored in bursts inside the stamp kernel's loop and once more at its end."""
    import gpu_util as G
    import terarkdb_amd.capi as capi
    eng = G.engine()
    rng = np.random.default_rng(5)
    n = 4096 * 8 * 9 + 321
    lens = rng.integers(0, 120, n).astype(np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.uint64))[:-1]]).astype(np.uint64) + 3
    src = O.splitmix((int(lens.sum()) + 64 + 7) // 8 * 8, 78)
    records = [src[int(o):int(o) + int(ln)].tobytes() for o, ln in zip(offs, lens)]
    want, _ = _oracle_image(records, False, 9)
    dst = torch.full((len(want) + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    written, bo, phys = eng.wal_frame(G.to_dev(src), offs.tolist(), lens.tolist(), 9, False, dst)
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    assert written == len(want) and bo == len(want) % 32768 and phys >= n
    assert got[:written].tobytes() == want and (got[written:] == 0xCD).all()
    eng.close()


def test_scan_of_garbage_and_preallocated_images_matches_oracle():
    """Images that are not logs at all: random bytes, zeros (a preallocated file), a valid log followed by
    zeros and by random bytes, odd sizes.  The structural walk must end every block where the CPU rules do."""
    import gpu_util as G
    eng = G.engine()
    rng = np.random.default_rng(99)
    good = np.frombuffer(O.build_wal_image(False, n_random=60, seed=3), dtype=np.uint8)
    images = [
        rng.integers(0, 256, (1 << 20) + 13, dtype=np.uint8),
        np.zeros(5 * 32768 + 100, dtype=np.uint8),
        np.concatenate([good, np.zeros(70000, dtype=np.uint8)]),
        np.concatenate([good, rng.integers(0, 256, 70001, dtype=np.uint8)]),
        rng.integers(0, 256, 6, dtype=np.uint8),          # shorter than a header
        rng.integers(0, 9, 300000, dtype=np.uint8),       # small values: plausible types and short lengths
    ]
    for k, img in enumerate(images):
        for log_number in (123, 0, int(img[7]) if img.size > 7 else 1):
            want = O.wal_scan_blocks(img, log_number)
            dev = G.to_dev(img)
            nb = (img.size + 32767) // 32768
            blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
            counts = torch.zeros(3, dtype=torch.int64, device="cuda")
            eng.counts_reset(counts)
            n = eng.wal_verify(dev, img.size, log_number, blocks, counts)
            torch.cuda.synchronize()
            got = blocks.cpu().numpy().view(np.uint32)
            assert (got == want).all(), (k, log_number, np.nonzero((got != want).any(axis=1))[0][:5])
            assert n == int(want[:, 0].sum())
    eng.close()


@pytest.mark.parametrize("phase", ["0", "1"])
@pytest.mark.parametrize("recycle", [False, True])
def test_framing_plan_is_enqueued_twice_over_other_bytes(recycle, phase):
    """plan / enqueue: the plan knows the image's size, block offset and record count before anything runs; one
    plan framed twice over different source bytes gives the CPU writer's image both times.  Both row alignments
    of the kernel (source lines, image lines) are exercised."""
    import os
    import gpu_util as G
    os.environ["TCRC_FRAME_PHASE"] = phase
    try:
        eng = G.engine()
    finally:
        os.environ.pop("TCRC_FRAME_PHASE", None)
    rnd = O.Random(17)
    hs = 11 if recycle else 7
    # whole-block fragments (sixteen pieces), fragments just above one and two pieces, empty and tiny records,
    # lengths around the sixteen-byte row edge, and a skewed mix
    lens = [32768 - hs, 0, 2048, 2049, 4096, 4097, 15, 16, 17, 31, 33, 70000, 1, 0, 0, 2047, 6000, 131071]
    lens += [rnd.skewed(15) for _ in range(400)]
    total = sum(lens)
    offs, pos = [], 5
    for ln in lens:
        offs.append(pos)
        pos += ln + (ln % 3)  # gaps between the records in the source
    plan = eng.wal_plan(offs, lens, 77, recycle, block_offset=32768 - hs - 3)
    dst = torch.full((plan.image_bytes + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    for seed in (5, 6):
        # (the second time with the caller in charge of waiting: no event behind the launch)
        plan.caller_waits(seed == 6)
        src = O.splitmix((pos + 64 + 7) // 8 * 8, seed)
        records = [src[o:o + ln].tobytes() for o, ln in zip(offs, lens)]
        # the CPU writer continuing a log that stands at the same block offset
        want, start = _oracle_image(records, recycle, 77, prefix=[b"x" * (32768 - hs - 3 - hs)])
        assert start == 32768 - hs - 3
        assert plan.image_bytes == len(want) and plan.block_offset == (start + len(want)) % 32768
        dsrc = G.to_dev(src)
        dst.fill_(0xEE)
        eng.wal_frame_enqueue(plan, dsrc, dst)
        torch.cuda.synchronize()
        got = dst.cpu().numpy()
        assert got[:len(want)].tobytes() == want, seed
        assert (got[len(want):] == 0xEE).all()
    assert plan.physical_records >= len(lens) and total > 0
    import terarkdb_amd.capi as capi
    big = torch.zeros(dsrc.numel() + plan.image_bytes + 256, dtype=torch.uint8, device="cuda")
    with pytest.raises(capi.TcrcError, match="overlaps"):
        eng.wal_frame_enqueue(plan, big[:dsrc.numel()], big[128:])
    with pytest.raises(capi.TcrcError, match="smaller"):
        eng.wal_frame_enqueue(plan, dsrc, dst[:100])
    plan.close()
    empty = eng.wal_plan([], [], 1, recycle)
    assert (empty.image_bytes, empty.block_offset, empty.physical_records) == (0, 0, 0)
    eng.wal_frame_enqueue(empty, dsrc, dst)
    empty.close()
    eng.close()


@pytest.mark.parametrize("blocks_per_workgroup,recycle", [(1, False), (2, True), (3, False), (5, True), (8, False), (17, True)])
def test_scan_with_few_blocks_per_workgroup(eng, blocks_per_workgroup, recycle):
    """The walker's lanes share the list's places by the number of blocks a workgroup has (one block: one queue of
    all places; seventeen: 32 queues): images sized for 1 to 17 blocks per workgroup, records of every size from
    tiny to several blocks, some of them damaged, block by block against the CPU reader's rules."""
    import terarkdb_amd.capi as capi
    props = torch.cuda.get_device_properties(0)
    nblocks = props.multi_processor_count * blocks_per_workgroup - 3
    rng = np.random.default_rng(100 + blocks_per_workgroup)
    filler = O.splitmix(1 << 18, 77).tobytes()
    t = O.OracleLog(recycle=recycle, log_number=41)
    while t.written_bytes() < nblocks * 32768:
        kind = int(rng.integers(0, 10))
        n = int(rng.integers(0, 60)) if kind < 4 else int(rng.integers(60, 3000)) if kind < 9 else int(rng.integers(3000, 100000))
        t.write(filler[:n])
    img = bytearray(t.contents())
    t.close()
    recs, n = O.list_physical(bytes(img), 41)
    for ri in rng.choice(n, size=40, replace=False):
        off, typ, ln, _ = (int(x) for x in recs[ri])
        img[off + int(rng.integers(0, (11 if recycle else 7) + ln))] ^= 1 << int(rng.integers(0, 8))
    img = bytes(img[:len(img) - int(rng.integers(0, 5000))])  # and a torn tail
    want = O.wal_scan_blocks(img, 41)
    got, counts, nrec = device_scan(eng, img, 41)
    assert (got == want).all()
    assert nrec == int(want[:, 0].sum()) == int(counts[capi.CNT_CHECKED])
    bad = np.nonzero(want[:, 1] != 0xffffffff)[0]
    assert int(counts[capi.CNT_FIRST_BAD]) == (int(bad[0]) if bad.size else 2**64 - 1)
