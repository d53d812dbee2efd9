"""Development aid: randomised parity soak of the device entry points against the CPU checker, for a time
budget.  Every case draws its own shape from the case number, so a failure names the case to replay:

    python tools/soak.py SECONDS [FIRST_CASE]

Cases rotate over: log scans (size distributions from tiny records to whole-block fragments and the skewed
mix, both header kinds, clean and damaged, whole and truncated images), descriptor batches (any offset, any
length, initial values, suffix bytes, masked or not, in-image expectations), uniform blocks, and long buffers
(both entry points, random segment lengths), the host-pointer entry points with sizes around the points
where the engine changes route, device-side framing followed by a scan and a record list, the
log::Writer / log::Reader classes beside the CPU writer and reader, whole-file table verification,
whole-file checksums through buffers of several sizes, and block trailers in the shape a table builder hands
them over.  Prints a progress line every few seconds; exits non-zero at the
first difference."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import terarkdb_amd.capi as capi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
case = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = capi.Engine(0)
FILL = O.splitmix(1 << 20, 11).tobytes()
done = {"log": 0, "batch": 0, "blocks": 0, "long": 0, "host": 0, "frame": 0, "classes": 0, "table": 0, "file": 0, "trailers": 0}


def log_sizes(kind, rng, total):
    out, acc = [], 0
    while acc < total:
        if kind == 0:
            n = int(rng.integers(0, 64))
        elif kind == 1:
            n = int(rng.integers(0, 1200))
        elif kind == 2:
            n = int(rng.choice([2040, 2041, 2042, 2047, 2048, 2049, 4089, 4090, 4096, 4097, 6137, 6138, 32761, 32762]))
        elif kind == 3:
            n = int(rng.integers(1, 1 << int(rng.integers(1, 18))))  # the reference test's skewed shape
        elif kind == 4:
            n = int(rng.integers(100000, 400000))
        else:
            n = int(rng.choice([0, 1, 5, 30000, 32761, 32768, 70000])) if rng.random() < 0.5 else int(rng.integers(0, 64))
        out.append(n)
        acc += n + 7
    return out


def case_log(c, rng):
    recycle = bool(rng.integers(0, 2))
    kind = int(rng.integers(0, 6))
    total = int(rng.integers(1, 1 << int(rng.integers(12, 26))))
    if kind == 0:
        total = min(total, 4 << 20)
    lognum = int(rng.integers(1, 1 << 20))
    t = O.OracleLog(recycle=recycle, log_number=lognum)
    for n in log_sizes(kind, rng, total):
        o = int(rng.integers(0, len(FILL) - n)) if n < len(FILL) else 0
        t.write(FILL[o:o + n])
    img = np.frombuffer(t.contents(), dtype=np.uint8).copy()
    t.close()
    mode = int(rng.integers(0, 4))
    if mode == 1 and img.size > 10:  # cut somewhere
        img = img[:int(rng.integers(1, img.size))].copy()
    if mode >= 2:  # damage
        for pos in rng.integers(0, img.size, int(rng.integers(1, 200))):
            img[int(pos)] ^= 1 << int(rng.integers(0, 8))
    if img.size == 0:
        return
    nb = (img.size + 32767) // 32768
    want = np.zeros((nb, 4), dtype=np.uint32)
    O.lib().oracle_wal_scan_blocks(img.ctypes.data, img.size, lognum, want.ctypes.data)
    dev = torch.from_numpy(img).cuda()
    blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(capi.CNT_WORDS, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    n = eng.wal_verify(dev, img.size, lognum, blocks, counts)
    torch.cuda.synchronize()
    got = blocks.cpu().numpy().view(np.uint32)
    cn = counts.cpu().numpy().view(np.uint64)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, ("log", c, kind, recycle, mode, img.size, bad[:5], got[bad[:5]], want[bad[:5]])
    assert n == int(want[:, 0].sum()), ("log records", c, n, int(want[:, 0].sum()))
    assert int(cn[capi.CNT_MISMATCH]) < (1 << 40), ("log scan gave up waiting", c)
    firsts = np.nonzero(want[:, 1] != 0xffffffff)[0]
    if firsts.size:
        assert int(cn[capi.CNT_FIRST_BAD]) == int(firsts[0]), ("first bad", c)
    else:
        assert int(cn[capi.CNT_MISMATCH]) == 0, ("mismatch on a clean image", c)
    done["log"] += 1


def case_batch(c, rng):
    size = int(rng.integers(64, 1 << int(rng.integers(8, 25)))) & ~7
    host = O.splitmix(max(size, 64), int(rng.integers(1, 1000)))
    n = int(rng.integers(1, 1 << int(rng.integers(1, 15))))
    maxlen = int(rng.choice([16, 130, 1000, 2100, 5000, 20000, 300000]))
    if maxlen > 20000:
        n = min(n, 512)  # keeps the checker's share of a case within a second
    lens = np.minimum(rng.integers(0, maxlen + 1, n), host.size - 8).astype(np.uint32)
    offs = (rng.integers(0, host.size - 8, n) % (host.size - 8 - lens + 1)).astype(np.uint64)
    inits = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32) if rng.random() < 0.5 else np.zeros(n, np.uint32)
    suffix = rng.integers(0, 256, n).astype(np.uint8) if rng.random() < 0.5 else None
    masked = bool(rng.integers(0, 2))
    flags = (capi.F_MASK if masked else 0) | (capi.F_SHORT_SPANS if (maxlen <= 1000 and rng.random() < 0.5) else 0)
    want = O.batch_extend(host, offs, lens, inits, suffix, masked)
    dev = torch.from_numpy(host).cuda()
    desc = capi.make_desc_tensor(offs, lens, inits, "cuda")
    dsuf = None if suffix is None else torch.from_numpy(suffix).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    eng.batch_stamp(dev, desc, n, out, suffix=dsuf, flags=flags)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, ("batch stamp", c, n, maxlen, bad[:5], offs[bad[:5]], lens[bad[:5]])
    # verify against expectations (stored values are always the masked ones) with a few wrong ones
    want = O.batch_extend(host, offs, lens, inits, suffix, True)
    exp = want.copy()
    wrong = rng.integers(0, n, int(rng.integers(0, 4)))
    exp[wrong] ^= 0x10
    counts = torch.zeros(capi.CNT_WORDS, dtype=torch.int64, device="cuda")
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    eng.counts_reset(counts)
    eng.batch_verify(dev, desc, n, counts, expected=torch.from_numpy(exp.view(np.int32)).cuda(), ok=ok, suffix=dsuf,
                     flags=flags)
    torch.cuda.synchronize()
    okh = ok.cpu().numpy()
    cn = counts.cpu().numpy().view(np.uint64)
    assert ((okh != 0) == (exp == want)).all(), ("batch verify verdicts", c)
    nbad = int((exp != want).sum())
    assert int(cn[capi.CNT_MISMATCH]) == nbad and int(cn[capi.CNT_CHECKED]) == n, ("batch verify counts", c, cn)
    if nbad:
        assert int(cn[capi.CNT_FIRST_BAD]) == int(np.nonzero(exp != want)[0][0]), ("batch first bad", c)
    done["batch"] += 1


def case_blocks(c, rng):
    blen = int(rng.choice([1, 15, 16, 17, 100, 1000, 1408, 1409, 4096, 4101, 8192, 8197, 16384, 33000]))
    pad = int(rng.choice([0, 0, 5, 1, 64, 123]))
    stride = blen + pad
    n = int(rng.integers(1, max(2, min(1 << 15, (48 << 20) // stride))))
    lead = int(rng.integers(0, 64))
    host = O.splitmix(((lead + n * stride + 64) + 7) & ~7, int(rng.integers(1, 1000)))
    tb = int(rng.integers(0, 256)) if rng.random() < 0.7 else -1
    if tb >= 0:
        want = O.sst_trailer_batch(host[lead:], blen, stride, n, tb)
    else:
        want = O.batch_extend(host, lead + np.arange(n, dtype=np.uint64) * stride, np.full(n, blen, np.uint32), None, None, True)
    dev = torch.from_numpy(host).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    eng.blocks_stamp(dev[lead:], n, blen, stride, out, suffix_byte=tb, flags=capi.F_MASK)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, ("blocks", c, blen, stride, n, lead, tb, bad[:5])
    if pad >= 5 and tb >= 0:
        # trailers stored in place, then verified in place
        eng.blocks_stamp(dev[lead:], n, blen, stride, out, suffix_byte=tb, flags=capi.F_MASK | capi.F_STORE_AFTER)
        counts = torch.zeros(capi.CNT_WORDS, dtype=torch.int64, device="cuda")
        eng.counts_reset(counts)
        eng.blocks_verify(dev[lead:], n, blen, stride, counts, suffix_byte=tb, flags=capi.F_EXPECT_AFTER)
        torch.cuda.synchronize()
        img = dev.cpu().numpy()
        tr = np.stack([img[lead + np.arange(n) * stride + blen + 1 + k] for k in range(4)], axis=1).astype(np.uint32)
        stored = tr[:, 0] | (tr[:, 1] << 8) | (tr[:, 2] << 16) | (tr[:, 3] << 24)
        assert (stored == want).all(), ("blocks stored trailers", c, blen, stride)
        assert (img[lead + np.arange(n) * stride + blen] == tb).all(), ("blocks stored type byte", c)
        cn = counts.cpu().numpy().view(np.uint64)
        assert int(cn[capi.CNT_MISMATCH]) == 0 and int(cn[capi.CNT_CHECKED]) == n, ("blocks verify in place", c, cn)
    done["blocks"] += 1


def case_long(c, rng):
    n = int(rng.choice([1, 1, 2, 7, 16, 64, 65, 300]))
    maxlen = int(rng.choice([1000, 100000, 1 << 20, 5 << 20, 40 << 20]))
    if n > 16:
        maxlen = min(maxlen, 1 << 20)
    size = min(96 << 20, max(4096, maxlen * min(n, 4)))
    host = O.splitmix(size & ~7, int(rng.integers(1, 1000)))
    lens = np.minimum(rng.integers(0, maxlen + 1, n), host.size - 1).astype(np.uint32)
    offs = (rng.integers(0, host.size, n) % (host.size - lens + 1)).astype(np.uint64)
    inits = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    seg = int(rng.choice([0, 0, 256, 512, 1024, 4096, 65536, 1 << 20]))
    noalloc = n <= 64 and rng.random() < 0.5
    want = O.batch_extend(host, offs, lens, inits, None, False)
    dev = torch.from_numpy(host).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    eng.extend_long(dev, offs.tolist(), lens.tolist(), inits.tolist(), out, segment_len=seg, noalloc=noalloc)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, ("long", c, n, seg, noalloc, bad[:5], offs[bad[:5]], lens[bad[:5]])
    done["long"] += 1


def around(rng, centre, spread):
    return max(0, int(centre + rng.integers(-spread, spread + 1)))


def case_host(c, rng):
    """Host-pointer entry points, with sizes drawn around the points where the engine changes route: 64 KiB and
    1 MiB of input (zero-copy block, in-place reads of pinned memory), 4096 items, 80 KiB spans (long-buffer
    route), 16 KiB for Extend, 16 MiB (staged upload from pageable memory)."""
    import ctypes as C
    L = capi.lib
    what = int(rng.integers(0, 4))
    pinned = bool(rng.integers(0, 2))
    if what == 0:  # Extend over one buffer
        n = around(rng, int(rng.choice([0, 16, 16 << 10, 64 << 10, 1 << 20, 5 << 20, 16 << 20])), 40)
        init = int(rng.integers(0, 1 << 32))
        host = O.splitmix((n + 64) & ~7, int(rng.integers(1, 1000)))
        lead = int(rng.integers(0, 32))
        handle = None
        buf = host
        if pinned:
            handle = L.tcrc_host_alloc(eng.ctx, host.size)
            buf = np.ctypeslib.as_array(C.cast(handle, C.POINTER(C.c_uint8)), shape=(host.size,))
            buf[:] = host
        out = C.c_uint32(0)
        capi.check(L.tcrc_extend(eng.ctx, init, C.c_void_p(buf.ctypes.data + lead), n, C.byref(out)))
        if handle:
            L.tcrc_host_free(eng.ctx, C.c_void_p(handle))
        assert out.value == O.extend(init, host[lead:lead + n], "hw"), ("host extend", c, n, lead, pinned)
    elif what == 3:  # a log image
        total = around(rng, int(rng.choice([1000, 32768, 65536, 1 << 20, 16 << 20])), 3000)
        recycle = bool(rng.integers(0, 2))
        t = O.OracleLog(recycle=recycle, log_number=9)
        kind = int(rng.integers(0, 6))
        for n in log_sizes(kind, rng, min(max(total, 8), (2 << 20) if kind == 0 else (1 << 30))):
            o = int(rng.integers(0, len(FILL) - n)) if n < len(FILL) else 0
            t.write(FILL[o:o + n])
        img = np.frombuffer(t.contents(), dtype=np.uint8).copy()
        t.close()
        if img.size == 0:
            return
        for pos in rng.integers(0, img.size, int(rng.integers(0, 5))):
            img[int(pos)] ^= 1 << int(rng.integers(0, 8))
        nb = (img.size + 32767) // 32768
        want = np.zeros((nb, 4), dtype=np.uint32)
        O.lib().oracle_wal_scan_blocks(img.ctypes.data, img.size, 9, want.ctypes.data)
        got = np.zeros((nb, 4), dtype=np.uint32)
        counts = np.zeros(3, np.uint64)
        nrec = C.c_uint64(0)
        capi.check(L.tcrc_host_wal_verify(eng.ctx, C.c_void_p(img.ctypes.data), img.size, 9, C.c_void_p(got.ctypes.data),
                                          C.c_void_p(counts.ctypes.data), C.byref(nrec)))
        assert (got == want).all(), ("host log", c, img.size)
        assert nrec.value == int(want[:, 0].sum()), ("host log records", c)
    else:  # descriptor batches, stamped (what == 1) or verified against the values stored behind them (what == 2)
        shape = int(rng.integers(0, 5))
        if shape == 0:    # one fetched block
            n, maxlen = 1, int(rng.choice([4096, 8192, 65536, 81920, 300000]))
        elif shape == 1:  # a handful of meta blocks, one of them long
            n, maxlen = int(rng.integers(2, 9)), int(rng.choice([5000, 81919, 81920, 400000]))
        elif shape == 2:  # around 4096 items
            n, maxlen = around(rng, 4096, 3), 200
        elif shape == 3:  # around 1 MiB of input
            n, maxlen = int(rng.integers(10, 300)), 9000
        else:             # an extent of blocks above the staging threshold
            n, maxlen = int(rng.integers(3000, 6000)), 4200
        lens = rng.integers(maxlen // 2, maxlen + 1, n).astype(np.uint32)
        gap = 5 + int(rng.integers(0, 3))
        offs = (np.cumsum(lens.astype(np.uint64) + gap) - lens - gap).astype(np.uint64) + int(rng.integers(0, 16))
        total = int(offs[-1] + lens[-1] + gap)
        if shape == 3:
            total = max(total, around(rng, 1 << 20, 64))
        if shape == 0 and rng.random() < 0.5:
            total = max(total, around(rng, 64 << 10, 8))
        host = O.splitmix((total + 15) & ~7, int(rng.integers(1, 1000)))[:total]
        inits = np.zeros(n, np.uint32) if what == 2 else rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        use_suffix = shape != 1 and rng.random() < 0.6
        sfx = rng.integers(0, 256, n).astype(np.uint8) if use_suffix else None
        want = O.batch_extend(host, offs, lens, inits, sfx, True)
        d = (capi.Desc * n)()
        for i in range(n):
            d[i].offset, d[i].len, d[i].init = int(offs[i]), int(lens[i]), int(inits[i])
        handle = None
        buf = host.copy()
        if pinned:
            handle = L.tcrc_host_alloc(eng.ctx, total)
            buf = np.ctypeslib.as_array(C.cast(handle, C.POINTER(C.c_uint8)), shape=(total,))
            buf[:] = host
        sp = None if sfx is None else C.c_void_p(sfx.ctypes.data)
        if what == 1:
            out = np.zeros(n, np.uint32)
            capi.check(L.tcrc_host_batch_stamp(eng.ctx, C.c_void_p(buf.ctypes.data), total, d, n, sp, capi.F_MASK,
                                               C.c_void_p(out.ctypes.data)))
            bad = np.nonzero(out != want)[0]
            assert bad.size == 0, ("host stamp", c, shape, n, total, pinned, bad[:5])
        else:
            # stored values behind every span (behind its suffix byte when there is one), a few of them wrong
            wrong = set(rng.integers(0, n, int(rng.integers(0, 3))).tolist())
            for i in range(n):
                at = int(offs[i] + lens[i])
                if sfx is not None:
                    buf[at] = sfx[i]
                    at += 1
                v = int(want[i]) ^ (0x100 if i in wrong else 0)
                buf[at:at + 4] = np.frombuffer(np.uint32(v).tobytes(), np.uint8)
            ok = np.full(n, 7, np.uint8)
            got = np.zeros(n, np.uint32)
            counts = np.zeros(3, np.uint64)
            capi.check(L.tcrc_host_batch_verify(eng.ctx, C.c_void_p(buf.ctypes.data), total, d, n, sp, capi.F_EXPECT_AFTER,
                                                None, C.c_void_p(got.ctypes.data), C.c_void_p(ok.ctypes.data),
                                                C.c_void_p(counts.ctypes.data)))
            assert (got == want).all(), ("host verify values", c, shape, n, total, pinned)
            verdict = np.array([0 if i in wrong else 1 for i in range(n)], np.uint8)
            assert (ok == verdict).all(), ("host verify verdicts", c, shape, n, total, pinned)
            assert int(counts[0]) == len(wrong) and int(counts[2]) == n, ("host verify counts", c, counts)
            if wrong:
                assert int(counts[1]) == min(wrong), ("host verify first bad", c)
        if handle:
            L.tcrc_host_free(eng.ctx, C.c_void_p(handle))
    done["host"] += 1


REC = np.dtype([("offset", "<u8"), ("length", "<u4"), ("type", "u1"), ("status", "u1"), ("header_size", "<u2")])


def case_frame(c, rng):
    """Records framed on the device, in two calls (the second continues at the block offset the first left),
    against the bytes of the CPU writer; then damaged, scanned and listed record by record."""
    recycle = bool(rng.integers(0, 2))
    lognum = int(rng.integers(1, 1 << 31))
    kind = int(rng.integers(0, 6))
    total = int(rng.integers(1, 1 << int(rng.integers(10, 23))))
    lens = log_sizes(kind, rng, min(total, (1 << 20) if kind == 0 else (1 << 30)))
    src = O.splitmix((sum(lens) + 64 + 7) & ~7, int(rng.integers(1, 1000)))
    offs, pos = [], int(rng.integers(0, 8))
    for ln in lens:
        offs.append(pos)
        pos += ln
    t = O.OracleLog(recycle=recycle, log_number=lognum)
    half = int(rng.integers(0, len(lens) + 1))
    for i in range(half):
        t.write(src[offs[i]:offs[i] + lens[i]].tobytes())
    first = t.written_bytes()
    for i in range(half, len(lens)):
        t.write(src[offs[i]:offs[i] + lens[i]].tobytes())
    want = np.frombuffer(t.contents(), dtype=np.uint8).copy()
    t.close()
    dsrc = torch.from_numpy(src).cuda()
    dst = torch.full((want.size + 64,), 0xAB, dtype=torch.uint8, device="cuda")
    w1, bo1, p1 = eng.wal_frame(dsrc, offs[:half], lens[:half], lognum, recycle, dst)
    assert (w1, bo1) == (first, first % 32768), ("frame first part", c, w1, bo1, first)
    w2, bo2, p2 = eng.wal_frame(dsrc, offs[half:], lens[half:], lognum, recycle, dst[first:], block_offset=bo1)
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    assert w1 + w2 == want.size and bo2 == want.size % 32768, ("frame sizes", c, w1, w2, want.size)
    assert (got[:want.size] == want).all(), ("frame bytes", c, kind, recycle, half, int(np.nonzero(got[:want.size] != want)[0][0]))
    assert (got[want.size:] == 0xAB).all(), ("frame wrote past its image", c)
    if want.size == 0:
        done["frame"] += 1
        return
    # damage, scan, list
    img = want.copy()
    for at in rng.integers(0, img.size, int(rng.integers(0, 20))):
        img[int(at)] ^= 1 << int(rng.integers(0, 8))
    dev = torch.from_numpy(img).cuda()
    nb = (img.size + 32767) // 32768
    want_blocks = O.wal_scan_blocks(img, lognum)
    accepted, n_acc = O.list_physical(img.tobytes(), lognum)
    blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(capi.CNT_WORDS, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    n = eng.wal_verify(dev, img.size, lognum, blocks, counts)
    assert n == int(want_blocks[:, 0].sum()), ("frame scan records", c)
    recs = torch.zeros(max(n, 1) * 16, dtype=torch.uint8, device="cuda")
    assert eng.wal_list(dev, img.size, lognum, blocks, recs, n) == n, ("list count", c)
    torch.cuda.synchronize()
    r = recs.cpu().numpy().view(REC)[:n]
    ok = r[r["status"] == 1]
    assert len(ok) == n_acc, ("list accepted", c, len(ok), n_acc)
    assert (ok["offset"] == accepted[:, 0]).all() and (ok["type"] == accepted[:, 1]).all() and \
        (ok["length"] == accepted[:, 2]).all(), ("list records", c)
    if img.size == want.size and (img == want).all():
        assert n == p1 + p2 and n_acc == n, ("clean image", c, n, p1, p2)
    done["frame"] += 1


def case_classes(c, rng):
    """The engine's log::Writer and log::Reader (through the C wrappers) beside the CPU writer and reader: the
    same records, the same file bytes; then the same damage to both files and the same records, dropped bytes
    and reports out of both readers, under every recovery mode."""
    import engine_log
    recycle = bool(rng.integers(0, 2))
    lognum = int(rng.integers(1, 1 << 20))
    manual = bool(rng.integers(0, 2))
    bulk = bool(rng.integers(0, 2))
    kind = int(rng.integers(0, 6))
    total = int(rng.integers(1, 1 << int(rng.integers(8, 21))))
    lens = log_sizes(kind, rng, min(total, (1 << 17) if kind == 0 else (1 << 30)))
    e = engine_log.EngineLog(recycle=recycle, log_number=lognum, ctx=eng.ctx, manual_flush=manual, bulk_image=bulk)
    o = O.OracleLog(recycle=recycle, log_number=lognum)
    for n in lens:
        at = int(rng.integers(0, len(FILL) - n)) if n < len(FILL) else 0
        e.write(FILL[at:at + n])
        o.write(FILL[at:at + n])
    if manual:
        e.flush()
    a, b = e.contents(), o.contents()
    assert a == b, ("writer bytes", c, kind, recycle, manual, len(a), len(b))
    size = len(a)
    what = int(rng.integers(0, 3))
    if what == 1 and size:
        for at in rng.integers(0, size, int(rng.integers(1, 6))):
            d = int(rng.integers(1, 256))
            e.increment_byte(int(at), d)
            o.increment_byte(int(at), d)
    elif what == 2 and size > 1:
        cut = int(rng.integers(1, min(size, 70000)))
        e.shrink_size(cut)
        o.shrink_size(cut)
    # (the reference's reader never gets past an old record when it skips every corrupted one, and a damaged
    # type byte can make any record an old one: that mode only where no byte was changed)
    mode = int(rng.integers(0, 3 if what == 1 else 4))
    for k in range(len(lens) + 3):
        ra, rb = e.read(mode), o.read(mode)
        assert ra == rb, ("reader record", c, k, kind, recycle, what, mode, bulk,
                          ra if ra == "EOF" else len(ra), rb if rb == "EOF" else len(rb))
        if ra == "EOF":
            break
    assert e.dropped_bytes() == o.dropped_bytes(), ("dropped bytes", c, e.dropped_bytes(), o.dropped_bytes())
    assert e.report_message() == o.report_message(), ("report", c, e.report_message(), o.report_message())
    e.close()
    o.close()
    done["classes"] += 1


def case_table(c, rng):
    """Whole-file verification of a block-based table written by the test-file writer (every index and footer
    variant), clean or with a few flipped bits anywhere in front of the footer: the status text must be the one
    the reference's visiting order gives (first failing block), worked out on the CPU."""
    import tempfile
    import sst_cases, sst_files as S, sst_verify_util as U
    kw = dict(sst_cases.VARIANTS[int(rng.integers(0, len(sst_cases.VARIANTS)))])
    if kw.get("n_data", 40) > 1:
        kw["n_data"] = int(rng.integers(2, 120))
    kw["seed"] = int(rng.integers(1, 1 << 30))
    kw["block_bytes"] = int(rng.choice([300, 1000, 4096, 16384]))
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "%06d.sst" % (c % 1000000))
        lay = S.build_table(path, **kw)
        size = os.path.getsize(path)
        footer = 48 if kw.get("legacy_footer") else 53
        if size > footer:
            for _ in range(int(rng.integers(0, 4))):
                S.flip(path, int(rng.integers(0, size - footer)), int(rng.integers(0, 8)))
        want = S.expected_status(lay)
        rc, msg, st = U.verify_with_engine(eng)(path, extent_bytes=int(rng.choice([0, 1 << 16, 1 << 20])),
                                                read_threads=int(rng.choice([0, 1, 4])))
        assert msg == want, ("table status", c, kw, msg, want)
        assert rc == (0 if want == "OK" else 1), ("table rc", c, rc)
    done["table"] += 1


def case_file(c, rng):
    """Whole-file crc32c the way the backup engine computes it: a file of a random size read through buffers of a
    random size (sizes around the buffer size and its multiples), with and without a size limit."""
    import ctypes as C
    import tempfile
    L = capi.lib
    buf = int(rng.choice([4096, 65536, 1 << 20, 5 << 20]))
    size = max(0, int(rng.choice([0, 1, buf - 1, buf, buf + 1, 2 * buf, 3 * buf + 17, int(rng.integers(0, 4 * buf))])))
    data = O.splitmix((size + 8) & ~7, int(rng.integers(1, 1000)))[:size]
    limit = 0 if rng.random() < 0.5 else int(rng.integers(1, size + 10))
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "f.bin")
        data.tofile(path)
        crc, got_size = C.c_uint32(0), C.c_uint64(0)
        msg = C.create_string_buffer(512)
        rc = L.tcrc_shim_file_checksum(eng.ctx, path.encode(), limit, buf, C.byref(crc), C.byref(got_size), msg, 512)
        assert rc == 0, ("file checksum", c, msg.value)
        n = size if limit == 0 else min(size, limit)
        assert got_size.value == n, ("file size", c, got_size.value, n)
        assert crc.value == O.value(data[:n], "hw"), ("file crc", c, size, buf, limit)
    done["file"] += 1


def case_trailers(c, rng):
    """Block trailers as WriteRawBlock forms them, for the blocks a table builder hands over in one go: sizes from
    empty to hundreds of KiB (a filter or index block among data blocks), any type byte."""
    L = capi.lib
    n = int(rng.integers(1, 1 << int(rng.integers(1, 13))))
    big = int(rng.choice([200, 9000, 9000, 120000, 600000]))
    sizes = rng.integers(0, big, size=n).astype(np.uint64)
    if big > 9000:
        sizes[rng.integers(0, n, max(1, n // 50))] = int(rng.integers(80000, big))
        sizes[sizes > 9000] *= (rng.random(int((sizes > 9000).sum())) < 0.05) | (np.arange(int((sizes > 9000).sum())) < 3)
    types = rng.integers(0, 256, size=n).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    base = O.splitmix((int(sizes.sum()) + 16) & ~7, int(rng.integers(1, 1000)))
    out = np.zeros(5 * n, dtype=np.uint8)
    rc = L.tcrc_shim_block_trailers(eng.ctx, base.ctypes.data, offs.ctypes.data, sizes.ctypes.data, types.ctypes.data, n, 1,
                                    out.ctypes.data)
    assert rc == 0, ("trailers rc", c, rc)
    want = O.batch_extend(base, offs, sizes.astype(np.uint32), None, types, True)
    tr = out.reshape(n, 5)
    got = tr[:, 1].astype(np.uint32) | (tr[:, 2].astype(np.uint32) << 8) | (tr[:, 3].astype(np.uint32) << 16) | \
        (tr[:, 4].astype(np.uint32) << 24)
    assert (tr[:, 0] == types).all(), ("trailer type bytes", c)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, ("trailer values", c, n, bad[:5], sizes[bad[:5]])
    done["trailers"] += 1


CASES = [case_log, case_batch, case_host, case_blocks, case_log, case_long, case_host, case_frame, case_classes,
         case_table, case_file, case_trailers]
t_end = time.perf_counter() + budget
t_print = time.perf_counter() + 5
while time.perf_counter() < t_end:
    rng = np.random.default_rng(1000003 * case + 17)
    CASES[case % len(CASES)](case, rng)
    case += 1
    if time.perf_counter() > t_print:
        print(f"case {case}: {done}", flush=True)
        t_print = time.perf_counter() + 5
print(f"soak clean: next case {case}; {done}", flush=True)
eng.close()
