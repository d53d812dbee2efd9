"""Concurrent use of one engine from several host threads (the reference's Extend is re-entrant and is
called from write, flush, compaction and recovery threads at once: db/db_impl_write.cc:844-854)."""
import threading

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_entry_points_from_four_threads():
    import gpu_util as G
    import terarkdb_amd.capi as capi
    eng = G.engine()
    img = O.build_wal_image(False, n_random=120, seed=5)
    want_blocks = O.wal_scan_blocks(img, 123)
    errors = []

    def worker(t):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            rng = np.random.default_rng(t)
            with torch.cuda.stream(stream):
                n, blen = 3000 + 17 * t, 4096
                host = O.splitmix(n * blen, 40 + t)
                dev = G.to_dev(host)
                want = O.sst_trailer_batch(host, blen, blen, n, 0)
                out = torch.zeros(n, dtype=torch.int32, device="cuda")
                dimg = G.to_dev(np.frombuffer(img, dtype=np.uint8).copy())
                nb = (len(img) + 32767) // 32768
                blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
                counts = torch.zeros(3, dtype=torch.int64, device="cuda")
                long_out = torch.zeros(1, dtype=torch.int32, device="cuda")
                for it in range(25):
                    eng.blocks_stamp(dev, n, blen, blen, out, suffix_byte=0, flags=capi.F_MASK,
                                     stream=stream.cuda_stream)
                    eng.counts_reset(counts, stream=stream.cuda_stream)
                    eng.wal_verify(dimg, len(img), 123, blocks, counts, stream=stream.cuda_stream)
                    off, ln = int(rng.integers(0, 1000)), int(rng.integers(300000, n * blen - 1000))
                    eng.extend_long(dev, [off], [ln], [t], long_out, stream=stream.cuda_stream)
                    piece = host[off:off + 5000 + it]
                    got_small = eng.extend(7, piece)  # host-pointer path, its own stream and staging
                    stream.synchronize()
                    assert (out.cpu().numpy().view(np.uint32) == want).all()
                    assert (blocks.cpu().numpy().view(np.uint32) == want_blocks).all()
                    assert int(long_out.cpu().numpy().view(np.uint32)[0]) == O.extend(t, host[off:off + ln], "hw")
                    assert got_small == O.extend(7, piece)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    eng.close()
    assert not errors, errors


def _host_mix(eng, t, rounds, check):
    """The host-pointer calls one flush / compaction / recovery thread makes: a handful of freshly built blocks
    stamped (WriteRawBlock shape), one fetched block verified (CheckBlockChecksum shape), a 2 MiB extent of
    blocks verified after a read, and a small log image scanned.  Returns the CRC-covered bytes."""
    import ctypes as C
    import terarkdb_amd.capi as capi
    L = capi.lib
    rng = np.random.default_rng(100 + t)
    blen = 4096
    small = O.splitmix(16 * blen, 300 + t)
    extent = O.splitmix(512 * blen, 400 + t)
    img = np.frombuffer(O.build_wal_image(False, n_random=40, seed=9 + t, edge=False), dtype=np.uint8).copy()
    nb = (img.size + 32767) // 32768
    want_small = O.sst_trailer_batch(small, blen, blen, 16, 0)
    want_extent = O.sst_trailer_batch(extent, blen, blen, 512, 0)
    want_blocks = O.wal_scan_blocks(img.tobytes(), 123)

    def descs(n):
        d = np.zeros(n, dtype=[("offset", "<u8"), ("len", "<u4"), ("init", "<u4")])
        d["offset"] = np.arange(n, dtype=np.uint64) * blen
        d["len"] = blen
        return d

    d16, d512 = descs(16), descs(512)
    sfx16, sfx512 = np.zeros(16, np.uint8), np.zeros(512, np.uint8)
    out16, out512 = np.zeros(16, np.uint32), np.zeros(512, np.uint32)
    ok512 = np.zeros(512, np.uint8)
    counts = np.zeros(3, np.uint64)
    blocks = np.zeros((nb, 4), np.uint32)
    covered = 0
    for it in range(rounds):
        capi.check(L.tcrc_host_batch_stamp(eng.ctx, small.ctypes.data, small.size, d16.ctypes.data, 16,
                                           sfx16.ctypes.data, capi.F_MASK, out16.ctypes.data))
        k = int(rng.integers(0, 16))
        capi.check(L.tcrc_host_batch_verify(eng.ctx, small.ctypes.data + k * blen, blen, d16.ctypes.data, 1,
                                            sfx16.ctypes.data, 0, want_small[k:k + 1].ctypes.data, out16.ctypes.data,
                                            None, counts.ctypes.data))
        bad_verdict = int(counts[0])
        capi.check(L.tcrc_host_batch_verify(eng.ctx, extent.ctypes.data, extent.size, d512.ctypes.data, 512,
                                            sfx512.ctypes.data, 0, want_extent.ctypes.data, out512.ctypes.data,
                                            ok512.ctypes.data, counts.ctypes.data))
        capi.check(L.tcrc_host_wal_verify(eng.ctx, img.ctypes.data, img.size, 123, blocks.ctypes.data, None, None))
        covered += 17 * (blen + 1) + 512 * (blen + 1) + img.size
        if check:
            assert bad_verdict == 0 and int(counts[0]) == 0 and bool(ok512.all())
            assert (out512 == want_extent).all() and (blocks == want_blocks).all()
    return covered


def test_eight_threads_share_one_context_through_host_calls():
    """N flush/compaction threads call WriteRawBlock at once (table/block_based_table_builder.cc:580): on one
    context their host-pointer calls take separate lanes and run side by side."""
    import time
    import gpu_util as G
    eng = G.engine()
    _host_mix(eng, 0, 3, True)  # warm: staging, streams
    t0 = time.perf_counter()
    one = _host_mix(eng, 0, 40, False)
    t_one = time.perf_counter() - t0
    assert eng.stats()["host_lanes"] == 1
    errors, done = [], []

    def worker(t):
        try:
            done.append(_host_mix(eng, t, 40, True))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    t_eight = time.perf_counter() - t0
    lanes = eng.stats()["host_lanes"]
    eng.close()
    assert not errors, errors
    assert len(done) == 8 and 2 <= lanes <= 16
    print(f"\none thread {one / t_one / 1e9:.2f} GB/s, eight threads on one context {sum(done) / t_eight / 1e9:.2f} GB/s "
          f"({lanes} lanes)")


def test_two_contexts_on_one_gpu_run_side_by_side():
    """Two contexts on the same GPU (two column families' flush threads that were each given an engine of their
    own): their launches want the same compute units and take turns, and each keeps its own results."""
    import gpu_util as G
    engines = [G.engine(), G.engine()]
    errors, done = [], []

    def worker(t):
        try:
            done.append(_host_mix(engines[t % 2], t, 25, True))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    lanes = [e.stats()["host_lanes"] for e in engines]
    for e in engines:
        e.close()
    assert not errors, errors
    assert len(done) == 4 and all(1 <= n <= 2 for n in lanes), lanes


def test_eight_threads_upload_64_mib_each_from_pageable_memory():
    """Large pageable uploads from several threads share the context's pinned slots and filling threads slot by
    slot (they used to take turns at whole copies): every thread's verdicts are right, and the rates are printed -
    one thread alone, and eight at once."""
    import ctypes as C
    import time
    import gpu_util as G
    import terarkdb_amd.capi as capi
    L = capi.lib
    eng = G.engine()
    blen, n = 8192, 8192  # 64 MiB of 8 KiB blocks per thread, each block followed by [type][masked crc]
    stride = blen + 5
    images = []
    for t in range(8):
        raw = O.splitmix(n * blen, 100 + t)
        tr = O.sst_trailer_batch(raw, blen, blen, n, 0)
        img = np.zeros(n * stride, dtype=np.uint8)
        v = img.reshape(n, stride)
        v[:, :blen] = raw.reshape(n, blen)
        v[:, blen + 1:] = tr.astype("<u4").view(np.uint8).reshape(n, 4)
        bad = [(7 * t + 3) % n, n - 1 - t]
        for b in bad:
            v[b, 100 + t] ^= 0x40
        images.append((img, sorted(bad)))

    def verify(t):
        img, bad = images[t]
        counts = (C.c_uint64 * 3)()
        capi.check(L.tcrc_host_blocks_verify(eng.ctx, img.ctypes.data, n, blen, stride, 0, 0, capi.F_EXPECT_AFTER, None,
                                             None, None, counts))
        assert list(counts) == [2, bad[0], n], (t, list(counts))

    verify(0)  # warm: slots, threads, staging memory on the device
    t0 = time.perf_counter()
    for _ in range(3):
        verify(0)
    one = 3 * images[0][0].size / (time.perf_counter() - t0) / 1e9
    errors = []

    def worker(t):
        try:
            for _ in range(3):
                verify(t)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    worker_threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in worker_threads[:1]:
        pass
    # a first pass so that every lane has its device staging before the timed one
    for th in [threading.Thread(target=verify, args=(t,)) for t in range(8)]:
        th.start()
        th.join()
    t0 = time.perf_counter()
    for th in worker_threads:
        th.start()
    for th in worker_threads:
        th.join()
    eight = 8 * 3 * images[0][0].size / (time.perf_counter() - t0) / 1e9
    eng.close()
    assert not errors, errors
    print(f"\n64 MiB pageable verifies: one thread {one:.1f} GB/s, eight threads on one context {eight:.1f} GB/s aggregate")
