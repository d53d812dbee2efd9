#!/usr/bin/env python3
"""Measures the configurations other than the headline one on a GPU box and writes one JSON
document (kept under profiles/): the CPU-reference log configuration, device-resident and
host-inclusive log verification, host-inclusive block stamping, the mixed-size log, and long
buffers.  GB/s = CRC-covered bytes / time, 1 GB = 1e9 B.

usage: python tools/measure_paths.py [out.json]
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G  # noqa: E402
import oracle_lib as O  # noqa: E402
import terarkdb_amd.capi as capi  # noqa: E402

L = capi.lib
OL = O.lib()


def wall(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def dev_time(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def pinned_copy(eng, arr):
    p = L.tcrc_host_alloc(eng.ctx, arr.size)
    C.memmove(p, arr.ctypes.data, arr.size)
    return p


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "paths.json")
    eng = capi.Engine(0)
    res = {"device": eng.describe(), "host_cpus_used": 1}

    # ---- config 1: CPU reference, 1M db_bench-shaped records through the log writer/reader ----------
    n_rec = int(os.environ.get("MP_RECORDS", "1000000"))
    t = O.OracleLog(recycle=False, log_number=7, hw=True)
    t0 = time.perf_counter()
    OL.oracle_wal_write_dbbench_records(t.writer, n_rec, 301)
    t_write = time.perf_counter() - t0
    size = t.written_bytes()
    phys = OL.oracle_wal_writer_physical_records(t.writer)
    crc_bytes = OL.oracle_wal_writer_crc_bytes(t.writer)
    total = C.c_uint64()
    t0 = time.perf_counter()
    got = OL.oracle_wal_read_all(t.reader, 0, C.byref(total))
    t_read = time.perf_counter() - t0
    assert got == n_rec
    img = np.frombuffer((C.c_uint8 * size).from_address(OL.oracle_sink_data(t.sink)), dtype=np.uint8)
    blocks_cpu = np.zeros(((size + 32767) // 32768, 4), dtype=np.uint32)
    t0 = time.perf_counter()
    OL.oracle_wal_scan_blocks(img.ctypes.data, size, 7, blocks_cpu.ctypes.data)  # table path, one thread
    t_scan_table = time.perf_counter() - t0
    res["config1_cpu_reference"] = {
        "logical_records": n_rec, "physical_records": int(phys), "file_bytes": int(size),
        "crc_covered_bytes": int(crc_bytes),
        "writer_seconds": round(t_write, 3), "writer_GBps": round(crc_bytes / t_write / 1e9, 2),
        "writer_note": "includes generating the 2000-byte values with the reference's Random",
        "reader_seconds": round(t_read, 3), "reader_GBps": round(crc_bytes / t_read / 1e9, 2),
        "kind": "oracle port, SSE4.2+PCLMUL path, 1 thread",
    }
    if n_rec == 1000000:
        assert (int(phys), int(size), int(crc_bytes)) == (1062207, 2047436274, 2041062207), (phys, size, crc_bytes)
        res["config1_cpu_reference"]["matches_captured_reference_facts"] = True

    # ---- config 1 on the device: verify the same image ------------------------------------------------
    dev = torch.from_numpy(img.copy()).cuda()
    nb = (size + 32767) // 32768
    dblocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")

    def dv():
        eng.counts_reset(counts)
        return eng.wal_verify(dev, size, 7, dblocks, counts)

    nrec_dev = dv()
    torch.cuda.synchronize()
    assert nrec_dev == phys
    assert (dblocks.cpu().numpy().view(np.uint32) == blocks_cpu).all()
    t_dev = wall(dv, reps=5)
    res["config1_device_log_verify"] = {
        "seconds": round(t_dev, 6), "GBps": round(crc_bytes / t_dev / 1e9, 1),
        "pct_hbm_peak": round(100 * crc_bytes / t_dev / 8e12, 1),
        "note": "one call to completion (counter reset, the fused scan kernel, record count back through pinned memory), wall clock, data resident in HBM; tools/measure_configs.py has the sustained rate",
    }
    hblocks = np.zeros((nb, 4), dtype=np.uint32)
    hcounts = np.zeros(3, dtype=np.uint64)

    def hv(ptr):
        rc = L.tcrc_host_wal_verify(eng.ctx, ptr, size, 7, hblocks.ctypes.data, hcounts.ctypes.data, None)
        assert rc == 0

    t_pageable = wall(lambda: hv(img.ctypes.data), reps=2)
    pin = pinned_copy(eng, img)
    t_pinned = wall(lambda: hv(pin), reps=3)
    assert (hblocks == blocks_cpu).all()
    res["config1_host_log_verify"] = {
        "pageable_GBps": round(crc_bytes / t_pageable / 1e9, 2), "pinned_GBps": round(crc_bytes / t_pinned / 1e9, 2),
        "note": "host image -> H2D -> device verify -> per-block results D2H",
    }
    L.tcrc_host_free(eng.ctx, pin)

    # ---- config 1 written on the device: logical records resident in HBM -> framed log image -----------
    lst_all, n_all = O.list_physical(bytes(img[: min(size, 256 * 1024 * 1024)]), 7)
    recs_dev, cur_len, rec_offs, rec_lens = [], 0, [], []
    # logical records = FULL records and FIRST..LAST runs of the oracle image; rebuild them back to back
    pieces, run = [], []
    for off, typ, ln, _ in lst_all:
        frag = img[int(off) + 7: int(off) + 7 + int(ln)]
        if typ == 1:
            pieces.append([frag])
        elif typ == 2:
            run = [frag]
        elif typ == 3:
            run.append(frag)
        elif typ == 4:
            pieces.append(run + [frag])
            run = []
    flat_src = np.concatenate([f for p_ in pieces for f in p_])
    pos = 0
    for p_ in pieces:
        ln = int(sum(len(f) for f in p_))
        rec_offs.append(pos)
        rec_lens.append(ln)
        pos += ln
    dsrc = torch.from_numpy(flat_src).cuda()
    arr = (capi.Desc * len(rec_offs))()
    for i_, (o_, l_) in enumerate(zip(rec_offs, rec_lens)):
        arr[i_].offset, arr[i_].len = o_, l_
    fsize = L.tcrc_wal_framed_size(arr, len(rec_offs), 0, 0)
    ddst = torch.zeros(fsize, dtype=torch.uint8, device="cuda")
    wr, bo_, ph_ = C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
    best = None
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        capi.check(L.tcrc_dev_wal_frame(eng.ctx, dsrc.data_ptr(), arr, len(rec_offs), 7, 0, 0, ddst.data_ptr(), fsize,
                                        C.byref(wr), C.byref(bo_), C.byref(ph_), None))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    same_frame = bool((ddst.cpu().numpy() == img[:fsize]).all())
    assert same_frame
    res["config1_device_log_frame"] = {
        "logical_records": len(rec_offs), "physical_records": int(ph_.value), "image_bytes": int(fsize),
        "seconds": round(best, 6), "GBps": round(fsize / best / 1e9, 1), "bytes_identical_to_cpu_writer": same_frame,
        "note": "records resident in HBM -> fragment plan on the host (lengths only) -> copy + headers + CRCs on the device",
    }

    # ---- config 1 through the engine's writer in group-commit mode --------------------------------
    f = L.tcrc_shim_memfile_new()
    w = L.tcrc_shim_wal_writer_new(eng.ctx, f, 7, 0, 1, 0)
    # logical records again, back to back (regenerated the same way through a scratch oracle writer)
    recs = np.empty(n_rec * 2040, dtype=np.uint8)
    rnd = O.Random(301)
    # take the payloads back out of the oracle image instead of regenerating them in Python
    lst, n_list = O.list_physical(bytes(img[: min(size, 40 * 1024 * 1024)]), 7)
    sample_records = []
    cur = b""
    for off, typ, ln, _ in lst:
        frag = bytes(img[int(off) + 7: int(off) + 7 + int(ln)])
        if typ == 1:
            sample_records.append(frag)
        elif typ == 2:
            cur = frag
        elif typ == 3:
            cur += frag
        elif typ == 4:
            sample_records.append(cur + frag)
    k = min(len(sample_records), 15000)
    flat = np.frombuffer(b"".join(sample_records[:k]), dtype=np.uint8)
    t0 = time.perf_counter()
    assert L.tcrc_shim_wal_writer_add_records_fixed(w, flat.ctypes.data, 2040, k) == 0
    assert L.tcrc_shim_wal_writer_write_buffer(w) == 0
    t_gc = time.perf_counter() - t0
    wsize = L.tcrc_shim_memfile_size(f)
    same = C.string_at(L.tcrc_shim_memfile_data(f), wsize) == bytes(img[:wsize])
    assert same
    # later groups on the same writer reuse its staging (the first one pays for growing it)
    steady = []
    for _ in range(4):
        t0 = time.perf_counter()
        assert L.tcrc_shim_wal_writer_add_records_fixed(w, flat.ctypes.data, 2040, k) == 0
        t1 = time.perf_counter()
        assert L.tcrc_shim_wal_writer_write_buffer(w) == 0
        steady.append((time.perf_counter() - t0, t1 - t0))
    best = min(steady)
    res["config1_engine_writer_group_commit"] = {
        "records": k, "seconds": round(t_gc, 4), "GBps": round(k * 2040 / t_gc / 1e9, 2),
        "steady_state_seconds": round(best[0], 4), "steady_state_GBps": round(k * 2040 / best[0] / 1e9, 2),
        "steady_state_host_framing_seconds": round(best[1], 4),
        "bytes_identical_to_cpu_writer": bool(same),
        "note": "AddRecord x k framed on the host into pinned staging, one device submission at WriteBuffer(); "
                "the first group also grows the staging buffer",
    }
    # group size at which one device submission beats the CPU writer on one core
    cross = []
    for kg in (1, 4, 16, 64, 256, 1024, 4096, min(k, 15000)):
        best_dev = best_cpu = None
        for _ in range(5):
            t0 = time.perf_counter()
            assert L.tcrc_shim_wal_writer_add_records_fixed(w, flat.ctypes.data, 2040, kg) == 0
            assert L.tcrc_shim_wal_writer_write_buffer(w) == 0
            dt = time.perf_counter() - t0
            best_dev = dt if best_dev is None else min(best_dev, dt)
        tc = O.OracleLog(recycle=False, log_number=7, hw=True)
        for _ in range(5):
            t0 = time.perf_counter()
            OL.oracle_wal_writer_add_records_fixed(tc.writer, flat.ctypes.data, 2040, kg)
            dt = time.perf_counter() - t0
            best_cpu = dt if best_cpu is None else min(best_cpu, dt)
        tc.close()
        cross.append({"records": kg, "bytes": kg * 2040, "engine_us": round(best_dev * 1e6, 1),
                      "cpu_one_core_us": round(best_cpu * 1e6, 1)})
    res["group_commit_crossover"] = {"rows": cross, "note": "2040-byte records per group; engine = framing on the host "
                                     "+ one device submission + append to an in-memory file; cpu = the oracle writer "
                                     "(SSE4.2 path) appending to an in-memory file, one thread"}
    t0 = time.perf_counter()
    w2 = L.tcrc_shim_wal_writer_new(eng.ctx, L.tcrc_shim_memfile_new(), 7, 0, 0, 0)
    kk = 2000
    assert L.tcrc_shim_wal_writer_add_records_fixed(w2, flat.ctypes.data, 2040, kk) == 0
    t_sync = time.perf_counter() - t0
    res["config1_engine_writer_per_record"] = {
        "records": kk, "microseconds_per_record": round(t_sync / kk * 1e6, 1),
        "note": "manual_flush off: every AddRecord is its own device round trip (latency-bound)",
    }
    del dev, dblocks
    torch.cuda.empty_cache()

    # ---- config 2 with the copies: 1 GiB of 8 KiB blocks from host memory ----------------------------
    nblk, blen = 131072, 8192
    host = O.splitmix(nblk * blen, 301)
    hout = np.zeros(nblk, dtype=np.uint32)

    def hs(ptr):
        assert L.tcrc_host_blocks_stamp(eng.ctx, ptr, nblk, blen, blen, 0, 0, capi.F_MASK, hout.ctypes.data) == 0

    t_pg = wall(lambda: hs(host.ctypes.data), reps=2)
    assert int(np.bitwise_xor.reduce(hout)) == 0x0744A84E
    pin = pinned_copy(eng, host)
    t_pin = wall(lambda: hs(pin), reps=3)
    assert int(np.bitwise_xor.reduce(hout)) == 0x0744A84E
    L.tcrc_host_free(eng.ctx, pin)
    res["config2_host_inclusive"] = {
        "pageable_GBps": round(nblk * (blen + 1) / t_pg / 1e9, 2), "pinned_GBps": round(nblk * (blen + 1) / t_pin / 1e9, 2),
        "note": "H2D of 1 GiB + kernel + D2H of 512 KiB of trailers; bounded by the host link (PCIe Gen5 x16, 63 GB/s spec)",
    }

    # ---- config 2 with the block sizes a table builder really produces: 4 KiB and 8 KiB blocks that overshoot
    #      by up to one entry, packed back to back with their 5-byte trailers (descriptor list, general path) -----------
    for target in (4096, 8192):
        rng2 = np.random.default_rng(target)
        nblk = (1 << 30) // (target + 133)
        # a block is closed once it passes block_size (table/flush_block_policy.cc:34-67): size in
        # [block_size, block_size + one entry)
        lens = (target + rng2.integers(0, 257, nblk)).astype(np.uint32)
        offs = np.concatenate([[0], np.cumsum(lens.astype(np.uint64) + 5)[:-1]]).astype(np.uint64)
        total = int(offs[-1] + lens[-1] + 5)
        image = G.splitmix_dev((total + 7) // 8 * 8, 31)
        types = torch.zeros(nblk, dtype=torch.uint8, device="cuda")
        d_stamp = capi.make_desc_tensor(offs, lens, np.zeros(nblk, np.uint32), "cuda")
        d_check = capi.make_desc_tensor(offs, lens + 1, np.zeros(nblk, np.uint32), "cuda")
        vcounts = torch.zeros(3, dtype=torch.int64, device="cuda")

        def t_stamp():
            eng.batch_stamp(image, d_stamp, nblk, None, suffix=types, flags=capi.F_MASK | capi.F_STORE_AFTER)

        def t_verify():
            eng.batch_verify(image, d_check, nblk, vcounts, flags=capi.F_EXPECT_AFTER)

        out_t = {}
        for name, fn in (("stamp_in_place", t_stamp), ("verify", t_verify)):
            for _ in range(30):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 50
            out_t[name + "_us"] = round(ms * 1e3, 1)
            out_t[name + "_GBps"] = round(float(lens.sum() + nblk) / ms / 1e6, 1)
        eng.counts_reset(vcounts)
        t_verify()
        cc = vcounts.cpu().numpy()
        assert int(cc[0]) == 0 and int(cc[2]) == nblk
        sample = rng2.integers(0, nblk, 200)
        img_h = image.cpu().numpy()
        for i_ in sample:
            o_, l_ = int(offs[i_]), int(lens[i_])
            want_crc = O.mask(O.extend(O.value(img_h[o_:o_ + l_]), b"\0"))
            assert int(img_h[o_ + l_ + 1:o_ + l_ + 5].copy().view("<u4")[0]) == want_crc
        out_t.update({"blocks": int(nblk), "image_bytes": total, "mean_block": target})
        res[f"config2_variable_blocks_{target}"] = out_t
        del image, img_h

    # ---- config 3: mixed-size log, >= 256 MiB, both header kinds, with seeded bit flips ------------
    for recycle in (0, 1):
        tl = O.OracleLog(recycle=bool(recycle), log_number=123)
        rnd = O.Random(301)
        piece = O.build_wal_image(bool(recycle), n_random=0)  # the edge-case records
        while tl.written_bytes() < 256 * 1024 * 1024:
            tl.write(O.big_string(b"%d." % tl.written_bytes(), rnd.skewed(17)))
        size3 = tl.written_bytes()
        img3 = np.frombuffer((C.c_uint8 * size3).from_address(OL.oracle_sink_data(tl.sink)), dtype=np.uint8).copy()
        rng = np.random.default_rng(recycle)
        for _ in range(1000):
            img3[int(rng.integers(0, size3))] ^= 1 << int(rng.integers(0, 8))
        nb3 = (size3 + 32767) // 32768
        want = np.zeros((nb3, 4), dtype=np.uint32)
        t0 = time.perf_counter()
        OL.oracle_wal_scan_blocks(img3.ctypes.data, size3, 123, want.ctypes.data)
        t_cpu3 = time.perf_counter() - t0
        d3 = torch.from_numpy(img3).cuda()
        b3 = torch.zeros((nb3, 4), dtype=torch.int32, device="cuda")

        def v3():
            eng.counts_reset(counts)
            eng.wal_verify(d3, size3, 123, b3, counts)

        v3()
        torch.cuda.synchronize()
        got = b3.cpu().numpy().view(np.uint32)
        assert (got == want).all()
        t3 = wall(v3, reps=5)
        covered = int(want[:, 2].sum())  # bytes walked, close to the covered bytes
        res[f"config3_mixed_log_{'recyclable' if recycle else 'legacy'}"] = {
            "image_bytes": int(size3), "records": int(want[:, 0].sum()),
            "blocks_with_bad_record": int((want[:, 1] != 0xffffffff).sum()),
            "identical_to_cpu_reader_rules": True,
            "device_seconds": round(t3, 6), "device_GBps": round(covered / t3 / 1e9, 1),
            "cpu_table_path_seconds": round(t_cpu3, 3),
        }
        tl.close()
        del d3, b3

    # ---- config 4: long buffers ----------------------------------------------------------------------
    S = O.splitmix(64 * 4 * 1024 * 1024, 301)
    dS = torch.from_numpy(S).cuda()
    long_res = {}
    for sz in (256 * 1024, 1024 * 1024, 4 * 1024 * 1024):
        o1 = torch.zeros(1, dtype=torch.int32, device="cuda")
        d1 = (capi.Desc * 1)()
        d1[0].offset, d1[0].len, d1[0].init = 0, sz, 0
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def run1():
            capi.check(L.tcrc_dev_extend_long(eng.ctx, C.c_void_p(dS.data_ptr()), d1, 1, 0,
                                              C.c_void_p(o1.data_ptr()), stream))

        t1 = wall(run1, reps=20)
        assert int(o1.cpu().numpy().view(np.uint32)[0]) == O.value(S[:sz], "hw")
        offs = [i * sz for i in range(64)]
        o64 = torch.zeros(64, dtype=torch.int32, device="cuda")
        d64 = (capi.Desc * 64)()
        for i in range(64):
            d64[i].offset, d64[i].len, d64[i].init = offs[i], sz, 0

        def run64():
            capi.check(L.tcrc_dev_extend_long(eng.ctx, C.c_void_p(dS.data_ptr()), d64, 64, 0,
                                              C.c_void_p(o64.data_ptr()), stream))

        t64 = wall(run64, reps=20)
        want = O.batch_extend(S, np.array(offs, np.uint64), np.full(64, sz, np.uint32))
        assert (o64.cpu().numpy().view(np.uint32) == want).all()
        long_res[str(sz)] = {"single_us": round(t1 * 1e6, 1), "single_GBps": round(sz / t1 / 1e9, 1),
                             "batch64_us": round(t64 * 1e6, 1), "batch64_GBps": round(64 * sz / t64 / 1e9, 1)}
    res["config4_long_buffers"] = long_res
    res["config4_note"] = "host call to completion, 20 calls after an idle period (clocks down, Python binding); tools/measure_configs.py and tools/floor_probe.hip have the settled figures; device-resident data"

    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
