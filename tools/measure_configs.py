#!/usr/bin/env python3
"""Measures the configurations other than the headline one (that is bench.py) on a GPU box, the way bench.py
measures: inputs resident in HBM, an untimed settle phase of the same launches (the clocks come up only after
about a second of sustained work), then K launches bracketed by events on the launch stream.  Prints ONE JSON
line per configuration with the same "roofline" keys as bench.py's line; every result is first checked against
the CPU oracle (checker only: nothing here times it).

    python tools/measure_configs.py [--only NAME[,NAME...]] [--launches K] [--settle SECONDS]

Names: log_2k (the 1M-record db_bench log, 2.05 GB), log_mixed, log_mixed_recyclable (256 MiB, sizes
Random(301).Skewed(17)), blocks4k_var, blocks8k_var (1 GiB of table-builder-sized blocks packed with their
trailers: verify against, and stamp into, the image), spans_2k (the batch kernel alone on 2046-byte spans at
every alignment), long_64x4m (64 buffers of 4 MiB in one call), long_256m and long_1g (one buffer of 256 MiB, of
1 GiB), long_single (one buffer of 256 KiB / 1 MiB / 4 MiB per call, host call to completion; each long shape
also with the calls enqueued back to back), log_small_128 / log_small_512 / log_small_1k
(64 MB logs of records of 100..160, 400..600 and 900..1100 bytes: the header walk is what they wait for), frame_2040
(device framing of 131 108 records of 2040 bytes into a 256 MiB log image: payload read once, image written once).
Under rocprofv3 run one name at a time.
"""
import argparse
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

PEAK = 8000.0
OL = O.lib()
ARGS = None


def sustained(fn):
    """average launch time in seconds over ARGS.launches launches after the settle phase"""
    t_end = time.perf_counter() + ARGS.settle
    while True:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        if time.perf_counter() >= t_end:
            break
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ARGS.launches):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ARGS.launches * 1e-3


def line(name, workload, kernel, alg_bytes, seconds, **extra):
    gbps = alg_bytes / seconds / 1e9
    doc = {"config": name, "workload": workload, "launches": ARGS.launches, "settle_seconds_untimed": ARGS.settle,
           "roofline": {"bound": "hbm", "achieved": round(gbps, 1), "peak": PEAK, "unit": "GB/s",
                        "frac": round(gbps / PEAK, 4), "traffic": None, "kernel": kernel,
                        "avg_launch_us": round(seconds * 1e6, 2), "algorithmic_bytes_per_launch": int(alg_bytes)}}
    doc.update(extra)
    print(json.dumps(doc), flush=True)


def log_image(kind):
    if kind == "2k":
        t = O.OracleLog(recycle=False, log_number=7, hw=True)
        OL.oracle_wal_write_dbbench_records(t.writer, 1000000, 301)
        size = t.written_bytes()
        covered = int(OL.oracle_wal_writer_crc_bytes(t.writer))
        lognum = 7
    elif kind.startswith("small:"):
        lo, hi = (int(x) for x in kind.split(":")[1:])
        rng = np.random.default_rng(1)
        filler = O.splitmix(1 << 16, 9).tobytes()
        t = O.OracleLog(recycle=False, log_number=9)
        acc = 0
        while acc < 64 << 20:
            n = int(rng.integers(lo, hi + 1))
            t.write(filler[:n])
            acc += n + 7
        size = t.written_bytes()
        covered = None
        lognum = 9
    else:
        recycle = kind == "mixed_recyclable"
        t = O.OracleLog(recycle=recycle, log_number=123)
        O.write_edge_records(t, recycle)
        rnd = O.Random(301)
        while t.written_bytes() < 256 * 1024 * 1024:
            t.write(O.big_string(b"%d." % t.written_bytes(), rnd.skewed(17)))
        size = t.written_bytes()
        covered = None
        lognum = 123
    img = np.frombuffer((C.c_uint8 * size).from_address(OL.oracle_sink_data(t.sink)), dtype=np.uint8).copy()
    t.close()
    return img, lognum, covered


def run_log(eng, name, kind, workload):
    img, lognum, covered = log_image(kind)
    nb = (img.size + 32767) // 32768
    want = np.zeros((nb, 4), np.uint32)
    OL.oracle_wal_scan_blocks(img.ctypes.data, img.size, lognum, want.ctypes.data)
    if covered is None:
        recs, _ = O.list_physical(img.tobytes(), lognum)
        covered = int(sum(int(r[2]) + (5 if int(r[1]) >= 5 else 1) for r in recs))  # payload + type (+ log number)
    dev = torch.from_numpy(img).cuda()
    blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    n = eng.wal_verify(dev, img.size, lognum, blocks, counts)
    torch.cuda.synchronize()
    assert (blocks.cpu().numpy().view(np.uint32) == want).all() and n == int(want[:, 0].sum())
    sec = sustained(lambda: eng.wal_verify(dev, img.size, lognum, blocks, counts, want_records=False))
    line(name, workload, "k_wal_split", covered, sec, image_bytes=int(img.size), physical_records=int(n),
         records_in_the_fullest_block=int(want[:, 0].max()), per_block_results_equal_cpu_reader_rules=True)


def run_blocks_var(eng, name, target):
    rng = np.random.default_rng(target)
    nblk = (1 << 30) // (target + 133)
    # a block is closed once it passes block_size (table/flush_block_policy.cc:34-67)
    lens = (target + rng.integers(0, 257, nblk)).astype(np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.uint64) + 5)[:-1]]).astype(np.uint64)
    total = int(offs[-1] + lens[-1] + 5)
    image = G.splitmix_dev((total + 7) // 8 * 8, 31)
    types = torch.zeros(nblk, dtype=torch.uint8, device="cuda")
    desc = capi.make_desc_tensor(offs, lens, np.zeros(nblk, np.uint32), "cuda")
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    stamp = lambda: eng.batch_stamp(image, desc, nblk, None, suffix=types, flags=capi.F_MASK | capi.F_STORE_AFTER)  # noqa: E731
    verify = lambda: eng.batch_verify(image, desc, nblk, counts, suffix=types, flags=capi.F_EXPECT_AFTER)  # noqa: E731
    stamp()
    eng.counts_reset(counts)
    verify()
    cc = counts.cpu().numpy()
    assert int(cc[0]) == 0 and int(cc[2]) == nblk
    img_h = image.cpu().numpy()
    for i in rng.integers(0, nblk, 200):
        o, l = int(offs[i]), int(lens[i])
        assert int(img_h[o + l + 1:o + l + 5].copy().view("<u4")[0]) == O.mask(O.extend(O.value(img_h[o:o + l]), b"\0"))
    del img_h
    alg = int(lens.sum()) + nblk
    what = f"1 GiB of {target}..{target + 256}-byte blocks packed with their 5-byte trailers ({nblk} blocks, descriptor list)"
    if ARGS.phase in ("", "verify"):
        line(name + "_verify", what + ": verify against the trailers in the image", "k_crc_batch<8>", alg, sustained(verify))
    if ARGS.phase in ("", "stamp"):
        line(name + "_stamp", what + ": stamp the trailers into the image", "k_crc_batch<8>", alg, sustained(stamp))


def run_spans(eng):
    total = 1 << 30
    data = G.splitmix_dev(total, 3)
    L, stride = 2046, 2053
    n = (total - L) // stride
    offs = np.arange(n, dtype=np.uint64) * stride
    desc = capi.make_desc_tensor(offs, np.full(n, L, np.uint32), np.zeros(n, np.uint32), "cuda")
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    fn = lambda: eng.batch_stamp(data, desc, n, out, flags=capi.F_MASK)  # noqa: E731
    fn()
    torch.cuda.synchronize()
    host = data[:64 * stride + L].cpu().numpy()
    want = O.batch_extend(host, offs[:64], np.full(64, L, np.uint32), np.zeros(64, np.uint32), None, True)
    assert (out[:64].cpu().numpy().view(np.uint32) == want).all()
    line("spans_2k", f"{n} spans of {L} bytes at a stride of {stride} (every alignment), descriptor list",
         "k_crc_batch<8>", n * L, sustained(fn))


def run_frame(eng):
    n, ln = 131108, 2040  # the first 256 MiB of the db_bench log
    lens = np.full(n, ln, np.uint32)
    offs = np.arange(n, dtype=np.uint64) * ln
    payload = n * ln
    src = O.splitmix((payload + 64 + 7) // 8 * 8, 301)
    dsrc = torch.from_numpy(src).cuda()
    plan = eng.wal_plan(offs.tolist(), lens.tolist(), 7, False)
    dst = torch.zeros(plan.image_bytes + 64, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    fn = lambda: eng.wal_frame_enqueue(plan, dsrc, dst, stream=stream)  # noqa: E731
    fn()
    torch.cuda.synchronize()
    w = O.OracleLog(recycle=False, log_number=7)
    for o in offs.tolist():
        w.write(src[int(o):int(o) + ln].tobytes())
    assert dst.cpu().numpy()[:plan.image_bytes].tobytes() == w.contents()
    w.close()
    sec_tracked = sustained(fn)  # an event behind every launch: the plan may be freed at any time
    plan.caller_waits(True)     # the loop below waits for its stream before the plan is closed
    sec = sustained(fn)
    torch.cuda.synchronize()
    line("frame_2040", f"frame {n} records of {ln} bytes resident in HBM into a log image ({plan.physical_records} physical "
                       f"records, {plan.image_bytes} bytes): one kernel reads the payload once, writes the image once and "
                       "checksums what passes through its registers; the plan is made once, outside the launches",
         "k_wal_frame", payload + plan.image_bytes, sec, payload_bytes=payload, image_bytes=int(plan.image_bytes),
         payload_GBps=round(payload / sec / 1e9, 1), image_equals_cpu_writer=True,
         note="launches queued with tcrc_wal_plan_caller_waits(plan, 1): no event behind each launch",
         avg_launch_us_with_an_event_behind_every_launch=round(sec_tracked * 1e6, 2))
    plan.close()


def run_long(eng, which):
    """which: "single" (256 KiB, 1 MiB, 4 MiB, one buffer per call), "64x4m", "256m", "1g" (the asymptote)"""
    shapes = {"single": (1, (256 << 10, 1 << 20, 4 << 20)), "64x4m": (64, (4 << 20,)), "256m": (1, (256 << 20,)),
              "1g": (1, (1 << 30,))}[which]
    S = O.splitmix((1 << 30) if which == "1g" else (256 << 20), 301)
    dS = torch.from_numpy(S).cuda()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = capi.lib
    for n, sizes in (shapes,):
        for sz in sizes:
            d = (capi.Desc * n)()
            for i in range(n):
                d[i].offset, d[i].len, d[i].init = i * sz, sz, 0
            out = torch.zeros(n, dtype=torch.int32, device="cuda")
            run = lambda: capi.check(L.tcrc_dev_extend_long(eng.ctx, C.c_void_p(dS.data_ptr()), d, n, 0,  # noqa: E731
                                                             C.c_void_p(out.data_ptr()), stream))
            run()
            want = O.batch_extend(S, np.arange(n, dtype=np.uint64) * sz, np.full(n, sz, np.uint32))
            assert (out.cpu().numpy().view(np.uint32) == want).all()
            t_end = time.perf_counter() + ARGS.settle
            while time.perf_counter() < t_end:
                run()
            t0 = time.perf_counter()
            for _ in range(ARGS.launches):
                run()
            sec = (time.perf_counter() - t0) / ARGS.launches  # the call returns with the result complete
            name = f"long_{n}x{sz >> 10}k"
            line(name, f"{n} buffer(s) of {sz} bytes per tcrc_dev_extend_long call, host call to completion "
                       "(one kernel launch and one stream synchronisation)", "k_crc_batch<8> (segments)", n * sz, sec,
                 note="wall clock per call: includes the launch and the wait for completion")
            # the same calls enqueued back to back (tcrc_dev_extend_long_enqueue), one wait at the end
            enq = lambda: capi.check(L.tcrc_dev_extend_long_enqueue(eng.ctx, C.c_void_p(dS.data_ptr()), d, n, 0,  # noqa: E731
                                                                     C.c_void_p(out.data_ptr()), stream))
            out.zero_()
            enq()
            torch.cuda.synchronize()
            assert (out.cpu().numpy().view(np.uint32) == want).all()
            line(name + "_enqueued", f"{n} buffer(s) of {sz} bytes per tcrc_dev_extend_long_enqueue call, calls "
                                     "enqueued back to back, HIP events around the row of calls",
                 "k_crc_batch<8> (segments)", n * sz, sustained(enq))


def main():
    global ARGS
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--launches", type=int, default=50)
    ap.add_argument("--settle", type=float, default=2.0)
    ap.add_argument("--phase", default="", help="blocks*_var only: verify or stamp (both when empty)")
    ARGS = ap.parse_args()
    only = set(x for x in ARGS.only.split(",") if x)
    eng = capi.Engine(0)
    want = lambda n: not only or n in only  # noqa: E731
    if want("log_2k"):
        run_log(eng, "log_2k", "2k", "verify the log of 1 000 000 db_bench-shaped records (2040-byte write batches, "
                                    "1 062 207 physical records, 2 047 436 274 bytes), one scan per launch")
    if want("log_mixed"):
        run_log(eng, "log_mixed", "mixed", "verify a 256 MiB log of records sized Random(301).Skewed(17) behind the "
                                           "edge-case records, legacy headers")
    if want("log_mixed_recyclable"):
        run_log(eng, "log_mixed_recyclable", "mixed_recyclable", "the same with recyclable headers")
    for nm, lo, hi in (("log_small_128", 100, 160), ("log_small_512", 400, 600), ("log_small_1k", 900, 1100)):
        if want(nm):
            run_log(eng, nm, f"small:{lo}:{hi}", f"verify a 64 MB log of records of {lo}..{hi} bytes, legacy headers")
    if want("frame_2040"):
        run_frame(eng)
    if want("blocks4k_var"):
        run_blocks_var(eng, "blocks4k_var", 4096)
    if want("blocks8k_var"):
        run_blocks_var(eng, "blocks8k_var", 8192)
    if want("spans_2k"):
        run_spans(eng)
    for nm in ("64x4m", "256m", "1g", "single"):
        if want("long_" + nm):
            run_long(eng, nm)
    eng.close()


if __name__ == "__main__":
    main()
