#!/usr/bin/env python3
"""Development aid / measurement: device log framing (tcrc_wal_plan_create + tcrc_dev_wal_frame_enqueue) over
records resident in HBM.  Prints one JSON line per shape: plan time on the host, kernel time per launch (HIP
events around K launches on the launch stream, after a settle phase), payload and image rates, and whether the
image equals the CPU writer's byte for byte.

usage: frame_probe.py [--only NAME] [--launches K] [--settle S] [--recycle]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import oracle_lib as O  # noqa: E402
import terarkdb_amd.capi as capi  # noqa: E402


def shapes(name):
    rnd = O.Random(301)
    if name == "db_bench_2040":  # the first 256 MiB of the 1M-record db_bench log: 131 108 records of 2040 bytes
        return [2040] * 131108
    if name == "mixed":
        out, tot = [], 0
        while tot < 256 << 20:
            out.append(rnd.skewed(17))
            tot += out[-1]
        return out
    if name == "small_128":
        rng = np.random.default_rng(1)
        return rng.integers(100, 161, 64 * 1000 * 1000 // 130).tolist()
    if name == "blocks_32k":
        return [32761] * 8192
    raise SystemExit(f"unknown shape {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--launches", type=int, default=30)
    ap.add_argument("--settle", type=float, default=1.0)
    ap.add_argument("--recycle", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()
    eng = capi.Engine(0)
    stream = torch.cuda.current_stream().cuda_stream
    for name in ["db_bench_2040", "mixed", "small_128", "blocks_32k"]:
        if args.only and args.only != name:
            continue
        lens = shapes(name)
        offs = np.concatenate([[0], np.cumsum(np.asarray(lens, dtype=np.uint64))[:-1]]).astype(np.uint64)
        payload = int(sum(lens))
        src = O.splitmix((payload + 64 + 7) // 8 * 8, 301)
        dsrc = torch.from_numpy(src).cuda()
        t0 = time.perf_counter()
        plan = eng.wal_plan(offs.tolist(), lens, 7, args.recycle)
        t_plan_py = time.perf_counter() - t0
        # the plan alone, without the Python loop that builds the descriptor array
        arr = (capi.Desc * len(lens))()
        a = np.frombuffer(arr, dtype=[("offset", "<u8"), ("len", "<u4"), ("init", "<u4")])
        a["offset"], a["len"] = offs, np.asarray(lens, dtype=np.uint32)
        import ctypes as C
        best_plan = None
        for _ in range(5):
            h = C.c_void_p()
            t0 = time.perf_counter()
            capi.check(capi.lib.tcrc_wal_plan_create(eng.ctx, arr, len(lens), 7, int(args.recycle), 0, C.byref(h)))
            dt = time.perf_counter() - t0
            capi.lib.tcrc_wal_plan_free(h)
            best_plan = dt if best_plan is None else min(best_plan, dt)
        dst = torch.zeros(plan.image_bytes + 64, dtype=torch.uint8, device="cuda")
        eng.wal_frame_enqueue(plan, dsrc, dst, stream=stream)
        torch.cuda.synchronize()
        same = None
        if not args.no_check:
            w = O.OracleLog(recycle=args.recycle, log_number=7)
            for o, ln in zip(offs.tolist(), lens):
                w.write(src[int(o):int(o) + ln].tobytes())
            want = w.contents()
            w.close()
            same = bool(dst.cpu().numpy()[:plan.image_bytes].tobytes() == want)
        t_end = time.perf_counter() + args.settle
        while time.perf_counter() < t_end:
            for _ in range(10):
                eng.wal_frame_enqueue(plan, dsrc, dst, stream=stream)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.launches):
            eng.wal_frame_enqueue(plan, dsrc, dst, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.launches
        # the same launches without an event behind each (the caller waits for its stream before it frees the plan)
        plan.caller_waits(True)
        for _ in range(20):
            eng.wal_frame_enqueue(plan, dsrc, dst, stream=stream)
        u0, u1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        u0.record()
        for _ in range(args.launches):
            eng.wal_frame_enqueue(plan, dsrc, dst, stream=stream)
        u1.record()
        torch.cuda.synchronize()
        ms_untracked = u0.elapsed_time(u1) / args.launches
        plan.caller_waits(False)
        # for scale: the runtime's own device-to-device copy of the payload (no headers, no checksums)
        flat = torch.empty(payload, dtype=torch.uint8, device="cuda")
        for _ in range(5):
            flat.copy_(dsrc[:payload])
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(args.launches):
            flat.copy_(dsrc[:payload])
        c1.record()
        torch.cuda.synchronize()
        copy_ms = c0.elapsed_time(c1) / args.launches
        del flat
        # the one-call form, host clock, to completion
        written, bo, phys = C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
        best_call = None
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            capi.check(capi.lib.tcrc_dev_wal_frame(eng.ctx, dsrc.data_ptr(), arr, len(lens), 7, int(args.recycle), 0,
                                                   dst.data_ptr(), dst.numel(), C.byref(written), C.byref(bo),
                                                   C.byref(phys), C.c_void_p(stream)))
            dt = time.perf_counter() - t0
            best_call = dt if best_call is None else min(best_call, dt)
        print(json.dumps({
            "shape": name, "recycle": args.recycle, "logical_records": len(lens), "physical_records": plan.physical_records,
            "payload_bytes": payload, "image_bytes": plan.image_bytes,
            "kernel_us_per_launch": round(ms * 1e3, 2), "us_per_launch_caller_waits": round(ms_untracked * 1e3, 2), "payload_GBps": round(payload / (ms * 1e-3) / 1e9, 1),
            "traffic_GBps_read_plus_write": round((payload + plan.image_bytes) / (ms * 1e-3) / 1e9, 1),
            "frac_hbm_peak_read_plus_write": round((payload + plan.image_bytes) / (ms * 1e-3) / 8e12, 4),
            "plain_device_copy_us": round(copy_ms * 1e3, 2),
            "plain_device_copy_GBps_read_plus_write": round(2 * payload / (copy_ms * 1e-3) / 1e9, 1),
            "plan_create_us": round(best_plan * 1e6, 1), "one_call_us": round(best_call * 1e6, 1),
            "python_plan_us": round(t_plan_py * 1e6, 1),
            "bytes_identical_to_cpu_writer": same, "launches": args.launches,
        }), flush=True)
        plan.close()
        del dsrc, dst
    eng.close()


if __name__ == "__main__":
    main()
