#!/usr/bin/env python3
"""Headline benchmark: device-resident masked CRC32C over a 1 GiB batch of 8 KiB SST blocks.

One "step" = one pass of the hot path over the batch: a single tcrc_dev_blocks_verify launch
that computes Mask(CRC32C(block || type)) for every block, stores it and compares it with
the stored trailer value.  Inputs are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 100 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

With N = 1 the corpus is the 1 GiB batch (131 072 blocks).  With N > 1 it is ONE 64 GiB corpus of the
same stream (8 388 608 blocks), sharded by contiguous block range: rank r owns one GPU and the blocks
shard_range(8 388 608, r, N), i.e. 32 / 16 / 8 GiB per GPU at N = 2 / 4 / 8 (strong scaling).  No data moves
between GPUs, and the only collective is one 24-byte all-gather (RCCL) of {mismatches, blocks checked,
first bad block} inside the timed region.  Before the timed region every rank compares the XOR of the
trailers of every GiB it owns with tests/golden/corpus64_folds.json (GiB 0: the value captured from a build
of the reference; the others: made by the CPU checker with tools/make_corpus_folds.py).

Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BLOCK_LEN = 8192
PER_GIB = 131072             # blocks per GiB
TYPE_BYTE = 0                # kNoCompression
SPLITMIX_STATE = 301
XOR_OF_MASKED = 0x0744A84E   # fold of all trailers of the first GiB, captured from a build of the reference
CORRUPT_PER_GIB = 5          # seeded single-bit corruptions per GiB of corpus (verification must find exactly these)
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak


def host_cpu_facts():
    """What the CPU baseline ran on: logical CPUs of the machine, CPUs this process may use, the share the
    box's environment states for worker pools, and the model name."""
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    affinity = len(os.sched_getaffinity(0))
    share = 0
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):
        try:
            share = int(os.environ.get(var, "0"))
        except ValueError:
            share = 0
        if share > 0:
            break
    return {"nproc": os.cpu_count(), "affinity": affinity, "stated_share": share or None, "model": model}


def cpu_baseline(seconds_budget=12.0):
    """The CPU oracle's SSE4.2+PCLMUL path on the same workload, timed on this host.
    Checker code (oracle/) is used here only as the thing a GPU number is shown beside.
    Threads: every CPU of the affinity mask, or the CPU share the box's environment states for a one-GPU job
    when that is smaller (the mask shows the whole machine there).  The rate is the best of three timed passes."""
    import oracle_lib as O
    L = O.lib()
    facts = host_cpu_facts()
    cores = facts["affinity"]
    if facts["stated_share"]:
        cores = min(cores, facts["stated_share"])
    cores = max(1, cores)
    sample_blocks = 32768  # 256 MiB of the same stream
    buf = O.splitmix(sample_blocks * BLOCK_LEN, SPLITMIX_STATE)
    out = np.empty(sample_blocks, dtype=np.uint32)
    hw = int(L.oracle_crc32c_hw_available())

    def run(nthreads, passes):
        per = sample_blocks // nthreads

        def work(t):
            base = buf.ctypes.data + t * per * BLOCK_LEN
            o = out.ctypes.data + t * per * 4
            for _ in range(passes):
                L.oracle_sst_trailer_batch(C.c_void_p(base), BLOCK_LEN, BLOCK_LEN, per, TYPE_BYTE, hw, C.c_void_p(o))

        ths = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        return time.perf_counter() - t0

    run(1, 1)  # warm
    t1 = min(run(1, 2) for _ in range(3))
    single = 2 * sample_blocks * (BLOCK_LEN + 1) / t1 / 1e9
    # three timed passes of about a fifth of the budget each
    passes = max(2, int(seconds_budget * 0.2 / max(t1 / 2 / cores, 1e-3)))
    passes = min(passes, 200)
    run(cores, 2)  # warm all threads
    times = [run(cores, passes) for _ in range(3)]
    rates = [passes * (sample_blocks // cores) * cores * (BLOCK_LEN + 1) / t / 1e9 for t in times]
    return {
        "value": round(max(rates), 2), "unit": "GB/s", "cores": cores, "kind": "port",
        "single_core_value": round(single, 2),
        "passes_gbs": [round(r, 1) for r in rates],
        "nproc": facts["nproc"], "affinity_cpus": facts["affinity"], "stated_cpu_share": facts["stated_share"],
        "cpu_model": facts["model"],
        "sample": f"{sample_blocks} x {BLOCK_LEN} B blocks (256 MiB of the same splitmix stream), best of 3 timed "
                  f"runs of {passes} passes on {cores} threads, static block-range split; "
                  f"oracle SSE4.2 crc32 + PCLMUL 3-way path (hw={hw})",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--corpus-gib", type=int, default=0,
                    help="GiB of 8 KiB blocks in the whole corpus, split over the ranks by block range "
                         "(default: 1 with one GPU = the headline batch, 64 with several)")
    ap.add_argument("--settle-seconds", type=float, default=2.5,
                    help="untimed run of the same launches before the warm-up steps (clock ramp)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Rehearsal of the N > 1 path on a box with ONE GPU: the ranks share cuda:0 and meet over gloo (RCCL refuses
    # two ranks on one device).  The sharding, the fixture checks and the collective's arithmetic are the real
    # ones; the rate is not a scaling result and the line says so.
    rehearsal = os.environ.get("TCRC_BENCH_REHEARSAL") == "1"
    dist = None
    if rehearsal:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        local_rank = 0
    elif world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        # under torchrun the process group is used even with one rank, so the same code path runs at every N
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    from terarkdb_amd.synth import splitmix_fill
    from terarkdb_amd.sharding import gather_verify_counts, shard_range
    import terarkdb_amd.capi as capi

    corpus_gib = args.corpus_gib or (1 if world == 1 else 64)
    if not 1 <= corpus_gib <= 64 * max(1, world):
        raise SystemExit("--corpus-gib out of range")
    TOTAL_BLOCKS = corpus_gib * PER_GIB
    lo, hi = shard_range(TOTAL_BLOCKS, rank, world)
    N_BLOCKS = hi - lo
    if N_BLOCKS * BLOCK_LEN > 200 << 30:
        raise SystemExit("shard does not fit one GPU")

    eng = capi.Engine(local_rank)
    # this rank's block range of the one corpus: the splitmix stream with state 301 from word lo * 1024 on
    data = torch.empty(N_BLOCKS * BLOCK_LEN, dtype=torch.uint8, device="cuda")
    splitmix_fill(data, SPLITMIX_STATE, first_word=lo * (BLOCK_LEN // 8))
    stored = torch.zeros(N_BLOCKS, dtype=torch.int32, device="cuda")
    out = torch.zeros(N_BLOCKS, dtype=torch.int32, device="cuda")
    counts = torch.zeros(capi.CNT_WORDS, dtype=torch.int64, device="cuda")

    # -- parity gate (untimed): stamp; the fold of the trailers of every GiB this rank owns whole, and the
    #    sampled trailers in its range, must equal the fixture; fresh stamps must verify clean ----------------
    eng.blocks_stamp(data, N_BLOCKS, BLOCK_LEN, BLOCK_LEN, stored, suffix_byte=TYPE_BYTE, flags=capi.F_MASK)
    torch.cuda.synchronize()
    host_tr = stored.cpu().numpy().view(np.uint32)
    fixture = json.load(open(os.path.join(ROOT, "tests", "golden", "corpus64_folds.json")))
    assert int(fixture["xor_of_masked_per_gib"][0], 16) == XOR_OF_MASKED
    gibs_checked = 0
    for g in range((lo + PER_GIB - 1) // PER_GIB, hi // PER_GIB):
        if g >= len(fixture["xor_of_masked_per_gib"]):
            break
        fold = int(np.bitwise_xor.reduce(host_tr[g * PER_GIB - lo:(g + 1) * PER_GIB - lo]))
        if fold != int(fixture["xor_of_masked_per_gib"][g], 16):
            raise SystemExit(f"parity gate failed on rank {rank}: xor of the trailers of GiB {g} is {fold:08x}, "
                             f"fixture says {fixture['xor_of_masked_per_gib'][g]}")
        gibs_checked += 1
    for b, want in fixture["trailer_of_block"].items():
        if lo <= int(b) < hi and int(host_tr[int(b) - lo]) != int(want, 16):
            raise SystemExit(f"parity gate failed on rank {rank}: trailer of block {b}")
    if gibs_checked == 0:
        raise SystemExit(f"rank {rank} owns no whole GiB of the corpus: nothing to check its trailers against")
    eng.counts_reset(counts)
    eng.blocks_verify(data, N_BLOCKS, BLOCK_LEN, BLOCK_LEN, counts, expected=stored, out=out, suffix_byte=TYPE_BYTE)
    torch.cuda.synchronize()
    c0 = counts.cpu().tolist()
    if c0[0] != 0 or c0[2] != N_BLOCKS or not torch.equal(out, stored):
        raise SystemExit(f"parity gate failed: fresh stamps do not verify ({c0})")
    # seeded corruptions over the WHOLE corpus; every rank flips the ones that fall into its range
    n_corrupt = CORRUPT_PER_GIB * corpus_gib
    bad_blocks = np.sort(np.random.default_rng(1000).choice(TOTAL_BLOCKS, n_corrupt, replace=False))
    flat = data.view(-1)
    for b in bad_blocks:
        if lo <= int(b) < hi:
            flat[(int(b) - lo) * BLOCK_LEN + 100] ^= 0x80
    torch.cuda.synchronize()

    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.blocks_verify(data, N_BLOCKS, BLOCK_LEN, BLOCK_LEN, counts, expected=stored, out=out,
                          suffix_byte=TYPE_BYTE, stream=stream)

    # Touch every other op of the timed region once first, so that no first-use set-up cost lands inside
    # it - or between the warm-up and it: a few milliseconds of idle GPU are enough for the clocks to drop,
    # and the launches after that run up to 10 % slower for the next hundred launches.
    shard_lo = lambda r: shard_range(TOTAL_BLOCKS, r, world)[0]  # noqa: E731
    step()
    on_wire = (lambda c: c.cpu()) if rehearsal else (lambda c: c)  # gloo carries host tensors
    gather_verify_counts(on_wire(counts), shard_lo)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    e1.record()
    torch.cuda.synchronize()
    _ = e0.elapsed_time(e1)
    if dist:
        dist.barrier()
    # The GPU raises its clocks only after about a second of sustained work; an untimed run of the
    # same launches first lets the timed steps see the steady state a long verification scan runs in.
    t_settle = time.perf_counter()
    chunk_ms = []
    per_chunk = max(4, 100 * PER_GIB // N_BLOCKS)  # about 17 ms of launches per chunk
    while True:
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(per_chunk):
            step()
        c1.record()
        torch.cuda.synchronize()
        chunk_ms.append(c0.elapsed_time(c1))
        waited = time.perf_counter() - t_settle
        # stop once the clocks have been up for the minimum time and the last chunks no longer speed up
        steady = len(chunk_ms) >= 4 and min(chunk_ms[-3:]) > 0.995 * min(chunk_ms[:-3])
        if (waited >= args.settle_seconds and steady) or waited >= 4 * args.settle_seconds:
            break
    for _ in range(args.warmup):
        step()
    eng.counts_reset(counts)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()

    # HIP events on the stream the kernel is launched on (torch's current stream) bracket the K launches
    ev_start, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_start.record()
    for _ in range(args.steps):
        step()
    ev_end.record()
    t_enq = time.perf_counter()
    # the path's only exchange: every rank's {mismatches, first bad block, checked}, one all-gather (RCCL)
    # of the engine's device counter block; the copy to the host waits for the K launches
    totals = gather_verify_counts(on_wire(counts), shard_lo)
    t_cnt = time.perf_counter()
    # That all-gather is the closing barrier: no rank gets its result before every rank has contributed its
    # counters, which a rank's stream produces only after its K launches.  A second collective would only add
    # its own latency to the timed region.
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if os.environ.get("BENCH_DEBUG"):
        print(f"[rank {rank}] enqueue {1e6*(t_enq-t0):.0f} us, counters back {1e6*(t_cnt-t0):.0f} us, "
              f"end {1e6*elapsed:.0f} us", file=sys.stderr)
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # The same exchange through the library's own entry point (ncclAllReduce sum + min on a communicator made
    # from an id that rank 0 hands out): outside the timed region, a cross-check of the product's collective on
    # whatever hardware this runs on.  A failure here is reported in the line, it does not void the measurement.
    product_collective = None
    if dist and not rehearsal:
        try:
            import ctypes as C
            L = capi.lib
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                raw = (C.c_uint8 * 128)()
                capi.check(L.tcrc_rccl_get_unique_id(raw))
                uid.copy_(torch.tensor(list(raw), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            raw = (C.c_uint8 * 128)(*uid.cpu().tolist())
            comm = C.c_void_p()
            capi.check(L.tcrc_rccl_comm_init_rank(eng.ctx, world, raw, rank, C.byref(comm)))
            mine = counts.clone()
            capi.check(L.tcrc_rccl_allreduce_counts(eng.ctx, comm, C.c_void_p(mine.data_ptr()), lo, C.c_void_p(stream)))
            torch.cuda.synchronize()
            got = mine.cpu().tolist()
            capi.check(L.tcrc_rccl_comm_destroy(comm))
            want = [totals.mismatches, totals.first_bad, totals.checked]
            product_collective = "ok" if got == want else f"differs: {got} != {want}"
        except Exception as e:  # noqa: BLE001
            product_collective = f"failed: {e}"

    mism, checked = totals.mismatches, totals.checked
    if mism != n_corrupt * args.steps or checked != TOTAL_BLOCKS * args.steps:
        raise SystemExit(f"verification miscounted: {mism} mismatches / {checked} checked")
    want_first = int(bad_blocks[0])
    if totals.first_bad != want_first:
        raise SystemExit(f"first bad block {totals.first_bad} != {want_first}")

    avg_ms = ev_start.elapsed_time(ev_end) / args.steps  # average launch duration over the timed region
    alg_bytes = N_BLOCKS * (BLOCK_LEN + 1)  # bytes covered by the CRC in this rank's launch: block + type byte
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    value = args.steps * TOTAL_BLOCKS * (BLOCK_LEN + 1) / elapsed / 1e9  # all ranks' units / slowest rank's time

    # Median launch duration, measured AFTER the timed region (an event pair around every launch would sit
    # between the launches of the timed steps): the same launch, each one bracketed by its own events while the
    # stream stays busy.
    n_med = max(args.steps, 20)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_med)]
    for a, b in pairs:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    median_ms = float(np.median([a.elapsed_time(b) for a, b in pairs]))

    if rank == 0:
        traffic = None
        traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and N_BLOCKS == PER_GIB:  # the counters were collected on the 1 GiB launch
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("bytes_per_launch")
                traffic_source = (f"not measured in this run: PMC passes (TCC FETCH_SIZE x2 + WRITE_SIZE) of this same "
                                  f"command kept in profiles/hbm_traffic.json, collected {tj.get('collected', 'date unknown')}")
            except Exception:
                traffic = None
        line = {
            "metric": "device-resident masked-CRC32C GB/s, 8 KiB blocks",
            "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": (f"{corpus_gib} GiB corpus of 8 KiB SST data blocks ({TOTAL_BLOCKS} blocks, type byte 0)"
                                    + (f" sharded by block range over {world} GPUs" if world > 1 else " on one GPU")
                                    + ": masked CRC32C stamped and verified against stored trailers"),
                       "settle_seconds_untimed": args.settle_seconds, "block_len": BLOCK_LEN,
                       "corpus_blocks": TOTAL_BLOCKS, "blocks_on_rank0": N_BLOCKS,
                       "sharding": "contiguous block range per GPU", "gib_folds_checked_on_rank0": gibs_checked,
                       "seeded_corruptions": n_corrupt,
                       "collective": "one all-gather of {mismatches, checked, first_bad} (24 B per rank)" if dist else "none",
                       **({"tcrc_rccl_allreduce_counts_cross_check": product_collective} if product_collective else {})},
            **({"rehearsal": f"{world} ranks sharing ONE GPU over gloo: a check of the sharded path, not a scaling result"}
               if rehearsal else {}),
            "pct_hbm_peak": round(100.0 * value / (HBM_PEAK_GBS * world), 2),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "kernel": "k_crc_batch", "avg_launch_ms": round(avg_ms, 4),
                         "median_launch_ms": round(median_ms, 4),
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    eng.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
