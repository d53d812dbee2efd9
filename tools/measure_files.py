#!/usr/bin/env python3
"""Host-memory-to-result rates of the file-level drivers on a GPU box: whole-file table verification,
whole-file checksum and the command-line tool, on files that sit in the page cache.

    python tools/measure_files.py [out.json]
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (loads the HIP runtime the engine library links against)
import oracle_lib as O  # noqa: E402
import sst_files as S  # noqa: E402
import sst_verify_util as U  # noqa: E402
import terarkdb_amd.capi as capi  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "files.json")
    n_blocks = int(os.environ.get("MF_BLOCKS", "131072"))
    eng = capi.Engine(0)
    res = {"device": eng.describe()}
    tmp = tempfile.mkdtemp(prefix="tcrc_files_")
    path = os.path.join(tmp, "000042.sst")
    t0 = time.perf_counter()
    lay = S.build_uniform_table(path, n_blocks)
    res["table_file"] = {"bytes": lay.file_size, "data_blocks": n_blocks, "build_seconds": round(time.perf_counter() - t0, 2)}
    verify = U.verify_with_engine(eng)
    runs = []
    for threads in (1, 4, 8, 16):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            rc, msg, st = verify(path, read_threads=threads)
            dt = time.perf_counter() - t0
            assert (rc, msg) == (0, "OK"), msg
            best = dt if best is None else min(best, dt)
        runs.append({"read_threads": threads, "seconds": round(best, 4), "GBps": round(st[4] / best / 1e9, 2),
                     "engine_calls": st[6]})
        print("verify", runs[-1], flush=True)
    res["sst_verify"] = {"covered_bytes": st[4], "blocks": st[1] + st[2] + st[3], "runs": runs}
    # seeded corruptions: the driver must name the first one in index order
    rng = np.random.default_rng(4)
    bad = np.sort(rng.choice(n_blocks, 5, replace=False))
    for b in bad:
        S.flip(path, int(b) * 8197 + 1234, 5)
    rc, msg, st = verify(path)
    raw0 = np.fromfile(path, dtype=np.uint8, count=8197, offset=int(bad[0]) * 8197)
    want = "Corruption: block checksum mismatch: expected %d, got %d  in %s offset %d size 8192" % (
        O.value(raw0[:8193]), O.unmask(int(raw0[8193:8197].view("<u4")[0])), path, int(bad[0]) * 8197)
    assert rc == 1 and msg == want and st[5] == 5, (msg, want, st)
    res["sst_verify"]["corrupted_run"] = {"mismatches": st[5], "first_status_matches_oracle": True}

    # whole-file checksum (backup engine shape)
    crc, size = C.c_uint32(0), C.c_uint64(0)
    m = C.create_string_buffer(256)
    want_crc = O.value(np.fromfile(path, dtype=np.uint8), variant="hw")
    runs = []
    for bs in (5 << 20, 64 << 20, 256 << 20):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            rc = capi.lib.tcrc_shim_file_checksum(eng.ctx, path.encode(), 0, bs, C.byref(crc), C.byref(size), m, 256)
            dt = time.perf_counter() - t0
            assert rc == 0 and crc.value == want_crc
            best = dt if best is None else min(best, dt)
        runs.append({"buffer_bytes": bs, "seconds": round(best, 4), "GBps": round(size.value / best / 1e9, 2)})
        print("file_checksum", runs[-1], flush=True)
    t0 = time.perf_counter()
    O.value(np.fromfile(path, dtype=np.uint8), variant="hw")
    res["file_checksum"] = {"bytes": size.value, "runs": runs,
                            "cpu_port_read_plus_crc_GBps": round(size.value / (time.perf_counter() - t0) / 1e9, 2)}

    # a log file the way recovery meets it: the db_bench-shaped log (2040-byte records), file -> per-block verdicts
    n_rec = int(os.environ.get("MF_LOG_RECORDS", "1000000"))
    OL = O.lib()
    t = O.OracleLog(recycle=False, log_number=7, hw=True)
    OL.oracle_wal_write_dbbench_records(t.writer, n_rec, 301)
    log_size = t.written_bytes()
    covered = OL.oracle_wal_writer_crc_bytes(t.writer)
    log_path = os.path.join(tmp, "000007.log")
    np.frombuffer((C.c_uint8 * log_size).from_address(OL.oracle_sink_data(t.sink)), dtype=np.uint8).tofile(log_path)
    t.close()
    runs = []
    report = (C.c_uint64 * 5)()
    for label, extent in (("one piece (read, then verify)", 1 << 40), ("16 MiB pieces", 16 << 20),
                          ("64 MiB pieces", 64 << 20), ("256 MiB pieces", 256 << 20)):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            rc = capi.lib.tcrc_shim_wal_verify_file(eng.ctx, log_path.encode(), 7, extent, 8, None, 0, report, m, 256)
            dt = time.perf_counter() - t0
            assert rc == 0 and report[0] == log_size and report[2] == 0, (rc, m.value, list(report))
            best = dt if best is None else min(best, dt)
        runs.append({"pieces": label, "seconds": round(best, 4), "file_GBps": round(log_size / best / 1e9, 2)})
        print("log_file", runs[-1], flush=True)
    res["log_file_verify"] = {"bytes": log_size, "covered_bytes": covered, "physical_records": int(report[1]),
                              "read_threads": 8, "runs": runs}
    os.remove(log_path)

    # the command-line tool
    tcrc = os.path.join(ROOT, "terarkdb_amd", "bin", "tcrc")
    p = subprocess.run([tcrc, "crc32c", "--block_size=4096"], capture_output=True, text=True, timeout=300)
    res["cli_crc32c_4096"] = p.stdout.splitlines()
    p = subprocess.run([tcrc, "crc32c", "--block_size=8192"], capture_output=True, text=True, timeout=300)
    res["cli_crc32c_8192"] = p.stdout.splitlines()
    print(res["cli_crc32c_4096"], flush=True)
    os.remove(path)
    os.rmdir(tmp)
    eng.close()
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
