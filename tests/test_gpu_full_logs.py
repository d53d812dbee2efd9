"""Log verification at the full sizes of the benchmark configurations, per-block results against the CPU oracle.

* the CPU-reference configuration: 1 000 000 db_bench-shaped write batches (24-byte key, 2000-byte value) written by
  the oracle's log writer - 1 062 207 physical records, 2 047 436 274 bytes, the facts captured from a build of
  the reference (db/log_writer.cc:45-151) - verified on the device as one image;
* the mixed-size configuration: at least 256 MiB of records with the size distribution of db/log_test.cc:353-364
  (Random(301).Skewed(17)) behind the edge-case records of db/log_test.cc:268-351, legacy and recyclable headers,
  clean and with 1000 seeded single-bit flips wherever they fall (payload, type, checksum, length, padding).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CAP = json.load(open(os.path.join(HERE, "golden", "reference_captured.json")))


@pytest.fixture(scope="module")
def eng():
    import gpu_util as G
    e = G.engine()
    yield e
    e.close()


def device_scan(eng, img, log_number):
    import terarkdb_amd.capi as capi
    nb = (img.size + 32767) // 32768
    dev = torch.from_numpy(img).cuda()
    blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(capi.CNT_WORDS, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    n = eng.wal_verify(dev, img.size, log_number, blocks, counts)
    torch.cuda.synchronize()
    return n, blocks.cpu().numpy().view(np.uint32), counts.cpu().numpy().view(np.uint64)


def oracle_scan(img, log_number):
    want = np.zeros(((img.size + 32767) // 32768, 4), dtype=np.uint32)
    O.lib().oracle_wal_scan_blocks(img.ctypes.data, img.size, log_number, want.ctypes.data)
    return want


def test_one_million_db_bench_records(eng):
    import terarkdb_amd.capi as capi
    OL = O.lib()
    t = O.OracleLog(recycle=False, log_number=7, hw=True)
    OL.oracle_wal_write_dbbench_records(t.writer, 1000000, 301)
    size = t.written_bytes()
    facts = (int(OL.oracle_wal_writer_physical_records(t.writer)), int(size), int(OL.oracle_wal_writer_crc_bytes(t.writer)))
    assert facts == (1062207, 2047436274, 2041062207)
    img = np.frombuffer((C.c_uint8 * size).from_address(OL.oracle_sink_data(t.sink)), dtype=np.uint8).copy()
    t.close()
    want = oracle_scan(img, 7)
    assert int(want[:, 0].sum()) == 1062207 and (want[:, 1] == 0xffffffff).all()
    n, got, c = device_scan(eng, img, 7)
    assert n == 1062207
    assert (got == want).all()
    assert (int(c[capi.CNT_MISMATCH]), int(c[capi.CNT_CHECKED])) == (0, 1062207)
    # damage: one bit in each of 300 records' payloads, 20 in type bytes, 20 in stored checksums, 20 in lengths
    rng = np.random.default_rng(1)
    nb = want.shape[0]
    for k, blk in enumerate(rng.choice(nb - 1, 360, replace=False)):
        base = int(blk) * 32768
        # the first record of a block starts at the block's start except behind a spilled fragment: find a header
        # by walking the oracle's result for the block is not needed - records are 2047 bytes apart on average,
        # so a flip at a random offset lands in some record's bytes
        pos = base + int(rng.integers(0, 32768 - 16))
        img[pos] ^= 1 << int(rng.integers(0, 8))
    want = oracle_scan(img, 7)
    n, got, c = device_scan(eng, img, 7)
    assert (got == want).all()
    assert int((want[:, 1] != 0xffffffff).sum()) >= 300
    first = int(np.nonzero(want[:, 1] != 0xffffffff)[0][0])
    assert int(c[capi.CNT_FIRST_BAD]) == first and int(c[capi.CNT_MISMATCH]) >= int((want[:, 1] != 0xffffffff).sum())


@pytest.mark.parametrize("recycle", [False, True])
def test_mixed_size_log_256_mib(eng, recycle):
    import terarkdb_amd.capi as capi
    OL = O.lib()
    tl = O.OracleLog(recycle=recycle, log_number=123)
    O.write_edge_records(tl, recycle)
    rnd = O.Random(301)
    while tl.written_bytes() < 256 * 1024 * 1024:
        tl.write(O.big_string(b"%d." % tl.written_bytes(), rnd.skewed(17)))
    size = tl.written_bytes()
    img = np.frombuffer((C.c_uint8 * size).from_address(OL.oracle_sink_data(tl.sink)), dtype=np.uint8).copy()
    tl.close()
    want = oracle_scan(img, 123)
    assert (want[:, 1] == 0xffffffff).all()
    n, got, c = device_scan(eng, img, 123)
    assert (got == want).all() and n == int(want[:, 0].sum())
    assert (int(c[capi.CNT_MISMATCH]), int(c[capi.CNT_CHECKED])) == (0, n)
    rng = np.random.default_rng(int(recycle))
    for _ in range(1000):
        img[int(rng.integers(0, size))] ^= 1 << int(rng.integers(0, 8))
    want = oracle_scan(img, 123)
    n, got, c = device_scan(eng, img, 123)
    assert (got == want).all()
    bad_blocks = np.nonzero(want[:, 1] != 0xffffffff)[0]
    assert bad_blocks.size > 500
    assert int(c[capi.CNT_FIRST_BAD]) == int(bad_blocks[0])
    # a wrong log number: recyclable records all turn into "old record" ends, legacy ones do not care
    n2, got2, _ = device_scan(eng, img[:8 << 20].copy(), 124)
    assert (got2 == oracle_scan(img[:8 << 20].copy(), 124)).all()
