"""The log scan under record-size distributions chosen to press on its bookkeeping: per-block results against
the CPU oracle (db/log_reader.cc:317-406 restated), clean and with seeded damage.

The scan cuts checksummed spans into pieces of at most 2 KiB that pass through a ring of rounds in LDS; spans of
several pieces meet in a small number of slots.  The distributions below keep one of those resources busy at a
time: nothing but whole-block fragments (sixteen pieces each, every slot taken), nothing but tiny records (rounds
of entries that carry almost no bytes, hundreds of records per block), records just around the piece size and its
multiples, and abrupt changes between the extremes inside one block.
"""
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


def sizes_for(kind, rng, total):
    """record sizes of one distribution, about `total` bytes in all"""
    out, acc = [], 0
    while acc < total:
        if kind == "whole_block_fragments":
            n = int(rng.integers(200000, 400000))  # every record spills over six to twelve blocks
        elif kind == "tiny":
            n = int(rng.integers(0, 40))
        elif kind == "around_the_piece_size":
            n = int(rng.choice([2040, 2041, 2042, 2047, 2048, 2049, 4089, 4090, 4096, 4097, 6137, 6138, 6139]))
        elif kind == "extremes":
            n = int(rng.choice([0, 1, 5, 30000, 32761, 32768, 70000])) if rng.random() < 0.5 else int(rng.integers(0, 64))
        elif kind == "bursts":
            # a run of tiny records, then a run of block-sized ones
            run = int(rng.integers(50, 400))
            big = rng.random() < 0.3
            for _ in range(run):
                n = int(rng.integers(20000, 40000)) if big else int(rng.integers(0, 100))
                out.append(n)
                acc += n + 7
            continue
        else:
            raise ValueError(kind)
        out.append(n)
        acc += n + 7
    return out


def build(kind, recycle, total, seed):
    rng = np.random.default_rng(seed)
    t = O.OracleLog(recycle=recycle, log_number=77)
    filler = O.splitmix(1 << 19, seed).tobytes()
    for n in sizes_for(kind, rng, total):
        o = int(rng.integers(0, len(filler) - n)) if n < len(filler) else 0
        t.write(filler[o:o + n])
    img = t.contents()
    t.close()
    return np.frombuffer(img, dtype=np.uint8).copy()


def scan_both(eng, img):
    import terarkdb_amd.capi as capi
    nb = (img.size + 32767) // 32768
    want = np.zeros((nb, 4), dtype=np.uint32)
    O.lib().oracle_wal_scan_blocks(img.ctypes.data, img.size, 77, want.ctypes.data)
    dev = torch.from_numpy(img).cuda()
    blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(capi.CNT_WORDS, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    n = eng.wal_verify(dev, img.size, 77, blocks, counts)
    torch.cuda.synchronize()
    return n, blocks.cpu().numpy().view(np.uint32), counts.cpu().numpy().view(np.uint64), want


KINDS = ["whole_block_fragments", "tiny", "around_the_piece_size", "extremes", "bursts"]


@pytest.mark.parametrize("recycle", [False, True])
@pytest.mark.parametrize("kind", KINDS)
def test_distribution(eng, kind, recycle):
    import terarkdb_amd.capi as capi
    total = (6 << 20) if kind == "tiny" else (48 << 20)
    img = build(kind, recycle, total, seed=KINDS.index(kind) * 2 + int(recycle))
    n, got, c, want = scan_both(eng, img)
    assert (want[:, 1] == 0xffffffff).all(), "the oracle must find the clean log clean"
    assert n == int(want[:, 0].sum())
    assert (got == want).all()
    assert int(c[capi.CNT_MISMATCH]) == 0 and int(c[capi.CNT_CHECKED]) == n
    # damage: single-bit flips wherever they fall, a few hundred of them
    rng = np.random.default_rng(99)
    for pos in rng.integers(0, img.size, 400):
        img[int(pos)] ^= 1 << int(rng.integers(0, 8))
    n, got, c, want = scan_both(eng, img)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, (kind, recycle, bad[:5], got[bad[:5]], want[bad[:5]])
    assert n == int(want[:, 0].sum())
    assert int(c[capi.CNT_FIRST_BAD]) == int(np.nonzero(want[:, 1] != 0xffffffff)[0][0])


def test_many_scans_in_a_row_share_nothing(eng):
    """Scans queued back to back on one stream, with and without the record count: every one must leave the
    same per-block results (the ring, the slots and the launch words start clean every time)."""
    import terarkdb_amd.capi as capi
    img = build("extremes", False, 8 << 20, seed=5)
    nb = (img.size + 32767) // 32768
    want = np.zeros((nb, 4), dtype=np.uint32)
    O.lib().oracle_wal_scan_blocks(img.ctypes.data, img.size, 77, want.ctypes.data)
    dev = torch.from_numpy(img).cuda()
    counts = torch.zeros(capi.CNT_WORDS, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    outs = [torch.zeros((nb, 4), dtype=torch.int32, device="cuda") for _ in range(40)]
    total = 0
    for i, blocks in enumerate(outs):
        n = eng.wal_verify(dev, img.size, 77, blocks, counts, want_records=(i % 3 == 0))
        if i % 3 == 0:
            assert n == int(want[:, 0].sum())
        total += int(want[:, 0].sum())
    torch.cuda.synchronize()
    for blocks in outs:
        assert (blocks.cpu().numpy().view(np.uint32) == want).all()
    assert int(counts.cpu().numpy().view(np.uint64)[capi.CNT_CHECKED]) == total
