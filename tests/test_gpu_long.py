"""Long buffers (256 KiB - 4 MiB and beyond): intra-buffer split + GF(2) combine against the CPU oracle."""
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


@pytest.fixture(scope="module")
def S():
    return O.splitmix(4 * 1024 * 1024 + 64, 301)


def run(eng, dev, offs, lens, inits, seg=0):
    out = torch.zeros(len(offs), dtype=torch.int32, device="cuda")
    eng.extend_long(dev, offs, lens, inits, out, segment_len=seg)
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint32)


def test_captured_reference_values(eng, S):
    import gpu_util as G
    dev = G.to_dev(S)
    rows = [(0, n, 0, int(h, 16)) for n, h in CAP["value_of_S_prefix"]]
    rows += [(e["offset"], e["length"], int(e["init"], 16), int(e["crc"], 16)) for e in CAP["extend_of_S"]]
    for seg in (0, 256, 2048, 65536):
        got = run(eng, dev, [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows], seg)
        assert got.tolist() == [r[3] for r in rows], seg


def test_sizes_alignments_inits_against_oracle(eng, S):
    import gpu_util as G
    dev = G.to_dev(S)
    rng = np.random.default_rng(23)
    offs, lens, inits = [], [], []
    for base_len in (256 * 1024, 1024 * 1024, 4 * 1024 * 1024 - 64):
        for delta in (-7, -1, 0, 1, 7):
            for a in (0, 1, 3, 8, 15):
                offs.append(a)
                lens.append(base_len + delta)
                inits.append(int(rng.integers(0, 2**32)))
    offs += [5, 0, 9, 2]
    lens += [0, 1, 3, 4]  # degenerate lengths still go through
    inits += [0x12345678, 0, 0xffffffff, 7]
    for seg in (0, 4096):
        got = run(eng, dev, offs, lens, inits, seg)
        want = O.batch_extend(S, np.array(offs, np.uint64), np.array(lens, np.uint32), np.array(inits, np.uint32))
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (seg, bad[:5], [lens[i] for i in bad[:5]])


def test_batch_of_64_and_chaining(eng):
    import gpu_util as G
    host = O.splitmix(64 * 1024 * 1024, 77)
    dev = G.to_dev(host)
    n = 64
    offs = np.arange(n, dtype=np.uint64) * (1 << 20)
    lens = np.full(n, 1 << 20, np.uint32)
    got = run(eng, dev, offs, lens, np.zeros(n, np.uint32))
    want = O.batch_extend(host, offs, lens, np.zeros(n, np.uint32))
    assert (got == want).all()
    # streaming a 64 MiB file in 5 MiB pieces, each call seeded with the previous result (backup shape)
    crc = 0
    pos = 0
    while pos < host.size:
        k = min(5 << 20, host.size - pos)
        crc = int(run(eng, dev, [pos], [k], [crc])[0])
        pos += k
    assert crc == O.value(host, "hw")
    # the same file in one call
    assert int(run(eng, dev, [0], [host.size], [0])[0]) == crc


def test_enqueued_calls_in_a_row(eng):
    """tcrc_dev_extend_long_enqueue: a row of calls on one stream, different buffers and outputs, one wait at the
    end; every call's values equal the waiting entry point's and the oracle's.  Chaining through init across the
    row is what a whole-file checksum does buffer after buffer."""
    import gpu_util as G
    import terarkdb_amd.capi as capi
    host = O.splitmix(24 << 20, 4711)
    dev = G.to_dev(host)
    rng = np.random.default_rng(3)
    calls = []
    for k in range(12):
        n = int(rng.integers(1, 65))
        lens = [int(rng.integers(0, 300000)) for _ in range(n)]
        offs = [int(rng.integers(0, host.size - ln)) for ln in lens]
        inits = [int(rng.integers(0, 2**32)) for _ in range(n)]
        out = torch.full((n,), 0x55555555, dtype=torch.int32, device="cuda")
        calls.append((offs, lens, inits, out))
    for offs, lens, inits, out in calls:
        eng.extend_long(dev, offs, lens, inits, out, enqueue=True)  # only enqueues
    torch.cuda.synchronize()
    for offs, lens, inits, out in calls:
        want = O.batch_extend(host, np.array(offs, np.uint64), np.array(lens, np.uint32), np.array(inits, np.uint32))
        assert (out.cpu().numpy().view(np.uint32) == want).all()
    # a waiting call right behind enqueued ones sees the lines clean
    offs, lens, inits, out = calls[0]
    ref = torch.zeros_like(out)
    eng.extend_long(dev, offs, lens, inits, ref)
    assert torch.equal(ref, out)
    with pytest.raises(capi.TcrcError, match="at most 64 buffers"):
        eng.extend_long(dev, [0] * 65, [10] * 65, [0] * 65, torch.zeros(65, dtype=torch.int32, device="cuda"), enqueue=True)
