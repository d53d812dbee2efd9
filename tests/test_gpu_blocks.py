"""GPU parity: the HIP path through the C ABI against the CPU oracle, bit for bit."""
import json
import os

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CAP = json.load(open(os.path.join(HERE, "golden", "reference_captured.json")))


@pytest.fixture(scope="module", params=[4, 8, 16, 64])
def eng(request):
    import gpu_util as G
    e = G.engine(request.param)
    yield e
    e.close()


def test_splitmix_generator_matches_oracle():
    import gpu_util as G
    d = G.splitmix_dev(1 << 20).cpu().numpy()
    assert bytes(d[:16]).hex() == CAP["splitmix_first16_hex"]
    assert (d == O.splitmix(1 << 20)).all()


def test_uniform_8k_blocks_with_type_byte(eng):
    import gpu_util as G
    import terarkdb_amd.capi as capi
    n, blen = 4099, 8192  # not a multiple of the wave's buffers-per-round
    host = O.splitmix(n * blen, 301)
    dev = G.to_dev(host)
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    eng.blocks_stamp(dev, n, blen, blen, out, suffix_byte=0, flags=capi.F_MASK)
    torch.cuda.synchronize()
    want = O.sst_trailer_batch(host, blen, blen, n, 0)
    got = out.cpu().numpy().view(np.uint32)
    assert (got == want).all()
    # the captured value for the first block: Value(S[:8193]) has type byte 0 == next stream byte? no: check 8192 prefix
    first = O.mask(O.extend(int(dict((a, b) for a, b in CAP["value_of_S_prefix"])[8192], 16), b"\0"))
    assert int(got[0]) == first


@pytest.mark.parametrize("blen,stride", [(4096, 4101), (8192, 8197), (1000, 1000), (16, 16), (129, 131), (5, 9),
                                         (32761, 32768), (65536 + 3, 65536 + 8)])
def test_uniform_other_geometries(eng, blen, stride):
    import gpu_util as G
    import terarkdb_amd.capi as capi
    n = max(3, min(700, (6 << 20) // stride))
    host = O.splitmix(((n * stride + 64) // 8) * 8, 17)
    dev = G.to_dev(host)
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    eng.blocks_stamp(dev, n, blen, stride, out, init=0x1234abcd, suffix_byte=-1, flags=0)
    torch.cuda.synchronize()
    offs = np.arange(n, dtype=np.uint64) * stride
    want = O.batch_extend(host, offs, np.full(n, blen, np.uint32), np.full(n, 0x1234abcd, np.uint32))
    assert (out.cpu().numpy().view(np.uint32) == want).all()


@pytest.mark.parametrize("blen,stride", [(1000, 1003), (1408, 1408), (1409, 1411), (17, 19)])
def test_many_short_uniform_blocks(eng, blen, stride):
    """Enough short blocks that the engine puts four lanes on each (up to 1408 bytes), and just above that."""
    import gpu_util as G
    n = 40000
    host = O.splitmix(((n * stride + 64) // 8) * 8, 23)
    dev = G.to_dev(host)
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    eng.blocks_stamp(dev, n, blen, stride, out, init=0, suffix_byte=3, flags=1)
    torch.cuda.synchronize()
    offs = np.arange(n, dtype=np.uint64) * stride
    want = O.batch_extend(host, offs, np.full(n, blen, np.uint32), np.zeros(n, np.uint32), np.full(n, 3, np.uint8), True)
    assert (out.cpu().numpy().view(np.uint32) == want).all()


def test_descriptor_batch_ragged(eng):
    import gpu_util as G
    import terarkdb_amd.capi as capi
    rng = np.random.default_rng(11)
    total = 8 << 20
    host = rng.integers(0, 256, size=total, dtype=np.uint8)
    lens = np.concatenate([
        np.arange(0, 300),                      # every tiny length, empty included
        rng.integers(0, 5000, size=1500),
        np.array([127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 8192, 8193, 32761, 65535, 65536, 100000]),
    ]).astype(np.uint32)
    rng.shuffle(lens)
    n = lens.size
    offs = rng.integers(0, total - 100001, size=n).astype(np.uint64)
    inits = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    inits[::3] = 0
    suffix = rng.integers(0, 256, size=n, dtype=np.uint8)
    dev = G.to_dev(host)
    desc = capi.make_desc_tensor(offs, lens, inits, "cuda")
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    # plain Extend
    eng.batch_stamp(dev, desc, n, out, flags=0)
    torch.cuda.synchronize()
    want = O.batch_extend(host, offs, lens, inits)
    got = out.cpu().numpy().view(np.uint32)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad[:10], lens[bad[:10]], offs[bad[:10]] % 16)
    # with suffix byte and Mask
    eng.batch_stamp(dev, desc, n, out, suffix=G.to_dev(suffix), flags=capi.F_MASK)
    torch.cuda.synchronize()
    want = O.batch_extend(host, offs, lens, inits, suffix, masked=True)
    assert (out.cpu().numpy().view(np.uint32) == want).all()


def test_reference_known_answers_on_device(eng):
    """util/crc32c_test.cc's table (offset, length, ~crc) over the FNV-chained buffer, one-shot and stitched."""
    import gpu_util as G
    import terarkdb_amd.capi as capi
    ka = json.load(open(os.path.join(HERE, "golden", "crc32c_known_answers.json")))
    host = O.fnv_chain(ka["fnv_chain_buffer_size"])
    dev = G.to_dev(host)
    rows = ka["fnv_chain"]
    n = len(rows)
    offs = np.array([r[0] for r in rows], np.uint64)
    lens = np.array([r[1] for r in rows], np.uint32)
    want = np.array([(~r[2]) & 0xffffffff for r in rows], np.uint32)
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    eng.batch_stamp(dev, capi.make_desc_tensor(offs, lens, np.zeros(n, np.uint32), "cuda"), n, out)
    torch.cuda.synchronize()
    assert (out.cpu().numpy().view(np.uint32) == want).all()
    # stitched: Extend(Value(first half), second half)
    half = lens // 2
    eng.batch_stamp(dev, capi.make_desc_tensor(offs, half, np.zeros(n, np.uint32), "cuda"), n, out)
    torch.cuda.synchronize()
    part = out.cpu().numpy().view(np.uint32).copy()
    eng.batch_stamp(dev, capi.make_desc_tensor(offs + half, lens - half, part, "cuda"), n, out)
    torch.cuda.synchronize()
    assert (out.cpu().numpy().view(np.uint32) == want).all()
    # RFC 3720 vectors
    for v in ka["rfc3720"]:
        b = np.frombuffer(bytes.fromhex(v["hex"]), np.uint8)
        d = G.to_dev(b)
        o1 = torch.zeros(1, dtype=torch.int32, device="cuda")
        eng.blocks_stamp(d, 1, b.size, b.size, o1, flags=0)
        torch.cuda.synchronize()
        assert int(o1.cpu().numpy().view(np.uint32)[0]) == int(v["crc"], 16), v["name"]


def test_verify_counts_and_first_bad(eng):
    import gpu_util as G
    import terarkdb_amd.capi as capi
    n, blen = 2000, 4096
    stride = blen + 5
    img = np.zeros(n * stride, np.uint8)
    data = O.splitmix(n * blen, 99).reshape(n, blen)
    view = img.reshape(n, stride)
    view[:, :blen] = data
    view[:, blen] = 0  # type byte
    tr = O.batch_extend(img, np.arange(n, dtype=np.uint64) * stride, np.full(n, blen + 1, np.uint32), None, None, True)
    view[:, blen + 1:] = tr.view(np.uint8).reshape(n, 4)
    bad = [1500, 37, 999]
    for b in bad:
        view[b, 100] ^= 0x80
    dev = G.to_dev(img)
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    eng.counts_reset(counts)
    # CheckBlockChecksum shape: Value(data, size + 1) against the LE32 after it
    eng.blocks_verify(dev, n, blen + 1, stride, counts, ok=ok, flags=capi.F_EXPECT_AFTER)
    torch.cuda.synchronize()
    c = counts.cpu().numpy().view(np.uint64)
    assert int(c[capi.CNT_MISMATCH]) == 3 and int(c[capi.CNT_FIRST_BAD]) == 37 and int(c[capi.CNT_CHECKED]) == n
    okh = ok.cpu().numpy()
    assert sorted(np.nonzero(okh == 0)[0].tolist()) == sorted(bad)
    # same through an explicit expected array and the suffix-byte form
    exp = G.to_dev(tr)
    eng.counts_reset(counts)
    eng.blocks_verify(dev, n, blen, stride, counts, expected=exp, ok=ok, suffix_byte=0)
    torch.cuda.synchronize()
    c = counts.cpu().numpy().view(np.uint64)
    assert int(c[capi.CNT_MISMATCH]) == 3 and int(c[capi.CNT_FIRST_BAD]) == 37


def test_trailers_stored_in_place(eng):
    """TCRC_F_STORE_AFTER writes [type][Mask(crc)] behind every block, like WriteRawBlock appends it
    (table/block_based_table_builder.cc:626-631), for uniform blocks and for a descriptor list."""
    import gpu_util as G
    import terarkdb_amd.capi as capi
    rng = np.random.default_rng(12)
    n, blen, stride = 777, 4096, 4101
    host = O.splitmix((n * stride + 7) // 8 * 8, 91).copy()
    dev = G.to_dev(host)
    eng.blocks_stamp(dev, n, blen, stride, None, suffix_byte=1, flags=capi.F_MASK | capi.F_STORE_AFTER)
    torch.cuda.synchronize()
    got = dev.cpu().numpy()
    want = O.sst_trailer_batch(host, blen, stride, n, 1)
    for i in (0, 1, 400, n - 1):
        t = got[i * stride + blen: i * stride + blen + 5]
        assert t[0] == 1 and int(t[1:5].view("<u4")[0]) == int(want[i])
    img = got[:n * stride].reshape(n, stride)
    assert (img[:, blen] == 1).all() and (img[:, blen + 1:].copy().view("<u4").reshape(n) == want).all()
    assert (img[:, :blen] == host[:n * stride].reshape(n, stride)[:, :blen]).all()  # contents untouched
    # variable-length blocks with per-block type bytes, packed back to back with their trailers
    lens = rng.integers(1, 9000, 300).astype(np.uint32)
    types = rng.integers(0, 8, 300).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.uint64) + 5)[:-1]]).astype(np.uint64)
    total = int(offs[-1] + lens[-1] + 5)
    host2 = O.splitmix((total + 7) // 8 * 8, 92).copy()
    dev2 = G.to_dev(host2)
    desc = capi.make_desc_tensor(offs, lens, np.zeros(300, np.uint32), "cuda")
    out = torch.zeros(300, dtype=torch.int32, device="cuda")
    eng.batch_stamp(dev2, desc, 300, out, suffix=G.to_dev(types), flags=capi.F_MASK | capi.F_STORE_AFTER)
    torch.cuda.synchronize()
    got2 = dev2.cpu().numpy()
    for i in range(300):
        o, ln = int(offs[i]), int(lens[i])
        crc = O.mask(O.extend(O.value(host2[o:o + ln]), bytes([int(types[i])])))
        assert got2[o + ln] == types[i] and int(got2[o + ln + 1:o + ln + 5].copy().view("<u4")[0]) == crc
        assert int(out.cpu().numpy().view(np.uint32)[i]) == crc
    # and the image now verifies with the in-image expectation
    counts = torch.zeros(capi.CNT_WORDS, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    desc_with_type = capi.make_desc_tensor(offs, lens + 1, np.zeros(300, np.uint32), "cuda")  # fetcher's span
    eng.batch_verify(dev2, desc_with_type, 300, counts, flags=capi.F_EXPECT_AFTER)
    c = counts.cpu().numpy()
    assert (int(c[capi.CNT_MISMATCH]), int(c[capi.CNT_CHECKED])) == (0, 300)
    # internal flag bits are refused at the ABI
    with pytest.raises(capi.TcrcError):
        eng.batch_stamp(dev2, desc, 300, out, flags=0x100)


def test_trailers_stored_in_place_over_many_rounds_per_wave(eng):
    """In-place trailers are held back and stored in bursts, one per lane of a group; with more rounds per wave
    than a group has lanes there are bursts inside the loop and a partial one at its end.  Whole images against
    the oracle: uniform short blocks (four lanes per block, sixteen blocks per wave round) and a packed
    descriptor list with per-block type bytes (eight lanes per span)."""
    import gpu_util as G
    import terarkdb_amd.capi as capi
    # 4096 waves x 16 blocks per round x (4 lanes + 1) rounds, and some
    n, blen, stride = 4096 * 16 * 5 + 777, 96, 101
    host = O.splitmix((n * stride + 7) // 8 * 8, 93).copy()
    dev = G.to_dev(host)
    eng.blocks_stamp(dev, n, blen, stride, None, suffix_byte=2, flags=capi.F_MASK | capi.F_STORE_AFTER)
    torch.cuda.synchronize()
    want_img = host.copy()
    rows = want_img[:n * stride].reshape(n, stride)
    rows[:, blen] = 2
    rows[:, blen + 1:blen + 5] = O.sst_trailer_batch(host, blen, stride, n, 2).astype("<u4").view(np.uint8).reshape(n, 4)
    assert (dev.cpu().numpy() == want_img).all()
    # 4096 waves x 8 spans per round x (8 lanes + 3) rounds, and some
    rng = np.random.default_rng(13)
    m = 4096 * 8 * 11 + 123
    lens = rng.integers(130, 260, m).astype(np.uint32)
    types = rng.integers(0, 8, m).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.uint64) + 5)[:-1]]).astype(np.uint64)
    total = int(offs[-1] + lens[-1] + 5)
    host2 = O.splitmix((total + 7) // 8 * 8, 94).copy()
    dev2 = G.to_dev(host2)
    desc = capi.make_desc_tensor(offs, lens, np.zeros(m, np.uint32), "cuda")
    out = torch.zeros(m, dtype=torch.int32, device="cuda")
    eng.batch_stamp(dev2, desc, m, out, suffix=G.to_dev(types), flags=capi.F_MASK | capi.F_STORE_AFTER)
    torch.cuda.synchronize()
    want = O.batch_extend(host2, offs, lens, None, types, True)
    assert (out.cpu().numpy().view(np.uint32) == want).all()
    want_img2 = host2.copy()
    at = (offs + lens).astype(np.int64)
    want_img2[at] = types
    wb = want.astype("<u4").view(np.uint8).reshape(m, 4)
    for k in range(4):
        want_img2[at + 1 + k] = wb[:, k]
    assert (dev2.cpu().numpy() == want_img2).all()


def test_batch_entry_points_can_be_captured_in_a_graph():
    """The batch entry points only enqueue kernels (no allocation, no synchronisation), so a caller with a
    launch-bound loop can record them into a HIP graph and replay it."""
    import gpu_util as G
    import terarkdb_amd.capi as capi
    eng = G.engine()
    n, blen = 2000, 4096
    host = O.splitmix(n * blen, 5)
    dev = G.to_dev(host)
    stored = torch.zeros(n, dtype=torch.int32, device="cuda")
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    counts = torch.zeros(capi.CNT_WORDS, dtype=torch.int64, device="cuda")
    eng.blocks_stamp(dev, n, blen, blen, stored, suffix_byte=0, flags=capi.F_MASK)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up on the capture stream, as graph capture wants
        eng.counts_reset(counts, stream=side.cuda_stream)
        eng.blocks_verify(dev, n, blen, blen, counts, expected=stored, out=out, suffix_byte=0, stream=side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream().cuda_stream
        eng.counts_reset(counts, stream=s)
        eng.blocks_verify(dev, n, blen, blen, counts, expected=stored, out=out, suffix_byte=0, stream=s)
    want = O.sst_trailer_batch(host, blen, blen, n, 0)
    for rep in range(3):
        out.zero_()
        if rep == 2:
            dev[5 * blen + 1] ^= 4  # the graph reads the same buffers: a change in them shows in the replay
        g.replay()
        torch.cuda.synchronize()
        c = counts.cpu().numpy()
        got = out.cpu().numpy().view(np.uint32)
        if rep < 2:
            assert (got == want).all() and int(c[capi.CNT_MISMATCH]) == 0 and int(c[capi.CNT_CHECKED]) == n
        else:
            assert int(c[capi.CNT_MISMATCH]) == 1 and int(c[capi.CNT_FIRST_BAD]) == 5
    eng.close()


@pytest.mark.parametrize("n", [37, 5000, 20000, 20001])  # the engine picks 64, 16, 8 and (with the hint) 4 lanes per buffer
def test_randomised_descriptor_batches_with_engine_chosen_geometry(n):
    import gpu_util as G
    import terarkdb_amd.capi as capi
    eng = G.engine()
    host = O.splitmix(3 << 20, 1234 + n)
    dev = G.to_dev(host)
    for seed in range(6):
        rng = np.random.default_rng(1000 * n + seed)
        kind = rng.integers(0, 5, n)
        lens = np.where(kind == 0, rng.integers(0, 40, n),                     # tiny, including empty
               np.where(kind == 1, 16 * rng.integers(0, 64, n),               # multiples of a column
               np.where(kind == 2, 128 * rng.integers(1, 40, n),              # multiples of a row
               np.where(kind == 3, rng.integers(1, 70000 if n < 100 else 6000, n),
                        rng.integers(1000, 1100, n))))).astype(np.uint32)
        offs = rng.integers(0, host.size - 70001, n).astype(np.uint64)
        if seed % 2:
            offs &= ~np.uint64(127)                                            # line-aligned starts
        inits = rng.integers(0, 2**32, n).astype(np.uint32)
        sfx = rng.integers(0, 256, n).astype(np.uint8) if seed % 3 == 0 else None
        masked = bool(seed & 1)
        desc = capi.make_desc_tensor(offs, lens, inits, "cuda")
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
        hint = capi.F_SHORT_SPANS if n == 20001 else 0  # results must not depend on the hint
        eng.batch_stamp(dev, desc, n, out, suffix=None if sfx is None else G.to_dev(sfx),
                        flags=(capi.F_MASK if masked else 0) | hint)
        torch.cuda.synchronize()
        want = O.batch_extend(host, offs, lens, inits, sfx, masked)
        got = out.cpu().numpy().view(np.uint32)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (n, seed, bad[:5], lens[bad[:5]], offs[bad[:5]] % 1024)
    eng.close()
