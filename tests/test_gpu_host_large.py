"""Host-pointer batches large enough for the threaded staging ring (uploads of 16 MiB and more from pageable
memory): a call that uploads its image AND its descriptors through the ring, back to back, against the CPU oracle;
and a stamp / verify round trip of trailers written behind a suffix byte."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib as O
import terarkdb_amd.capi as capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import gpu_util as G
    e = G.engine()
    yield e
    e.close()


def test_image_and_descriptors_both_through_the_staging_ring(eng):
    # 64 MiB pageable image, 1 Mi descriptors (16 MiB, pageable): both uploads take the ring of pinned slots,
    # the second one right behind the first
    n, blen = 1 << 20, 64
    img = O.splitmix(n * blen, 41)
    rng = np.random.default_rng(7)
    desc = np.zeros(n, dtype=[("offset", "<u8"), ("len", "<u4"), ("init", "<u4")])
    desc["offset"] = np.arange(n, dtype=np.uint64) * blen + rng.integers(0, 8, n).astype(np.uint64)
    desc["len"] = rng.integers(0, blen - 8, n).astype(np.uint32)
    desc["init"] = rng.integers(0, 2**32, n).astype(np.uint32)
    want = O.batch_extend(img, desc["offset"], desc["len"], desc["init"], None, True)
    out = np.zeros(n, dtype=np.uint32)
    for _ in range(3):  # the ring's slots carry state from call to call
        out[:] = 0
        capi.check(capi.lib.tcrc_host_batch_stamp(eng.ctx, img.ctypes.data, img.size, desc.ctypes.data, n, None,
                                                  capi.F_MASK, out.ctypes.data))
        bad = np.nonzero(out != want)[0]
        assert bad.size == 0, (bad[:5], bad.size)
    # verify against expectations with 50 of them spoiled
    expected = want.copy()
    spoiled = np.sort(rng.choice(n, 50, replace=False))
    expected[spoiled] ^= 1
    ok = np.zeros(n, dtype=np.uint8)
    counts = np.zeros(3, dtype=np.uint64)
    capi.check(capi.lib.tcrc_host_batch_verify(eng.ctx, img.ctypes.data, img.size, desc.ctypes.data, n, None, 0,
                                               expected.ctypes.data, out.ctypes.data, ok.ctypes.data, counts.ctypes.data))
    assert (out == want).all()
    assert (int(counts[0]), int(counts[1]), int(counts[2])) == (50, int(spoiled[0]), n)
    assert np.nonzero(ok == 0)[0].tolist() == spoiled.tolist()


def test_trailers_stored_behind_a_suffix_byte_verify_with_the_same_arguments(eng):
    # [block][type][Mask(crc32c(block || type))] written by TCRC_F_STORE_AFTER with a suffix byte
    # (table/block_based_table_builder.cc:598-631) must verify with (block_len, suffix byte, TCRC_F_EXPECT_AFTER)
    import gpu_util as G
    n, blen, stride, typ = 3000, 4000, 4008, 1
    host = O.splitmix(n * stride, 8)
    dev = G.to_dev(host)
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    eng.blocks_stamp(dev, n, blen, stride, out, suffix_byte=typ, flags=capi.F_MASK | capi.F_STORE_AFTER)
    torch.cuda.synchronize()
    img = dev.cpu().numpy()
    want = O.sst_trailer_batch(host, blen, stride, n, typ)
    tr = np.frombuffer(img.tobytes(), dtype=np.uint8).reshape(n, stride)[:, blen:blen + 5]
    assert (tr[:, 0] == typ).all()
    assert (np.ascontiguousarray(tr[:, 1:5]).view("<u4").ravel() == want).all()
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    eng.counts_reset(counts)
    eng.blocks_verify(dev, n, blen, stride, counts, ok=ok, suffix_byte=typ, flags=capi.F_EXPECT_AFTER)
    torch.cuda.synchronize()
    assert counts.tolist()[0] == 0 and counts.tolist()[2] == n and bool(ok.all())
    # per-block suffix bytes and descriptors: the same rule
    types = (np.arange(n) % 3).astype(np.uint8)
    dtypes = G.to_dev(types)
    desc = capi.make_desc_tensor(np.arange(n, dtype=np.uint64) * stride, np.full(n, blen, np.uint32), np.zeros(n, np.uint32), "cuda")
    eng.batch_stamp(dev, desc, n, out, suffix=dtypes, flags=capi.F_MASK | capi.F_STORE_AFTER)
    eng.counts_reset(counts)
    eng.batch_verify(dev, desc, n, counts, ok=ok, suffix=dtypes, flags=capi.F_EXPECT_AFTER)
    torch.cuda.synchronize()
    assert counts.tolist()[0] == 0 and bool(ok.all())
    dev[5 * stride + 17] ^= 4
    eng.counts_reset(counts)
    eng.batch_verify(dev, desc, n, counts, ok=ok, suffix=dtypes, flags=capi.F_EXPECT_AFTER)
    torch.cuda.synchronize()
    assert counts.tolist()[:2] == [1, 5]
