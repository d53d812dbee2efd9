"""Several GPUs in one process (tcrc_group_*) and the process-per-GPU collective (tcrc_rccl_allreduce_counts),
through the C ABI.  The box has one GPU, so a group is made of one member, or of two members on GPU 0: the
sharding, the per-member threads and the reduction are the real ones.  Shards are checked against the CPU oracle:
totals and the first bad block's index must be exact for seeded corruptions."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu

BLEN = 4096


def _group(devices):
    import terarkdb_amd.capi as capi
    arr = (C.c_int * len(devices))(*devices)
    g = C.c_void_p()
    capi.check(capi.lib.tcrc_group_init(arr, len(devices), C.byref(g)))
    return g


def _corpus(n, seed, bad):
    """n blocks of BLEN bytes, each followed by [type 0][masked crc] (stride BLEN + 5); bad: blocks with a flipped bit"""
    raw = O.splitmix((n * BLEN + 7) // 8 * 8, seed)[:n * BLEN]
    tr = O.sst_trailer_batch(raw, BLEN, BLEN, n, 0)
    img = np.zeros(n * (BLEN + 5), dtype=np.uint8)
    v = img.reshape(n, BLEN + 5)
    v[:, :BLEN] = raw.reshape(n, BLEN)
    v[:, BLEN] = 0
    v[:, BLEN + 1:] = tr.astype("<u4").view(np.uint8).reshape(n, 4)
    for b in bad:
        v[b, (b * 7) % BLEN] ^= 0x10
    return img, tr


def test_shard_range_is_the_python_one():
    import terarkdb_amd.capi as capi
    from terarkdb_amd.sharding import shard_range
    for n in (0, 1, 7, 1003, 8388608):
        for count in (1, 2, 3, 8):
            for i in range(count):
                lo, hi = C.c_uint64(0), C.c_uint64(0)
                capi.lib.tcrc_shard_range(n, i, count, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == shard_range(n, i, count)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_group_verifies_host_and_device_shards(devices):
    import terarkdb_amd.capi as capi
    L = capi.lib
    g = _group(devices)
    members = L.tcrc_group_size(g)
    assert members == len(devices) and all(L.tcrc_group_device(g, m) == 0 for m in range(members))
    n = 1003
    bad = [17, 500, 501, 1002]
    img, tr = _corpus(n, 55, bad)
    counts = (C.c_uint64 * 3)()
    ok = np.full(n, 9, dtype=np.uint8)
    # one image in host memory, stored values in the image behind every block and its type byte
    capi.check(L.tcrc_group_host_blocks_verify(g, img.ctypes.data, n, BLEN, BLEN + 5, 0, 0, capi.F_EXPECT_AFTER, None,
                                               ok.ctypes.data, counts))
    assert list(counts) == [len(bad), bad[0], n]
    want_ok = np.ones(n, dtype=np.uint8)
    want_ok[bad] = 0
    assert (ok == want_ok).all()
    # a clean image
    clean, _ = _corpus(n, 55, [])
    capi.check(L.tcrc_group_host_blocks_verify(g, clean.ctypes.data, n, BLEN, BLEN + 5, 0, 0, capi.F_EXPECT_AFTER, None,
                                               None, counts))
    assert list(counts) == [0, 2**64 - 1, n]
    # shards resident on the members' GPUs, stored values in arrays of their own
    from terarkdb_amd.sharding import shard_range
    v = img.reshape(n, BLEN + 5)
    shards, exps = [], []
    for m in range(members):
        lo, hi = shard_range(n, m, members)
        shards.append(torch.from_numpy(np.ascontiguousarray(v[lo:hi, :BLEN])).cuda())
        exps.append(torch.from_numpy(tr[lo:hi].astype(np.uint32).view(np.int32).copy()).cuda())
    bases = (C.c_void_p * members)(*[s.data_ptr() for s in shards])
    exp = (C.c_void_p * members)(*[e.data_ptr() for e in exps])
    capi.check(L.tcrc_group_dev_blocks_verify(g, bases, n, BLEN, BLEN, 0, 0, 0, exp, counts))
    assert list(counts) == [len(bad), bad[0], n]
    # only the last shard is damaged: the first bad index is still the corpus's
    img2, _ = _corpus(n, 55, [1001])
    capi.check(L.tcrc_group_host_blocks_verify(g, img2.ctypes.data, n, BLEN, BLEN + 5, 0, 0, capi.F_EXPECT_AFTER, None,
                                               None, counts))
    assert list(counts) == [1, 1001, n]
    # fewer blocks than members, and none at all
    capi.check(L.tcrc_group_host_blocks_verify(g, img.ctypes.data, 1, BLEN, BLEN + 5, 0, 0, capi.F_EXPECT_AFTER, None,
                                               None, counts))
    assert list(counts) == [0, 2**64 - 1, 1]
    capi.check(L.tcrc_group_host_blocks_verify(g, img.ctypes.data, 0, BLEN, BLEN + 5, 0, 0, capi.F_EXPECT_AFTER, None,
                                               None, counts))
    assert list(counts) == [0, 2**64 - 1, 0]
    assert L.tcrc_group_shutdown(g) == 0


def test_host_blocks_verify_single_context():
    """tcrc_host_blocks_verify, the one-context entry point the group's host form stands on."""
    import gpu_util as G
    import terarkdb_amd.capi as capi
    L = capi.lib
    eng = G.engine()
    n = 300
    img, tr = _corpus(n, 9, [3, 299])
    counts = (C.c_uint64 * 3)()
    out = np.zeros(n, dtype=np.uint32)
    ok = np.zeros(n, dtype=np.uint8)
    capi.check(L.tcrc_host_blocks_verify(eng.ctx, img.ctypes.data, n, BLEN, BLEN + 5, 0, 0, capi.F_EXPECT_AFTER, None,
                                         out.ctypes.data, ok.ctypes.data, counts))
    assert list(counts) == [2, 3, n] and ok.sum() == n - 2
    good = np.ones(n, dtype=bool)
    good[[3, 299]] = False
    assert (out[good] == tr[good]).all() and (out[~good] != tr[~good]).all()
    # stored values handed in instead
    capi.check(L.tcrc_host_blocks_verify(eng.ctx, img.ctypes.data, n, BLEN, BLEN + 5, 0, 0, 0, tr.ctypes.data, None, None, counts))
    assert list(counts) == [2, 3, n]
    assert L.tcrc_host_blocks_verify(eng.ctx, img.ctypes.data, n, BLEN, BLEN + 5, 0, 0, 0, None, None, None, counts) == capi.ERR_INVALID
    assert L.tcrc_host_blocks_verify(eng.ctx, img.ctypes.data, n, BLEN, BLEN - 1, 0, 0, capi.F_EXPECT_AFTER, None, None, None,
                                     counts) == capi.ERR_INVALID
    eng.close()


def test_rccl_allreduce_of_the_counter_block_one_rank():
    """The collective of the process-per-GPU form on a communicator of one rank: librccl is found and called,
    the counter block comes back in its own layout with first_bad in the corpus's numbering."""
    import gpu_util as G
    import terarkdb_amd.capi as capi
    L = capi.lib
    eng = G.engine()
    uid = (C.c_uint8 * 128)()
    capi.check(L.tcrc_rccl_get_unique_id(uid))
    comm = C.c_void_p()
    capi.check(L.tcrc_rccl_comm_init_rank(eng.ctx, 1, uid, 0, C.byref(comm)))
    stream = torch.cuda.current_stream().cuda_stream
    for block, first_unit, want in (([4, 17, 1003], 5000, [4, 5017, 1003]), ([0, -1, 77], 5000, [0, -1, 77])):
        counts = torch.tensor(block, dtype=torch.int64, device="cuda")
        capi.check(L.tcrc_rccl_allreduce_counts(eng.ctx, comm, C.c_void_p(counts.data_ptr()), first_unit, C.c_void_p(stream)))
        torch.cuda.synchronize()
        assert counts.cpu().tolist() == want
    capi.check(L.tcrc_rccl_comm_destroy(comm))
    eng.close()


def test_verify_db_directory_over_a_group(tmp_path):
    import test_gpu_cli as T
    import sst_files as S
    import terarkdb_amd.capi as capi
    L = capi.lib
    d, lays, imgs = T._make_db_dir(tmp_path)
    eng_rep = (C.c_uint64 * 8)()
    msg, bad = C.create_string_buffer(1024), C.create_string_buffer(1024)
    g1, g2 = _group([0]), _group([0, 0])
    assert L.tcrc_shim_verify_db_group(g1, str(d).encode(), eng_rep, msg, 1024, bad, 1024) == 0
    one = [int(x) for x in eng_rep]
    assert L.tcrc_shim_verify_db_group(g2, str(d).encode(), eng_rep, msg, 1024, bad, 1024) == 0
    assert [int(x) for x in eng_rep] == one and msg.value == b"OK"
    # two damaged tables on different members: the status is that of the table with the smaller number
    off, _ = lays[3].handle("data2")
    S.flip(lays[3].path, off + 3, 1)
    off, _ = lays[1].handle("data1")
    S.flip(lays[1].path, off + 5, 1)
    rc = L.tcrc_shim_verify_db_group(g2, str(d).encode(), eng_rep, msg, 1024, bad, 1024)
    assert rc == 1 and bad.value.decode() == lays[1].path and msg.value.decode() == S.expected_status(lays[1])
    L.tcrc_group_shutdown(g1)
    L.tcrc_group_shutdown(g2)
