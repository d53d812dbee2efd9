"""Argument checking of the C ABI on a GPU box: bad arguments come back as error codes with a text,
empty batches are no-ops, and nothing is launched for either."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_bad_arguments_and_empty_batches():
    import gpu_util as G
    import terarkdb_amd.capi as capi
    L = capi.lib
    eng = G.engine()
    dev = G.to_dev(O.splitmix(1 << 16, 1))
    out = torch.zeros(16, dtype=torch.int32, device="cuda")
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    s = eng._stream(None)
    p = lambda t: C.c_void_p(t.data_ptr())

    def err(rc, code=capi.ERR_INVALID):
        assert rc == code, (rc, L.tcrc_last_error())
        assert L.tcrc_last_error()

    # empty batches succeed without touching anything
    assert L.tcrc_dev_blocks_stamp(eng.ctx, p(dev), 0, 4096, 4096, 0, -1, None, 1, p(out), s) == 0
    assert L.tcrc_dev_batch_stamp(eng.ctx, p(dev), None, 0, None, 1, p(out), s) == 0
    assert L.tcrc_dev_extend_long(eng.ctx, p(dev), None, 0, 0, p(out), s) == 0
    n = C.c_uint64(7)
    assert L.tcrc_dev_wal_verify(eng.ctx, p(dev), 0, 1, None, None, C.byref(n), s) == 0 and n.value == 0
    v = C.c_uint32(5)
    assert L.tcrc_extend(eng.ctx, 0x1234, None, 0, C.byref(v)) == 0 and v.value == 0x1234
    # geometry and pointers
    err(L.tcrc_dev_blocks_stamp(eng.ctx, p(dev), 4, 4096, 100, 0, -1, None, 1, p(out), s))      # stride < block_len
    err(L.tcrc_dev_blocks_stamp(eng.ctx, p(dev), 4, 4096, 4096, 0, -1, None, 1, None, s))       # no output
    err(L.tcrc_dev_blocks_stamp(eng.ctx, None, 4, 4096, 4096, 0, -1, None, 1, p(out), s))       # no base
    err(L.tcrc_dev_blocks_stamp(eng.ctx, p(dev), 1, (1 << 30) + 1, 1 << 31, 0, -1, None, 1, p(out), s), capi.ERR_TOO_LARGE)
    err(L.tcrc_dev_blocks_verify(eng.ctx, p(dev), 4, 4096, 4096, 0, -1, None, 0, None, None, None, p(counts), s))  # nothing to compare with
    err(L.tcrc_dev_blocks_verify(eng.ctx, p(dev), 4, 4096, 4101, 0, -1, None, 2, None, None, None, None, s))       # no counts
    # flag bits: internal ones, stamp-only and verify-only ones in the wrong place
    err(L.tcrc_dev_blocks_stamp(eng.ctx, p(dev), 4, 4096, 4096, 0, -1, None, 0x100, p(out), s))
    err(L.tcrc_dev_blocks_stamp(eng.ctx, p(dev), 4, 4096, 4096, 0, -1, None, capi.F_EXPECT_AFTER, p(out), s))
    err(L.tcrc_dev_blocks_verify(eng.ctx, p(dev), 4, 4096, 4101, 0, -1, None, capi.F_STORE_AFTER, None, None, None, p(counts), s))
    # long buffers
    d = (capi.Desc * 1)()
    d[0].offset, d[0].len = 0, 4096
    err(L.tcrc_dev_extend_long(eng.ctx, p(dev), d, 1, 100, p(out), s))                          # segment not a power of two
    err(L.tcrc_dev_extend_long(eng.ctx, p(dev), d, 1, 128, p(out), s))                          # segment too small
    # log images
    err(L.tcrc_dev_wal_verify(eng.ctx, p(dev), 1 << 16, 1, None, p(counts), None, s))           # no per-block results
    err(L.tcrc_dev_wal_frame(eng.ctx, p(dev), d, 1, 1, 0, 40000, p(dev), 1 << 16, None, None, None, s))  # block offset
    err(L.tcrc_dev_wal_frame(eng.ctx, p(dev), d, 1, 1, 0, 0, p(dev), 100, None, None, None, s))          # capacity
    # host forms
    hb = np.zeros(4096, dtype=np.uint8)
    d[0].offset, d[0].len = 4000, 200
    ho = (C.c_uint32 * 1)()
    err(L.tcrc_host_batch_stamp(eng.ctx, hb.ctypes.data, 4096, d, 1, None, 1, ho))             # span outside the buffer
    d[0].offset, d[0].len = 2, 10
    hc = (C.c_uint64 * 3)()
    err(L.tcrc_host_batch_verify(eng.ctx, hb.ctypes.data, 4096, d, 1, None, capi.F_EXPECT_BEFORE, None, None, None, hc))
    err(L.tcrc_host_batch_stamp(eng.ctx, hb.ctypes.data, 4096, d, 1, None, capi.F_STORE_AFTER, ho))
    # a null context
    err(L.tcrc_dev_blocks_stamp(None, p(dev), 4, 4096, 4096, 0, -1, None, 1, p(out), s))
    assert L.tcrc_debug_tables(5, None, None, 0, None) == capi.ERR_INVALID
    # the engine still works after all of that
    eng.blocks_stamp(dev, 16, 4096, 4096, out, flags=capi.F_MASK)
    torch.cuda.synchronize()
    want = O.batch_extend(dev.cpu().numpy(), np.arange(16, dtype=np.uint64) * 4096, np.full(16, 4096, np.uint32), None,
                          None, True)
    assert (out.cpu().numpy().view(np.uint32) == want).all()
    eng.close()


def test_running_totals():
    """tcrc_get_stats counts what went through a context."""
    import gpu_util as G
    import terarkdb_amd.capi as capi
    eng = G.engine()
    assert all(v == 0 for v in eng.stats().values())
    host = O.splitmix(64 * 4096, 3)
    dev = G.to_dev(host)
    out = torch.zeros(64, dtype=torch.int32, device="cuda")
    eng.blocks_stamp(dev, 64, 4096, 4096, out, flags=capi.F_MASK)
    eng.blocks_stamp(dev, 32, 4096, 4096, out, flags=capi.F_MASK)
    eng.extend_long(dev, [0], [200000], [0], out)
    assert eng.extend(0, host[:1000]) == O.value(host[:1000])
    img = np.frombuffer(O.build_wal_image(False, n_random=20, edge=False), dtype=np.uint8)
    blocks = torch.zeros(((img.size + 32767) // 32768, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    eng.counts_reset(counts)
    eng.wal_verify(G.to_dev(img), img.size, 123, blocks, counts)
    st = eng.stats()
    # two stamps, the long buffer and the short host input; the log scan has a kernel of its own
    assert st["batch_launches"] == 4 and st["batch_buffers"] >= 64 + 32
    assert (st["long_calls"], st["long_bytes"]) == (1, 200000)
    assert (st["log_images"], st["log_bytes"]) == (1, img.size)
    assert st["host_calls"] == 1 and st["host_bytes_in"] == 1000 and st["host_nanos"] > 0
    assert st["host_lanes"] == 1  # one thread: one lane
    assert capi.lib.tcrc_reset_stats(eng.ctx) == 0
    assert all(v == 0 for k, v in eng.stats().items() if k != "host_lanes")
    eng.close()


def test_calls_that_wait_with_a_stream_synchronisation_give_the_same_results():
    """TCRC_NO_POLL=1 makes short calls wait with hipStreamSynchronize instead of watching the kernel's completion
    word; the results must not depend on it (a fresh process: the setting is read when a context is created)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import oracle_lib as O
import terarkdb_amd.capi as capi
eng = capi.Engine(0)
rng = np.random.default_rng(5)
pool = rng.integers(0, 256, 1 << 22, dtype=np.uint8)
for n in (1, 100, 4096, 9000, 70000, 300000, 1 << 22):
    init = int(rng.integers(0, 1 << 32))
    assert eng.extend(init, pool[:n]) == O.extend(init, pool[:n].tobytes()), n
eng.close()
print("same")
""" % (os.path.dirname(here), here)
    env = dict(os.environ, TCRC_NO_POLL="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("same"), out.stderr[-2000:]
