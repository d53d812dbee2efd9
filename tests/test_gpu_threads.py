"""Concurrent use of one engine from several host threads (the reference's Extend is re-entrant and is
called from write, flush, compaction and recovery threads at once: db/db_impl_write.cc:844-854)."""
import threading

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_entry_points_from_four_threads():
    import gpu_util as G
    import terarkdb_amd.capi as capi
    eng = G.engine()
    img = O.build_wal_image(False, n_random=120, seed=5)
    want_blocks = O.wal_scan_blocks(img, 123)
    errors = []

    def worker(t):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            rng = np.random.default_rng(t)
            with torch.cuda.stream(stream):
                n, blen = 3000 + 17 * t, 4096
                host = O.splitmix(n * blen, 40 + t)
                dev = G.to_dev(host)
                want = O.sst_trailer_batch(host, blen, blen, n, 0)
                out = torch.zeros(n, dtype=torch.int32, device="cuda")
                dimg = G.to_dev(np.frombuffer(img, dtype=np.uint8).copy())
                nb = (len(img) + 32767) // 32768
                blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")
                counts = torch.zeros(3, dtype=torch.int64, device="cuda")
                long_out = torch.zeros(1, dtype=torch.int32, device="cuda")
                for it in range(25):
                    eng.blocks_stamp(dev, n, blen, blen, out, suffix_byte=0, flags=capi.F_MASK,
                                     stream=stream.cuda_stream)
                    eng.counts_reset(counts, stream=stream.cuda_stream)
                    eng.wal_verify(dimg, len(img), 123, blocks, counts, stream=stream.cuda_stream)
                    off, ln = int(rng.integers(0, 1000)), int(rng.integers(300000, n * blen - 1000))
                    eng.extend_long(dev, [off], [ln], [t], long_out, stream=stream.cuda_stream)
                    piece = host[off:off + 5000 + it]
                    got_small = eng.extend(7, piece)  # host-pointer path, its own stream and staging
                    stream.synchronize()
                    assert (out.cpu().numpy().view(np.uint32) == want).all()
                    assert (blocks.cpu().numpy().view(np.uint32) == want_blocks).all()
                    assert int(long_out.cpu().numpy().view(np.uint32)[0]) == O.extend(t, host[off:off + ln], "hw")
                    assert got_small == O.extend(7, piece)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    eng.close()
    assert not errors, errors
