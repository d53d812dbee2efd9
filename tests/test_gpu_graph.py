"""Device entry points that only enqueue, recorded into a HIP graph and replayed."""
import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_enqueue_only_entry_points_replay_from_a_graph():
    """The device entry points that only enqueue - block stamps and verifies, descriptor batches, the counter
    reset, a log scan without the record count - can be recorded into a HIP graph (a verification loop that is
    bound by launches): every launch of the graph gives the results of a direct call."""
    import gpu_util as G
    import terarkdb_amd.capi as capi
    eng = G.engine()
    n, blen = 4096, 4096
    host = O.splitmix(n * blen, 21)
    dev = G.to_dev(host)
    want = O.sst_trailer_batch(host, blen, blen, n, 1)
    img = np.frombuffer(O.build_wal_image(True, n_random=300, seed=12), dtype=np.uint8).copy()
    want_blocks = O.wal_scan_blocks(img.tobytes(), 123)
    dimg = G.to_dev(img)
    nb = (img.size + 32767) // 32768
    stored = torch.from_numpy(want.view(np.int32).copy()).cuda()
    stored[100] ^= 4  # one wrong trailer
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    stamped = torch.zeros(n, dtype=torch.int32, device="cuda")
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    counts = torch.zeros(capi.CNT_WORDS, dtype=torch.int64, device="cuda")
    wcounts = torch.zeros(capi.CNT_WORDS, dtype=torch.int64, device="cuda")
    blocks = torch.zeros((nb, 4), dtype=torch.int32, device="cuda")

    def enqueue(stream):
        eng.counts_reset(counts, stream=stream)
        eng.counts_reset(wcounts, stream=stream)
        eng.blocks_stamp(dev, n, blen, blen, stamped, suffix_byte=1, flags=capi.F_MASK, stream=stream)
        eng.blocks_verify(dev, n, blen, blen, counts, expected=stored, out=out, ok=ok, suffix_byte=1, stream=stream)
        eng.wal_verify(dimg, img.size, 123, blocks, wcounts, stream=stream, want_records=False)

    enqueue(torch.cuda.current_stream().cuda_stream)  # nothing is set up for the first time inside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        enqueue(torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        for t in (out, stamped, ok, blocks):
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert (stamped.cpu().numpy().view(np.uint32) == want).all(), rep
        assert (out.cpu().numpy().view(np.uint32) == want).all(), rep
        okh = ok.cpu().numpy()
        assert okh[100] == 0 and okh.sum() == n - 1, rep
        c = counts.cpu().numpy().view(np.uint64)
        assert int(c[capi.CNT_MISMATCH]) == 1 and int(c[capi.CNT_FIRST_BAD]) == 100 and int(c[capi.CNT_CHECKED]) == n, rep
        assert (blocks.cpu().numpy().view(np.uint32) == want_blocks).all(), rep
        wc = wcounts.cpu().numpy().view(np.uint64)
        assert int(wc[capi.CNT_MISMATCH]) == 0 and int(wc[capi.CNT_CHECKED]) == int(want_blocks[:, 0].sum()), rep
    eng.close()
