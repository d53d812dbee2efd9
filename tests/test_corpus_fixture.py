"""tests/golden/corpus64_folds.json (the 64 GiB multi-GPU verification corpus) against the CPU oracle and the value
captured from a build of the reference; the generating script is tools/make_corpus_folds.py."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "corpus64_folds.json")))
CAP = json.load(open(os.path.join(HERE, "golden", "reference_captured.json")))
GOLDEN = 0x9E3779B97F4A7C15


def block_bytes(b, blen=8192):
    return O.splitmix(blen, (301 + GOLDEN * (blen // 8) * b) & (2**64 - 1))


def test_first_gib_fold_is_the_captured_reference_value():
    assert FIX["xor_of_masked_per_gib"][0] == CAP["sst_1gib_splitmix"]["xor_of_masked"].lower().replace("0x", "")
    assert len(FIX["xor_of_masked_per_gib"]) == 64


def test_sampled_trailers_against_the_table_path():
    # every sampled block regenerated from the stream position alone, checksummed by the slice-by-4 variant
    # (the fixture was made with the SSE4.2 variant)
    for b, want in FIX["trailer_of_block"].items():
        got = O.sst_trailer_batch(block_bytes(int(b)), 8192, 8192, 1, 0, hw=False)[0]
        assert f"{int(got):08x}" == want, b
    # the stream position arithmetic: block 5 of the plain stream
    s = O.splitmix(6 * 8192, 301)
    assert (block_bytes(5) == s[5 * 8192:]).all()


@pytest.mark.gpu
def test_device_fill_matches_the_stream_at_any_position():
    import torch
    from terarkdb_amd.synth import splitmix_fill
    for first_block in (0, 3, 131072 * 9 + 17, 8388608 - 4):
        buf = torch.empty(4 * 8192, dtype=torch.uint8, device="cuda")
        splitmix_fill(buf, 301, first_word=first_block * 1024, chunk_bytes=8192 + 8)
        want = np.concatenate([block_bytes(first_block + i) for i in range(4)])
        assert (buf.cpu().numpy() == want).all(), first_block
