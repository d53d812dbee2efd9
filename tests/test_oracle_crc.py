"""Pins the CPU oracle's CRC arithmetic to the reference's own known answers
(util/crc32c_test.cc) and to values captured from a build of the reference."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "crc32c_known_answers.json")))
CAP = json.load(open(os.path.join(HERE, "golden", "reference_captured.json")))
VARIANTS = ["table", "bitwise", "hw"]


@pytest.fixture(scope="module")
def fnv_buf():
    return O.fnv_chain(KA["fnv_chain_buffer_size"])


@pytest.fixture(scope="module")
def S():
    return O.splitmix(4 * 1024 * 1024 + 64, CAP["splitmix_state"])


def test_hw_path_is_really_exercised_here():
    # the container and the GPU box hosts are x86 with SSE4.2+PCLMUL; if not, hw == table path
    assert O.lib().oracle_crc32c_hw_available() in (0, 1)


@pytest.mark.parametrize("variant", VARIANTS)
def test_rfc3720_vectors(variant):
    # util/crc32c_test.cc:67-94
    for v in KA["rfc3720"]:
        assert O.value(bytes.fromhex(v["hex"]), variant) == int(v["crc"], 16), v["name"]


@pytest.mark.parametrize("variant", ["table", "hw"])
def test_fnv_chain_table_single_and_stitched(variant, fnv_buf):
    # util/crc32c_test.cc:96-111: one-shot, then Extend over two halves
    for off, length, notcrc in KA["fnv_chain"]:
        want = (~notcrc) & 0xffffffff
        data = fnv_buf[off:off + length]
        assert O.value(data, variant) == want, (off, length)
        half = length // 2
        part = O.value(data[:half], variant)
        assert O.extend(part, data[half:], variant) == want, (off, length, "stitched")


def test_fnv_chain_small_entries_bitwise(fnv_buf):
    for off, length, notcrc in KA["fnv_chain"]:
        if length <= 64:
            assert O.value(fnv_buf[off:off + length], "bitwise") == (~notcrc) & 0xffffffff


@pytest.mark.parametrize("variant", VARIANTS)
def test_values_extend_mask(variant):
    # util/crc32c_test.cc:114-126
    assert O.value(b"a", variant) != O.value(b"foo", variant)
    e = KA["extend_identity"]
    assert O.value(e["whole"].encode(), variant) == O.extend(O.value(e["first"].encode(), variant),
                                                             e["second"].encode(), variant)
    crc = O.value(b"foo", variant)
    assert crc != O.mask(crc)
    assert crc != O.mask(O.mask(crc))
    assert crc == O.unmask(O.mask(crc))
    assert crc == O.unmask(O.unmask(O.mask(O.mask(crc))))


def test_captured_small_values():
    for s, h in CAP["value_ascii"].items():
        for variant in VARIANTS:
            assert O.value(s.encode(), variant) == int(h, 16)
    assert O.mask(O.value(b"foo")) == int(CAP["mask_of_value_foo"], 16)
    for t, h in enumerate(CAP["wal_type_crc_0_to_8"]):
        assert O.value(bytes([t])) == int(h, 16)


def test_captured_splitmix_prefixes(S):
    assert bytes(S[:16]).hex() == CAP["splitmix_first16_hex"]
    for n, h in CAP["value_of_S_prefix"]:
        assert O.value(S[:n], "table") == int(h, 16), n
        assert O.value(S[:n], "hw") == int(h, 16), n
    for e in CAP["extend_of_S"]:
        d = S[e["offset"]: e["offset"] + e["length"]]
        for variant in ("table", "hw"):
            assert O.extend(int(e["init"], 16), d, variant) == int(e["crc"], 16)


def test_captured_sst_trailers():
    for t in CAP["sst_trailer"]:
        blk = np.full(t["length"], t["fill_byte"], dtype=np.uint8)
        crc = O.value(blk)
        if t["covers_type_byte"]:
            crc = O.extend(crc, bytes([t["type"]]))
        assert crc == int(t["crc"], 16)
        assert O.mask(crc) == int(t["masked"], 16)
        if t["covers_type_byte"]:
            for hw in (0, 1):
                got = O.lib().oracle_sst_trailer_crc(blk.ctypes.data, blk.size, t["type"], hw)
                assert got == int(t["masked"], 16)


def test_captured_sst_corpus_prefix_consistency():
    # the full 1 GiB fold is checked by the GPU tests and the bench; here: the first
    # 4 MiB of the same stream, table path against hardware path, block by block
    S4 = O.splitmix(4 * 1024 * 1024, CAP["splitmix_state"])
    a = O.sst_trailer_batch(S4, 8192, 8192, 512, 0, hw=False)
    b = O.sst_trailer_batch(S4, 8192, 8192, 512, 0, hw=True)
    assert (a == b).all()
    assert a[0] == O.mask(O.extend(int(dict((n, h) for n, h in CAP["value_of_S_prefix"])[8192], 16), b"\0"))


def test_hw_against_table_all_alignments_and_lengths():
    rng = np.random.default_rng(7)
    buf = rng.integers(0, 256, size=70000, dtype=np.uint8)
    lens = list(range(0, 300)) + [1023, 1024, 1025, 3071, 3072, 3073, 6144, 9215, 9216, 9217, 32761, 65536]
    for a in range(0, 16):
        for n in lens[:: (1 if a < 2 else 7)]:
            init = int(rng.integers(0, 2**32))
            d = buf[a:a + n]
            assert O.extend(init, d, "hw") == O.extend(init, d, "table"), (a, n)


def test_sst_check_block():
    import ctypes as C
    blk = O.splitmix(4096, 5)
    trailer = O.lib().oracle_sst_trailer_crc(blk.ctypes.data, blk.size, 0, 0)
    img = np.concatenate([blk, np.array([0], np.uint8), np.frombuffer(int(trailer).to_bytes(4, "little"), np.uint8)])
    actual, stored = C.c_uint32(), C.c_uint32()
    assert O.lib().oracle_sst_check_block(img.ctypes.data, 4096, C.byref(actual), C.byref(stored)) == 1
    img[100] ^= 0x80  # the corruption the reference's corruption_test injects
    assert O.lib().oracle_sst_check_block(img.ctypes.data, 4096, C.byref(actual), C.byref(stored)) == 0
    assert actual.value != stored.value
