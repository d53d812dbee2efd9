"""Host logic of the whole-file table verification (footer, metaindex, index walking, extents, status
texts) with the checksum arithmetic supplied by the CPU oracle through the callback seam."""
import pytest

import sst_cases
import sst_verify_util as U


@pytest.mark.parametrize("case", sst_cases.ALL, ids=lambda f: f.__name__)
def test_sst_verify_host_logic(case, tmp_path):
    case(U.verify_with_oracle, str(tmp_path))


def _footer_roundtrip(checksum, version, legacy, handles=(10, 5, 20, 15), prefix=b""):
    import ctypes as C
    import terarkdb_amd.capi as capi
    out = (C.c_uint8 * 64)()
    n = capi.lib.tcrc_shim_footer_encode(checksum, version, int(legacy), *handles, out)
    enc = prefix + bytes(out[:n])
    ck, ver, magic = C.c_int(-1), C.c_uint32(99), C.c_uint64(0)
    h = (C.c_uint64 * 4)()
    msg = C.create_string_buffer(200)
    rc = capi.lib.tcrc_shim_footer_decode(enc, len(enc), C.byref(ck), C.byref(ver), C.byref(magic), h, msg, 200)
    return n, enc, rc, ck.value, ver.value, magic.value, tuple(h), msg.value.decode()


def test_footer_round_trips_like_the_reference_test():
    """table/table_test.cc:2918-3036 (block-based cases): every checksum kind and version survives
    EncodeTo/DecodeFrom, the legacy magic number is up-converted."""
    import sst_files as S
    magic = S.MAGIC
    n, enc, rc, ck, ver, mg, h, msg = _footer_roundtrip(1, 0, True)   # upconvert legacy block based
    assert (n, rc, ck, ver, mg, h) == (48, 0, 1, 0, magic, (10, 5, 20, 15)), msg
    assert enc[-8:] == S.LEGACY_MAGIC.to_bytes(8, "little")
    for checksum, version in ((2, 1), (3, 1), (1, 2), (1, 4), (0, 3)):  # xxhash, xxhash64, version 2, ...
        n, enc, rc, ck, ver, mg, h, msg = _footer_roundtrip(checksum, version, False)
        assert (n, rc, ck, ver, mg, h) == (53, 0, checksum, version, magic, (10, 5, 20, 15)), msg
    # large handles use every varint length; the decoder finds the footer at the end of a longer tail
    big = (2**40 + 3, 2**33 + 1, 2**62 + 7, 2**31)
    n, enc, rc, ck, ver, mg, h, msg = _footer_roundtrip(1, 2, False, big, prefix=b"\x55" * 100)
    assert (rc, ck, ver, h) == (0, 1, 2, big), msg
    # the same bytes as the python builder of the test files writes
    f = bytes([1]) + S.handle_bytes(10, 5) + S.handle_bytes(20, 15)
    f += bytes(41 - len(f)) + (2).to_bytes(4, "little") + magic.to_bytes(8, "little")
    assert _footer_roundtrip(1, 2, False)[1] == f
    # errors (table/format.cc:188-197,60-69)
    import ctypes as C
    import terarkdb_amd.capi as capi
    msg = C.create_string_buffer(200)
    short_new = f[-50:]  # new-style magic but fewer than 53 bytes
    assert capi.lib.tcrc_shim_footer_decode(short_new, len(short_new), None, None, None, None, msg, 200) == -1
    assert msg.value.decode() == "Corruption: input is too short to be an sstable"
    bad = bytearray(f)
    bad[1:21] = b"\xff" * 20  # handle varints that never end
    assert capi.lib.tcrc_shim_footer_decode(bytes(bad), len(bad), None, None, None, None, msg, 200) == -1
    assert msg.value.decode() == "Corruption: bad block handle"
