"""Byte-level facts about the block-based table format, written out by hand from what the reference states, and
held against tests/sst_files.py (the writer of every table file the verification tests read) and against the
product's own footer code.  The reference ships no table file, so whole-file parity stays unpinned; what these
tests pin is that the writer follows the format as the reference's sources and tests describe it, byte for byte,
instead of merely agreeing with the parser it feeds.

Sources: entry and block-trailer layout table/block_builder.cc:10-32,137-197; packed restart count
table/data_block_footer.cc:25-40; delta-encoded index values table/block.cc:575-606 and
table/block_test.cc:157-186; handles table/format.h:71, table/format.cc:56-69; footers table/format.cc:113-148,
table/format.h:145-155, values of table/table_test.cc:2935-3036; block trailer table/format.h:190-196,
table/block_based_table_builder.cc:591-631; varsigned util/coding.h:79-84,286."""
import ctypes as C
import struct

import oracle_lib as O
import sst_files as S
import terarkdb_amd.capi as capi


def test_size_constants_of_the_format():
    # BlockHandle::kMaxEncodedLength = 10 + 10; Footer: 2 * 20 + 8 and 1 + 2 * 20 + 4 + 8; kBlockTrailerSize = 5
    assert S.TRAILER == 5
    assert S.MAGIC == 0x88e241b785f4cff7 and S.LEGACY_MAGIC == 0xdb4775248b80fb57
    out = (C.c_uint8 * 64)()
    assert capi.lib.tcrc_shim_footer_encode(1, 0, 1, 10, 5, 20, 15, out) == 2 * 20 + 8
    assert capi.lib.tcrc_shim_footer_encode(1, 2, 0, 10, 5, 20, 15, out) == 1 + 2 * 20 + 4 + 8
    # the longest handle: two ten-byte varint64
    assert len(S.handle_bytes(2**64 - 1, 2**64 - 1)) == 20


def test_varints_and_varsigned_by_hand():
    assert S.varint(0) == b"\x00" and S.varint(127) == b"\x7f" and S.varint(128) == b"\x80\x01"
    assert S.varint(300) == b"\xac\x02" and S.varint(2**32) == b"\x80\x80\x80\x80\x10"
    # i64ToZigzag: (l << 1) ^ (l >> 63)
    assert [S.zigzag(v) for v in (0, -1, 1, -2, 2, 2**31)] == [0, 1, 2, 3, 4, 2**32]
    assert S.zigzag(-2**63) == 2**64 - 1
    assert S.handle_bytes(10, 5) == b"\x0a\x05" and S.handle_bytes(300, 2**14) == b"\xac\x02\x80\x80\x01"


def test_data_block_bytes_by_hand():
    # restart interval 2: entries 0 and 2 are restart points (shared = 0), entry 1 shares "appl" with entry 0
    b = S.BlockBuilder(restart_interval=2)
    b.add(b"apple", b"A")
    b.add(b"apply", b"BB")
    b.add(b"banana", b"")
    want = (b"\x00\x05\x01" + b"apple" + b"A"        # shared 0, unshared 5, value_length 1
            + b"\x04\x01\x02" + b"y" + b"BB"          # shared 4, unshared 1, value_length 2
            + b"\x00\x06\x00" + b"banana"             # restart point: whole key; empty value
            + struct.pack("<II", 0, 15)               # restarts[0], restarts[1]: offsets of entries 0 and 2
            + struct.pack("<I", 2))                   # num_restarts (binary-search index type: high bit clear)
    assert b.finish() == want
    # a block with no entries still has one restart point at offset 0 (BlockBuilder's constructor)
    assert S.BlockBuilder().finish() == struct.pack("<II", 0, 1)
    # without prefix compression every entry stores its whole key but restart points still come every interval
    b = S.BlockBuilder(restart_interval=3, prefix_compress=False)
    for k in (b"aa", b"ab", b"ac", b"ad"):
        b.add(k, b"v")
    assert b.finish() == (b"\x00\x02\x01aav" + b"\x00\x02\x01abv" + b"\x00\x02\x01acv" + b"\x00\x02\x01adv"
                          + struct.pack("<III", 0, 18, 2))


def test_delta_encoded_index_block_bytes_by_hand():
    # format_version 4 index block: <shared><non_shared> only; the value is the full handle (varint64 offset,
    # varint64 size) when shared == 0, else varsigned(size - size of the previous handle); the reader rebuilds
    # the offset as previous offset + previous size + 5
    handles = [(0, 100), (105, 90), (200, 130)]
    keys = [b"k01", b"k02", b"m"]
    b = S.BlockBuilder(restart_interval=16, delta_values=True)
    last = None
    for k, (off, size) in zip(keys, handles):
        delta = S.varint(S.zigzag(size - last)) if last is not None else b""
        b.add(k, S.handle_bytes(off, size), delta)
        last = size
    want = (b"\x00\x03" + b"k01" + b"\x00\x64"     # restart point: handle (0, 100)
            + b"\x02\x01" + b"2" + b"\x13"         # shares "k0": delta 90 - 100 = -10 -> zigzag 19
            + b"\x00\x01" + b"m" + b"\xc8\x01\x82\x01"  # shares nothing, so the FULL handle (200, 130) again
            + struct.pack("<II", 0, 1))
    assert b.finish() == want
    # the offsets the reader derives for delta entries are where the writer of the test files puts the blocks
    assert handles[1][0] == handles[0][0] + handles[0][1] + S.TRAILER


def test_footer_bytes_by_hand():
    # the handles of table/table_test.cc:2935-3036: metaindex (10, 5), index (20, 15)
    magic = struct.pack("<II", 0x85f4cff7, 0x88e241b7)          # low word first, then high word
    legacy_magic = struct.pack("<II", 0x8b80fb57, 0xdb477524)
    new_xxhash_v1 = b"\x02" + b"\x0a\x05" + b"\x14\x0f" + bytes(36) + struct.pack("<I", 1) + magic
    legacy = b"\x0a\x05" + b"\x14\x0f" + bytes(36) + legacy_magic
    assert len(new_xxhash_v1) == 53 and len(legacy) == 48
    out = (C.c_uint8 * 64)()
    n = capi.lib.tcrc_shim_footer_encode(2, 1, 0, 10, 5, 20, 15, out)
    assert bytes(out[:n]) == new_xxhash_v1
    n = capi.lib.tcrc_shim_footer_encode(1, 0, 1, 10, 5, 20, 15, out)
    assert bytes(out[:n]) == legacy
    # and the writer of the test files ends its tables with the same bytes
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for legacy_footer, version in ((False, 2), (True, 0)):
            path = os.path.join(d, "t.sst")
            lay = S.build_table(path, n_data=3, format_version=version, legacy_footer=legacy_footer)
            tail = open(path, "rb").read()[-(48 if legacy_footer else 53):]
            mo, ms = lay.handle("metaindex")
            io_, is_ = lay.handle("index")
            n = capi.lib.tcrc_shim_footer_encode(1, version, int(legacy_footer), mo, ms, io_, is_, out)
            assert bytes(out[:n]) == tail
            # metaindex and index handles point at block contents; each block is followed by its 5-byte trailer
            data = open(path, "rb").read()
            for role, off, size in lay.order:
                t = data[off + size:off + size + 5]
                assert t[0] == 0  # kNoCompression
                crc = O.extend(O.value(data[off:off + size]), t[:1])  # crc32c(block || type)
                assert struct.unpack("<I", t[1:])[0] == O.mask(crc), role
