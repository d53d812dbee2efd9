"""Whole-file verification scenarios shared by the CPU (callback seam) and GPU (engine) tests.
`verify(path, **kw)` returns (rc, message, stats[7])."""
import os

import numpy as np

import sst_files as S


def _check(verify, lay, want_rc=None):
    want = S.expected_status(lay)
    rc, msg, st = verify(lay.path)
    assert msg == want, (msg, want)
    assert rc == (0 if want == "OK" else 1)
    return st


VARIANTS = [
    dict(),
    dict(format_version=4, index_restart=16),
    dict(format_version=4, index_restart=1),
    dict(format_version=0, legacy_footer=True),
    dict(index="two_level", n_data=45, partition_entries=7),
    dict(index="two_level", format_version=4, index_restart=16, n_data=33, partition_entries=5),
    dict(filter_bytes=5000, range_del=True, compression_dict=700, compressed_every=3),
    dict(with_properties=False),
    dict(n_data=1, block_bytes=300),
    dict(n_data=0),
]


def clean_files(verify, tmp):
    for i, kw in enumerate(VARIANTS):
        lay = S.build_table(os.path.join(tmp, "clean%d.sst" % i), seed=10 + i, **kw)
        st = _check(verify, lay)
        assert st[0] == lay.file_size
        assert (st[1], st[2], st[3]) == (lay.n_meta, lay.n_index, lay.n_data), (kw, st)
        assert st[5] == 0
        covered = sum(size + 1 for _, _, size in lay.order)
        assert st[4] == covered


def corrupt_each_role(verify, tmp):
    """One flipped bit in every kind of block (contents, type byte, stored checksum), one at a time."""
    rng = np.random.default_rng(7)
    for i, kw in enumerate(VARIANTS[:8]):
        base = S.build_table(os.path.join(tmp, "base%d.sst" % i), seed=30 + i, **kw)
        raw = open(base.path, "rb").read()
        for role, off, size in base.order:
            if role.startswith("data") and int(role[4:]) % 6 not in (0, 5):
                continue
            for where in ("contents", "type", "crc"):
                if where == "contents" and size == 0:
                    continue
                pos = {"contents": off + int(rng.integers(0, max(size, 1))), "type": off + size,
                       "crc": off + size + 1 + int(rng.integers(0, 4))}[where]
                path = os.path.join(tmp, "c.sst")
                open(path, "wb").write(raw)
                S.flip(path, pos, int(rng.integers(0, 8)))
                lay = base
                lay_path, lay.path = lay.path, path
                try:
                    want = S.expected_status(lay)
                    rc, msg, st = verify(path)
                finally:
                    lay.path = lay_path
                if role == "index" and where == "type":
                    # a flipped type byte also fails the checksum, which is reported first
                    pass
                assert want != "OK"
                assert msg == want, (kw, role, where, msg, want)
                assert rc == 1


def several_bad_blocks(verify, tmp):
    """With many bad blocks the status is the first in visiting order; the count sees all data blocks."""
    rng = np.random.default_rng(8)
    for i, kw in enumerate([dict(n_data=300, block_bytes=2000), dict(index="two_level", n_data=120, partition_entries=9)]):
        lay = S.build_table(os.path.join(tmp, "multi%d.sst" % i), seed=50 + i, **kw)
        datas = [(r, o, s) for r, o, s in lay.order if r.startswith("data")]
        picks = rng.choice(len(datas), 11, replace=False)
        for k in picks:
            _, o, s = datas[k]
            S.flip(lay.path, o + int(rng.integers(0, s)), 3)
        rc, msg, st = verify(lay.path, extent_bytes=1 << 20)
        assert msg == S.expected_status(lay) and rc == 1
        assert st[5] == 11
        assert st[6] >= 3  # several extents


def structural_errors(verify, tmp):
    # too short
    p = os.path.join(tmp, "short.sst")
    open(p, "wb").write(b"x" * 20)
    rc, msg, _ = verify(p)
    assert (rc, msg) == (1, "Corruption: file is too short (20 bytes) to be an sstable: " + p)
    # wrong magic
    lay = S.build_table(os.path.join(tmp, "magic.sst"))
    S.flip(lay.path, lay.file_size - 1, 0)
    rc, msg, _ = verify(lay.path)
    assert rc == 1 and msg.startswith("Corruption: Bad table magic number: expected 9863518390377041911, found ")
    assert msg.endswith(" in " + lay.path)
    # the reference's other table formats are declined, not called corrupt
    for magic in (0x8242229663bf9564, 0x4f3418eb7a8f13b8, 0x926789d0c5f17873, 0x1122334455667788):
        lay = S.build_table(os.path.join(tmp, "other.sst"), n_data=3)
        raw = bytearray(open(lay.path, "rb").read())
        raw[-8:] = magic.to_bytes(8, "little")
        open(lay.path, "wb").write(raw)
        rc, msg, _ = verify(lay.path)
        assert rc == 2 and msg.startswith("Not implemented: not a block-based table"), msg
    # unknown footer version
    lay = S.build_table(os.path.join(tmp, "ver.sst"), format_version=9)
    rc, msg, _ = verify(lay.path)
    assert (rc, msg) == (1, "Corruption: Unknown Footer version. Maybe this file was created with newer version of "
                            "RocksDB?")
    # checksum kinds: none is accepted without arithmetic, xxHash kinds are declined, others are unknown
    lay = S.build_table(os.path.join(tmp, "nochk.sst"), checksum=0)
    rc, msg, st = verify(lay.path)
    assert (rc, msg) == (0, "OK") and st[6] == 0 and st[3] == lay.n_data
    for kind in (2, 3):
        lay = S.build_table(os.path.join(tmp, "xx%d.sst" % kind), checksum=kind)
        rc, msg, _ = verify(lay.path)
        assert rc == 2 and msg == "Not implemented: block checksum type is not CRC32c: " + lay.path
    lay = S.build_table(os.path.join(tmp, "unk.sst"), checksum=9)
    rc, msg, _ = verify(lay.path)
    mo, ms = lay.handle("metaindex")
    assert (rc, msg) == (1, "Corruption: unknown checksum type 9 in %s offset %d size %d" % (lay.path, mo, ms))
    # truncated file: the footer is gone, whatever is at the end is not a magic number
    lay = S.build_table(os.path.join(tmp, "trunc.sst"))
    raw = open(lay.path, "rb").read()
    open(lay.path, "wb").write(raw[:len(raw) - 100])
    rc, msg, _ = verify(lay.path)
    assert rc == 1 and "Bad table magic number" in msg
    # a handle that points past the end of the file: footer rewritten with a huge index handle
    lay = S.build_table(os.path.join(tmp, "past.sst"))
    raw = bytearray(open(lay.path, "rb").read())
    mo, ms = lay.handle("metaindex")
    foot = bytes([1]) + S.handle_bytes(mo, ms) + S.handle_bytes(lay.file_size - 60, 5000)
    raw[-53:-12] = foot + bytes(41 - len(foot))
    open(lay.path, "wb").write(raw)
    rc, msg, _ = verify(lay.path)
    assert (rc, msg) == (1, "Corruption: truncated block read from %s offset %d, expected 5005 bytes, got 60" % (
        lay.path, lay.file_size - 60))
    # an undecodable handle in the metaindex is checked as block (0, 0), like the reference does
    lay = S.build_table(os.path.join(tmp, "badhandle.sst"), filter_bytes=100, bad_meta_handle=True)
    rc, msg, _ = verify(lay.path)
    assert msg == S.expected_status(lay) and rc == 1 and " offset 0 size 0" in msg
    # missing file
    rc, msg, _ = verify(os.path.join(tmp, "nope.sst"))
    assert rc == -1 and msg.startswith("IO error")


def compressed_index_is_declined(verify, tmp):
    lay = S.build_table(os.path.join(tmp, "cidx.sst"))
    off, size = lay.handle("index")
    raw = bytearray(open(lay.path, "rb").read())
    raw[off + size] = 1  # pretend snappy
    import struct
    import oracle_lib as O
    raw[off + size + 1:off + size + 5] = struct.pack("<I", O.mask(O.value(bytes(raw[off:off + size + 1]))))
    open(lay.path, "wb").write(raw)
    rc, msg, _ = verify(lay.path)
    assert rc == 2 and msg == "Not implemented: compressed index block (type 1) in " + lay.path


ALL = [clean_files, corrupt_each_role, several_bad_blocks, structural_errors, compressed_index_is_declined]
