"""Block-based table files for the whole-file verification tests.

The reference ships no .sst files, so these are written here from the format its sources
describe: block entries and restart arrays (table/block_builder.cc:115-197), the format_version 4
index-value delta encoding (table/block.cc:575-606), block trailers (table/format.h:190-196,
table/block_based_table_builder.cc:580-646), the meta blocks and their order
(table/block_based_table_builder.cc:877-956), property names (table/table_properties.cc:179-243,
table/block_based_table_factory.cc:600-609) and the footer (table/format.cc:113-148).
The block checksums come from the CPU oracle.  `Layout` records where every block went and
in which order BlockBasedTable::VerifyChecksum visits them
(table/block_based_table_reader.cc:2616-2695), so a test can predict the first error.
"""
import struct

import numpy as np

import oracle_lib as O

MAGIC = 0x88e241b785f4cff7
LEGACY_MAGIC = 0xdb4775248b80fb57
TRAILER = 5


def varint(v):
    out = bytearray()
    while v >= 128:
        out.append((v & 127) | 128)
        v >>= 7
    out.append(v)
    return bytes(out)


def zigzag(v):
    return ((v << 1) ^ (v >> 63)) & (2**64 - 1)


def handle_bytes(offset, size):
    return varint(offset) + varint(size)


class BlockBuilder:
    def __init__(self, restart_interval=16, delta_values=False, prefix_compress=True):
        self.interval = restart_interval
        self.delta_values = delta_values
        self.prefix_compress = prefix_compress
        self.buf = bytearray()
        self.restarts = [0]
        self.counter = 0
        self.last_key = b""

    def add(self, key, value, delta_value=None):
        shared = 0
        if self.counter >= self.interval:
            self.restarts.append(len(self.buf))
            self.counter = 0
        elif self.prefix_compress:
            m = min(len(key), len(self.last_key))
            while shared < m and key[shared] == self.last_key[shared]:
                shared += 1
        self.last_key = key
        non_shared = len(key) - shared
        if self.delta_values:
            self.buf += varint(shared) + varint(non_shared)
        else:
            self.buf += varint(shared) + varint(non_shared) + varint(len(value))
        self.buf += key[shared:]
        self.buf += delta_value if (shared != 0 and self.delta_values) else value
        self.counter += 1

    def finish(self):
        out = bytes(self.buf)
        for r in self.restarts:
            out += struct.pack("<I", r)
        return out + struct.pack("<I", len(self.restarts))


class Layout:
    """Where the blocks of one written table are, by role, plus the verification order."""

    def __init__(self, path):
        self.path = path
        self.order = []   # (role, offset, size) in the order VerifyChecksum visits them
        self.blocks = {}  # role -> (offset, size)
        self.file_size = 0
        self.checksum = 1

    def handle(self, role):
        return self.blocks[role]


def _trailer(contents, type_byte, checksum):
    if checksum == 1:
        crc = O.mask(O.extend(O.value(contents), bytes([type_byte])))
    else:
        crc = 0
    return bytes([type_byte]) + struct.pack("<I", crc)


def build_table(path, n_data=40, block_bytes=4096, format_version=2, index="binary", partition_entries=7,
                filter_bytes=0, range_del=False, compression_dict=0, checksum=1, legacy_footer=False,
                with_properties=True, index_restart=1, seed=1, compressed_every=0, bad_meta_handle=False):
    rng = np.random.default_rng(seed)
    out = bytearray()
    lay = Layout(path)
    lay.checksum = checksum
    delta = format_version >= 4

    def write_raw(role, contents, type_byte=0):
        off = len(out)
        out.extend(contents)
        out.extend(_trailer(contents, type_byte, checksum))
        lay.blocks[role] = (off, len(contents))
        return off, len(contents)

    # data blocks: real entries, sizes scattered around block_bytes
    data_handles = []
    last_keys = []
    key_no = 0
    for b in range(n_data):
        bb = BlockBuilder(16)
        target = int(block_bytes * (0.6 + 0.8 * rng.random()))
        while len(bb.buf) < target:
            key = b"key%010d" % key_no + struct.pack("<Q", (key_no << 8) | 1)
            bb.add(key, rng.integers(0, 256, int(rng.integers(1, 200)), dtype=np.uint8).tobytes())
            key_no += 1
        contents = bb.finish()
        type_byte = 0
        if compressed_every and b % compressed_every == compressed_every - 1:
            # verification never decompresses a data block: any bytes with a non-zero type do
            contents = rng.integers(0, 256, len(contents) // 2 + 1, dtype=np.uint8).tobytes()
            type_byte = 1
        data_handles.append(write_raw("data%d" % b, contents, type_byte))
        last_keys.append(b"key%010d" % (key_no - 1) + struct.pack("<Q", (key_no << 8) | 1))

    meta = {}  # metaindex key -> handle
    if filter_bytes:
        h = write_raw("filter", rng.integers(0, 256, filter_bytes, dtype=np.uint8).tobytes())
        meta[b"fullfilter.rocksdb.BuiltinBloomFilter"] = h

    def index_block(keys, handles):
        bb = BlockBuilder(index_restart, delta_values=delta)
        prev = None
        for k, (o, s) in zip(keys, handles):
            dv = None
            if delta and prev is not None:
                # only meaningful when the entry shares key bytes; handles must be adjacent then
                dv = varint(zigzag(s - prev[1]))
            bb.add(k, handle_bytes(o, s), dv)
            prev = (o, s)
        return bb.finish()

    partitions = []
    if index == "two_level":
        part_handles, part_keys = [], []
        for p0 in range(0, n_data, partition_entries):
            p1 = min(n_data, p0 + partition_entries)
            h = write_raw("partition%d" % len(part_handles), index_block(last_keys[p0:p1], data_handles[p0:p1]))
            partitions.append((h, list(range(p0, p1))))
            part_handles.append(h)
            part_keys.append(last_keys[p1 - 1])
        index_handle = write_raw("index", index_block(part_keys, part_handles))
    else:
        index_handle = write_raw("index", index_block(last_keys, data_handles))

    if compression_dict:
        meta[b"rocksdb.compression_dict"] = write_raw(
            "compression_dict", rng.integers(0, 256, compression_dict, dtype=np.uint8).tobytes())
    if range_del:
        bb = BlockBuilder(1)
        bb.add(b"a" + struct.pack("<Q", (5 << 8) | 15), b"c")
        meta[b"rocksdb.range_del"] = write_raw("range_del", bb.finish())
    if with_properties:
        props = {
            b"rocksdb.block.based.table.index.type": struct.pack("<I", {"binary": 0, "hash": 1, "two_level": 2}[index]),
            b"rocksdb.data.size": varint(len(out)),
            b"rocksdb.num.data.blocks": varint(n_data),
            b"rocksdb.format.version": varint(format_version),
            b"rocksdb.index.key.is.user.key": varint(0),
            b"rocksdb.index.value.is.delta.encoded": varint(1 if delta else 0),
            b"rocksdb.comparator": b"leveldb.BytewiseComparator",
            b"rocksdb.compression": b"NoCompression",
        }
        bb = BlockBuilder(1)
        for k in sorted(props):
            bb.add(k, props[k])
        meta[b"rocksdb.properties"] = write_raw("properties", bb.finish())

    bb = BlockBuilder(1)
    for i, k in enumerate(sorted(meta)):
        if bad_meta_handle and i == 0:
            bb.add(k, b"\xff\xff")  # undecodable handle: checked as (offset 0, size 0)
        else:
            bb.add(k, handle_bytes(*meta[k]))
    metaindex_handle = write_raw("metaindex", bb.finish())

    if legacy_footer:
        f = handle_bytes(*metaindex_handle) + handle_bytes(*index_handle)
        f += bytes(40 - len(f)) + struct.pack("<II", LEGACY_MAGIC & 0xffffffff, LEGACY_MAGIC >> 32)
    else:
        f = bytes([checksum]) + handle_bytes(*metaindex_handle) + handle_bytes(*index_handle)
        f += bytes(41 - len(f)) + struct.pack("<I", format_version) + struct.pack("<II", MAGIC & 0xffffffff, MAGIC >> 32)
    out.extend(f)
    with open(path, "wb") as fh:
        fh.write(out)
    lay.file_size = len(out)

    # the order VerifyChecksum visits the blocks in
    lay.order.append(("metaindex",) + metaindex_handle)
    role_of = {v: k for k, v in lay.blocks.items()}
    for i, k in enumerate(sorted(meta)):
        if bad_meta_handle and i == 0:
            lay.order.append(("offset0", 0, 0))
        else:
            lay.order.append((role_of[meta[k]],) + meta[k])
    lay.order.append(("index",) + index_handle)
    if index == "two_level":
        for (h, members) in partitions:
            lay.order.append((role_of[h],) + h)
            for b in members:
                lay.order.append(("data%d" % b,) + data_handles[b])
    else:
        for b in range(n_data):
            lay.order.append(("data%d" % b,) + data_handles[b])
    lay.n_meta = 1 + len(meta)
    lay.n_index = 1 + len(partitions)
    lay.n_data = n_data
    return lay


def expected_status(lay, path=None):
    """Status text VerifyChecksum would return for the file as it is now on disk (checksum kCRC32c)."""
    raw = open(lay.path, "rb").read()
    name = path or lay.path
    for role, off, size in lay.order:
        if off + size + TRAILER > len(raw):
            got = max(0, len(raw) - off)
            msg = "Corruption: truncated block read from %s offset %d, expected %d bytes, got %d" % (
                name, off, size + TRAILER, got)
        else:
            actual = O.value(raw[off:off + size + 1])
            stored = O.unmask(struct.unpack_from("<I", raw, off + size + 1)[0])
            if actual == stored:
                continue
            msg = "Corruption: block checksum mismatch: expected %d, got %d  in %s offset %d size %d" % (
                actual, stored, name, off, size)
        # a bad partition block ends the walk: the data blocks of later partitions are unreachable
        return msg
    return "OK"


def flip(path, offset, bit=0):
    with open(path, "r+b") as fh:
        fh.seek(offset)
        b = fh.read(1)[0]
        fh.seek(offset)
        fh.write(bytes([b ^ (1 << bit)]))


def build_uniform_table(path, n_data, block_len=8192, seed_state=301, index="binary"):
    """A large table written fast: data blocks are block_len bytes of the splitmix64 stream (verification
    never parses data blocks), trailers from the oracle's batch path; index, properties, metaindex and
    footer as in build_table.  Returns the Layout (order lists only the first/last few data blocks)."""
    lay = Layout(path)
    stride = block_len + TRAILER
    body = O.splitmix(n_data * block_len, seed_state).reshape(n_data, block_len)
    crcs = O.sst_trailer_batch(body, block_len, block_len, n_data, 0, hw=True)
    img = np.zeros((n_data, stride), dtype=np.uint8)
    img[:, :block_len] = body
    img[:, block_len + 1:] = crcs.astype("<u4").view(np.uint8).reshape(n_data, 4)
    out = bytearray(img.tobytes())
    del img, body

    def write_raw(role, contents):
        off = len(out)
        out.extend(contents)
        out.extend(_trailer(contents, 0, 1))
        lay.blocks[role] = (off, len(contents))
        return off, len(contents)

    handles = [(b * stride, block_len) for b in range(n_data)]
    keys = [b"key%010d" % b + b"\x01\0\0\0\0\0\0\0" for b in range(n_data)]

    def index_block(ks, hs):
        bb = BlockBuilder(1)
        for k, (o, s) in zip(ks, hs):
            bb.add(k, handle_bytes(o, s))
        return bb.finish()

    n_index = 1
    if index == "two_level":
        per = 256
        ph, pk = [], []
        for p0 in range(0, n_data, per):
            p1 = min(n_data, p0 + per)
            ph.append(write_raw("partition%d" % len(ph), index_block(keys[p0:p1], handles[p0:p1])))
            pk.append(keys[p1 - 1])
        index_handle = write_raw("index", index_block(pk, ph))
        n_index += len(ph)
    else:
        index_handle = write_raw("index", index_block(keys, handles))
    props = {
        b"rocksdb.block.based.table.index.type": struct.pack("<I", 2 if index == "two_level" else 0),
        b"rocksdb.num.data.blocks": varint(n_data),
        b"rocksdb.index.value.is.delta.encoded": varint(0),
    }
    bb = BlockBuilder(1)
    for k in sorted(props):
        bb.add(k, props[k])
    ph_ = write_raw("properties", bb.finish())
    bb = BlockBuilder(1)
    bb.add(b"rocksdb.properties", handle_bytes(*ph_))
    mh = write_raw("metaindex", bb.finish())
    f = bytes([1]) + handle_bytes(*mh) + handle_bytes(*index_handle)
    f += bytes(41 - len(f)) + struct.pack("<I", 2) + struct.pack("<II", MAGIC & 0xffffffff, MAGIC >> 32)
    out.extend(f)
    with open(path, "wb") as fh:
        fh.write(out)
    lay.file_size = len(out)
    lay.n_meta, lay.n_index, lay.n_data = 2, n_index, n_data
    lay.data_handles = handles
    return lay
