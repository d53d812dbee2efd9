"""Records of the persistent block cache (utilities/persistent_cache/block_cache_tier_file.cc:55-192)
through the engine: serialized bytes and Deserialize verdicts against records laid out here with the CPU
oracle's crc32c.  The reference holds no stored cache file; the layout follows its source."""
import ctypes as C
import struct

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
MAGIC = 0xfefa


@pytest.fixture(scope="module")
def eng():
    import gpu_util as G
    e = G.engine()
    yield e
    e.close()


def oracle_record(key, val):
    hdr0 = struct.pack("<IIII", MAGIC, 0, len(key), len(val))
    crc = O.extend(O.extend(O.extend(0, hdr0), key), val)  # ComputeCRC, :108-118
    return struct.pack("<IIII", MAGIC, crc, len(key), len(val)) + key + val


def _pairs(rng):
    sizes = [(0, 0), (1, 0), (0, 1), (24, 2000), (7, 15), (16, 4096), (100, 70000), (3, 200000)]
    sizes += [(int(rng.integers(1, 64)), int(rng.integers(0, 9000))) for _ in range(200)]
    return [(rng.integers(0, 256, k, dtype=np.uint8).tobytes(), rng.integers(0, 256, v, dtype=np.uint8).tobytes())
            for k, v in sizes]


def _u64(a):
    return np.array(a, dtype=np.uint64)


def _u32(a):
    return np.array(a, dtype=np.uint32)


def test_serialize_matches_records_built_with_the_oracle(eng):
    import terarkdb_amd.capi as capi
    rng = np.random.default_rng(1)
    pairs = _pairs(rng)
    blob = b"".join(k + v for k, v in pairs)
    base = np.frombuffer(blob, dtype=np.uint8).copy() if blob else np.zeros(1, dtype=np.uint8)
    koff, klen, voff, vlen, pos = [], [], [], [], 0
    for k, v in pairs:
        koff.append(pos)
        klen.append(len(k))
        voff.append(pos + len(k))
        vlen.append(len(v))
        pos += len(k) + len(v)
    want = b"".join(oracle_record(k, v) for k, v in pairs)
    out = np.zeros(len(want), dtype=np.uint8)
    n = C.c_size_t(0)
    koff, klen, voff, vlen = _u64(koff), _u32(klen), _u64(voff), _u32(vlen)  # kept alive across the calls
    args = (eng.ctx, base.ctypes.data, koff.ctypes.data, klen.ctypes.data, voff.ctypes.data, vlen.ctypes.data,
            len(pairs))
    assert capi.lib.tcrc_shim_cache_records_serialize(*args, out.ctypes.data, out.size, C.byref(n)) == 0
    assert n.value == len(want) and out.tobytes() == want
    # sizing call
    assert capi.lib.tcrc_shim_cache_records_serialize(*args, None, 0, C.byref(n)) == -3 and n.value == len(want)


def test_deserialize_verdicts(eng):
    import terarkdb_amd.capi as capi
    rng = np.random.default_rng(2)
    pairs = _pairs(rng)
    recs = [oracle_record(k, v) for k, v in pairs]
    offs, sizes, pos = [], [], 5
    for r in recs:
        offs.append(pos)
        sizes.append(len(r))
        pos += len(r) + int(rng.integers(0, 9))
    img = np.zeros(pos + 32, dtype=np.uint8)
    for o, r in zip(offs, recs):
        img[o:o + len(r)] = np.frombuffer(r, dtype=np.uint8)
    want = [1] * len(recs)

    def damage(i, at, bit=1):
        img[offs[i] + at] ^= bit
        want[i] = 0

    damage(3, 0)                      # magic
    damage(4, 5, 0x80)                # stored crc
    damage(5, 16 + 3)                 # key byte
    damage(6, sizes[6] - 1, 0x40)     # last value byte
    damage(7, 16 + 100 + 150000)      # deep inside a long value
    damage(9, 9)                      # key size: no longer adds up to the record's size
    damage(10, 12)                    # value size
    sizes2 = list(sizes)
    sizes2[12] -= 1                   # address shorter than the record
    want[12] = 0
    offs2, sizes3 = offs + [1, 2], sizes2 + [15, 0]  # too short to hold a header (:164-166)
    want += [0, 0]
    ok = np.full(len(offs2), 9, dtype=np.uint8)
    a_offs, a_sizes = _u64(offs2), _u32(sizes3)
    rc = capi.lib.tcrc_shim_cache_records_check(eng.ctx, img.ctypes.data, img.size, a_offs.ctypes.data,
                                                a_sizes.ctypes.data, len(offs2), ok.ctypes.data)
    assert rc == 0
    assert ok.tolist() == want
    # an address outside the image is an argument error
    b_offs, b_sizes = _u64([img.size - 4]), _u32([16])
    assert capi.lib.tcrc_shim_cache_records_check(eng.ctx, img.ctypes.data, img.size, b_offs.ctypes.data,
                                                  b_sizes.ctypes.data, 1, ok.ctypes.data) == -1
