"""The host side of device log framing: the piece list the framing kernel walks (one entry per piece of at most
2 KiB of payload) against the physical records the CPU oracle's writer produces for the same records
(db/log_writer.cc:45-151).  No GPU: the list is obtained through a debug export of the library."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

PIECE = 4096


def plan_entries(lens, offs, recycle, block_offset):
    import terarkdb_amd.capi as capi
    n = len(lens)
    arr = (capi.Desc * max(n, 1))()
    for i in range(n):
        arr[i].offset, arr[i].len = offs[i], lens[i]
    fn = capi.lib.tcrc_debug_wal_plan_entries
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                   C.POINTER(C.c_uint32)]
    cnt, run = C.c_uint64(0), C.c_uint32(0)
    assert fn(arr, n, int(recycle), block_offset, None, 0, C.byref(cnt), C.byref(run)) == 0
    out = np.zeros((cnt.value, 2), dtype=np.uint64)
    assert fn(arr, n, int(recycle), block_offset, out.ctypes.data, cnt.value, C.byref(cnt), C.byref(run)) == 0
    return out, run.value


def decode(e):
    a, b = int(e[0]), int(e[1])
    return dict(src=a & ((1 << 40) - 1), plen=(a >> 40) & 0x1fff, q=(a >> 53) & 7, first=(a >> 56) & 1,
                last=(a >> 57) & 1, valid=(a >> 58) & 1, pad=(a >> 59) & 15, dst=b & ((1 << 40) - 1),
                flen=(b >> 40) & 0xffff, type=(b >> 56) & 15)


@pytest.mark.parametrize("recycle", [False, True])
@pytest.mark.parametrize("start", [0, 11, 32768 - 7, 32768 - 12, 20000])
def test_piece_list_matches_the_cpu_writers_records(recycle, start):
    hs = 11 if recycle else 7
    rnd = O.Random(301 + start)
    lens = [0, 1, 32768 - hs, 32768 - hs + 1, 65536, 100000, 17, 32761, 32762, 5, 2040, 2048, 2049, 4096, 4097, 0, 0]
    lens += [rnd.skewed(17) for _ in range(200)] + [rnd.uniform(200) for _ in range(300)]
    offs, pos = [], 11
    for ln in lens:
        offs.append(pos)
        pos += ln + 1
    # the oracle's image of the same records, continuing a log that stands at block offset `start`
    prefix_len = start - hs if start >= hs else None
    w = O.OracleLog(recycle=recycle, log_number=9)
    base = 0
    if start:
        if prefix_len is None:
            pytest.skip("a log cannot stand inside its first header")
        w.write(b"p" * prefix_len)
        base = w.written_bytes()
        assert base % 32768 == start % 32768
    for o, ln in zip(offs, lens):
        w.write(bytes((o + k) & 0xff for k in range(min(ln, 4))) + b"\0" * max(0, ln - 4))
    img = w.contents()
    w.close()
    phys, _ = O.list_physical(img, 9)
    phys = [r for r in phys if r[0] >= base]

    ent, run = plan_entries(lens, offs, recycle, start % 32768)
    assert run == 8 and len(ent) % run == 0
    frags, cur = [], None
    for i, raw in enumerate(ent):
        d = decode(raw)
        if i % run == 0:
            assert cur is None, "a fragment's pieces lie in two rounds"
        if not d["valid"]:
            assert cur is None, "an empty entry inside a fragment"
            assert int(raw[0]) == 0 and int(raw[1]) == 0
            continue
        if d["first"]:
            assert cur is None
            cur = dict(first_index=i, pieces=[d])
        else:
            assert cur is not None
            cur["pieces"].append(d)
        if d["last"]:
            assert cur["first_index"] // run == i // run, "a fragment's pieces lie in two rounds"
            frags.append(cur)
            cur = None
    assert cur is None
    assert len(frags) == len(phys)
    # the list is laid out for the kernel, not in file order: rounds of several-piece fragments first, then the
    # one-piece fragments longest first in classes of 256 bytes
    multi_rounds = {f["first_index"] // run for f in frags if len(f["pieces"]) > 1}
    assert multi_rounds == set(range(len(multi_rounds)))
    singles_after = [f["pieces"][0]["plen"] for f in frags if f["first_index"] // run >= len(multi_rounds)]
    classes = [(ln + 255) // 256 for ln in singles_after]
    assert classes == sorted(classes, reverse=True)
    frags.sort(key=lambda f: f["pieces"][0]["dst"])
    ri, consumed = -1, 0
    for f, r in zip(frags, phys):
        off, rtype, ln = int(r[0]), int(r[1]), int(r[2])
        ps = f["pieces"]
        c = len(ps)
        assert c == max(1, -(-ln // PIECE)) and c <= 8
        assert [p["q"] for p in ps] == list(range(c - 1, -1, -1))
        assert all(p["flen"] == ln and p["type"] == rtype for p in ps)
        assert all(p["plen"] == PIECE for p in ps[1:]) and sum(p["plen"] for p in ps) == ln
        assert all(p["pad"] == 0 for p in ps[1:])
        # payload position in the image, piece after piece; the header sits hs bytes before the first
        at = off - base + hs
        for p in ps:
            assert p["dst"] == at
            at += p["plen"]
        # zero bytes in front of the header: exactly what the oracle's image holds there beyond the previous record
        pad = ps[0]["pad"]
        assert pad <= 10 and img[off - pad:off] == b"\0" * pad
        # source bytes: a FULL or FIRST fragment begins the next logical record, the others continue it
        if rtype in (1, 2, 5, 6):
            ri, consumed = ri + 1, 0
        at = offs[ri] + consumed
        for p in ps:
            assert p["src"] == at
            at += p["plen"]
        consumed += ln
        if rtype in (1, 4, 5, 8):
            assert consumed == lens[ri]
    assert ri == len(lens) - 1
