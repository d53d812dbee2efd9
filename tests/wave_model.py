"""CPU model of the batch kernel's arithmetic for ONE buffer (one group of G lanes), driven by the very
tables the library builds for the GPU (tcrc_debug_tables).  It follows terarkdb_amd/csrc/tcrc_engine.hip
step by step - address-aligned rows of 16*G bytes, four column states per lane, leading bytes masked,
trailing pad undone with x^(-8 tpad) split over UL lanes, Horner + shuffle reduction - so that the table
construction and the algebra can be checked against the oracle without a GPU."""
import ctypes as C

import numpy as np

import terarkdb_amd.capi as capi

P = 0x82f63b78


def tables(G):
    main = np.zeros(1024, dtype=np.uint32)
    n = C.c_size_t(0)
    capi.check(capi.lib.tcrc_debug_tables(G, None, None, 0, C.byref(n)))
    aux = np.zeros(n.value, dtype=np.uint32)
    capi.check(capi.lib.tcrc_debug_tables(G, main.ctypes.data, aux.ctypes.data, n.value, C.byref(n)))
    return main, aux


def fold(t, c):
    """Z_n through one 4 x 256 set."""
    return int(t[c & 0xff]) ^ int(t[256 + ((c >> 8) & 0xff)]) ^ int(t[512 + ((c >> 16) & 0xff)]) ^ int(t[768 + (c >> 24)])


def group_crc(data, off, length, init, G, main, aux, extra_rows=0, suffix=-1):
    """Extend(init, data[off:off+length]) the way a group of G lanes computes it; `off` also stands for
    the address alignment (the kernel aligns rows to addresses), extra_rows for the empty leading rows a
    buffer gets when a longer buffer shares its wave."""
    D = 16 * G
    log2g = G.bit_length() - 1
    UL = 4 if G <= 16 else 1
    r0 = ~init & 0xffffffff
    msg = bytes(data[off:off + length])
    if length >= 16:
        a0 = off % D
        span = a0 + length
        R = (span + D - 1) // D
        tpad = R * D - span
        rows = R + extra_rows
        s = [[0, 0, 0, 0] for _ in range(G)]
        for r in range(rows):
            for j in range(G):
                m = 16 * j - a0 - extra_rows * D + r * D
                col = bytearray(16)
                for k in range(16):
                    if 0 <= m + k < length:
                        col[k] = msg[m + k]
                    if 0 <= m + k < 4:
                        col[k] ^= (r0 >> (8 * (m + k))) & 0xff
                for q in range(4):
                    w = int.from_bytes(col[4 * q:4 * q + 4], "little")
                    s[j][q] = fold(main, s[j][q]) ^ w
        u = []
        for j in range(G):
            v = fold(aux, s[j][0]) ^ s[j][1]
            v = fold(aux, v) ^ s[j][2]
            v = fold(aux, v) ^ s[j][3]
            u.append(v)
        for st in range(log2g, 0, -1):
            d = 1 << (st - 1)
            t = [fold(aux[st * 1024:], x) for x in u]
            u = [u[j] ^ (t[j - d] if j >= d else t[j]) for j in range(G)]
        r = fold(aux, u[G - 1])
        if tpad:
            inv = aux[(1 + log2g) * 1024:]
            S = 32 // UL
            prod = 0
            for lane in range(UL):
                b = int(inv[tpad * UL + lane])
                for k in range(S):
                    if (r >> (31 - S * lane - k)) & 1:
                        prod ^= b
                    b = (b >> 1) ^ (P if b & 1 else 0)
            r = prod
    else:
        r = r0
        t0 = aux[768:1024]  # the one-byte step table is the last quarter of the Z_4 set
        for b in msg:
            r = int(t0[(r ^ b) & 0xff]) ^ (r >> 8)
    if suffix >= 0:
        r = int(aux[768 + ((r ^ suffix) & 0xff)]) ^ (r >> 8)
    return ~r & 0xffffffff
