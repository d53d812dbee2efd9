"""The kernel's arithmetic modelled on the CPU with the library's own tables against the oracle:
every geometry, every row alignment class, lengths around the row and burst boundaries."""
import numpy as np
import pytest

import oracle_lib as O
import wave_model as W


@pytest.mark.parametrize("G", [4, 8, 16, 64])
def test_tables_have_the_documented_shape(G):
    main, aux = W.tables(G)
    log2g = G.bit_length() - 1
    ul = 4 if G <= 16 else 1
    assert main.size == 1024 and aux.size == (1 + log2g) * 1024 + 16 * G * ul
    # Z_4's last quarter is the reference's one-byte table: table0_[1] and table0_[0x80] (util/crc32c.cc:51,83)
    assert int(aux[768 + 1]) == 0xf26b8303 and int(aux[768 + 0x80]) == 0x82f63b78
    # zero trailing bytes need no correction: the multiplier is 1 (bit 31 = x^0) for lane 0
    assert int(aux[(1 + log2g) * 1024]) == 0x80000000


@pytest.mark.parametrize("G", [4, 8, 16, 64])
def test_group_model_matches_oracle(G):
    main, aux = W.tables(G)
    D = 16 * G
    data = O.splitmix(8 * D + 4096 + 64, 7 + G)
    rng = np.random.default_rng(G)
    cases = []
    for ln in [0, 1, 3, 4, 5, 15, 16, 17, 31, 32, 33, D - 1, D, D + 1, 2 * D - 1, 2 * D, 2 * D + 5, 4 * D, 4 * D + 3, 5 * D - 2]:
        for off in [0, 1, 3, 4, 13, 15, 16, D - 4, D - 3, D - 1, D, D + 7]:
            cases.append((off, ln))
    for _ in range(60):
        cases.append((int(rng.integers(0, 3 * D)), int(rng.integers(0, 6 * D))))
    for i, (off, ln) in enumerate(cases):
        init = [0, 0xffffffff, 0x12345678, int(rng.integers(0, 2**32))][i % 4]
        want = O.extend(init, data[off:off + ln])
        assert W.group_crc(data, off, ln, init, G, main, aux) == want, (G, off, ln, init)
        if i % 5 == 0:  # a longer buffer in the same wave: empty leading rows; and the SST type byte
            assert W.group_crc(data, off, ln, init, G, main, aux, extra_rows=3) == want
            sfx = int(rng.integers(0, 256))
            assert W.group_crc(data, off, ln, init, G, main, aux, suffix=sfx) == O.extend(want, bytes([sfx]))
