"""tcrc_crc32c_extend: the export with the reference's own scalar signature
(`typedef uint32_t (*Function)(uint32_t, const char*, size_t)`, util/crc32c.cc:439,1225-1228).

With a GPU it must reproduce the table of util/crc32c_test.cc:31-111 without ever touching the fallback; without
one it must hand every call to the function given to tcrc_set_scalar_fallback (the integration passes the
reference's own ExtendImpl there; the test passes the CPU oracle), and abort loudly when there is none."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
import terarkdb_amd.capi as capi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KAT = json.load(open(os.path.join(HERE, "golden", "crc32c_known_answers.json")))


def oracle_fallback():
    calls = []

    @capi.EXTEND_FN
    def fn(init, data, n):
        calls.append(n)
        return O.extend(init, C.string_at(data, n))

    return fn, calls


def ext(init, b):
    buf = (C.c_char * max(len(b), 1)).from_buffer_copy(b or b"\0")
    return capi.lib.tcrc_crc32c_extend(init, C.cast(buf, C.c_void_p), len(b))


def check_table():
    B = O.fnv_chain()
    for off, ln, stored in KAT["fnv_chain"]:
        row = (off, ln, stored)
        data = B[off:off + ln].tobytes()
        assert ext(0, data) == (~stored) & 0xffffffff, row
        half = ln // 2  # util/crc32c_test.cc:103-111
        assert ext(ext(0, data[:half]), data[half:]) == (~stored) & 0xffffffff, row
    for row in KAT["rfc3720"]:
        assert ext(0, bytes.fromhex(row["hex"])) == int(row["crc"], 16), row


def test_known_answer_file_has_the_expected_shape():
    assert len(KAT["fnv_chain"]) >= 27 and len(KAT["rfc3720"]) >= 5


def test_without_a_gpu_every_call_goes_to_the_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    fn, calls = oracle_fallback()
    capi.lib.tcrc_set_scalar_fallback(C.cast(fn, C.c_void_p))
    try:
        before = capi.lib.tcrc_scalar_fallback_calls()
        assert ext(0, b"123456789") == 0xe3069283
        assert ext(0x12345678, b"") == 0x12345678  # no bytes: no call at all
        check_table()
        assert capi.lib.tcrc_scalar_fallback_calls() - before == len(calls) > 50
    finally:
        capi.lib.tcrc_set_scalar_fallback(None)


def test_without_gpu_and_without_fallback_the_call_aborts():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); import terarkdb_amd.capi as capi; "
            "b = C.create_string_buffer(b'abc'); capi.lib.tcrc_crc32c_extend(0, C.cast(b, C.c_void_p), 3)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "no fallback was set" in r.stderr and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_reference_table_on_the_gpu_never_touches_the_fallback():
    fn, calls = oracle_fallback()
    capi.lib.tcrc_set_scalar_fallback(C.cast(fn, C.c_void_p))
    try:
        assert ext(0, b"123456789") == 0xe3069283
        check_table()
        rng = np.random.default_rng(3)
        S = O.splitmix(1 << 20, 9)
        for _ in range(40):
            a, n = int(rng.integers(0, 4096)), int(rng.integers(0, 1 << 19))
            init = int(rng.integers(0, 2**32))
            assert ext(init, S[a:a + n].tobytes()) == O.extend(init, S[a:a + n].tobytes())
        assert calls == []
    finally:
        capi.lib.tcrc_set_scalar_fallback(None)
