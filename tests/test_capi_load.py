"""No-GPU checks of the C ABI: the library loads, exports every function the headers declare, refuses to
run without a device (no CPU fallback), and its two scalar helpers match the reference's Mask/Unmask."""
import ctypes as C
import os
import re

import pytest

import oracle_lib as O
import terarkdb_amd.capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECL = re.compile(r"^\s*(?:const\s+)?(?:int|void|uint32_t|uint64_t|size_t|char|tcrc_ctx)\s*\*?\s*(tcrc_\w+)\s*\(", re.M)


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(n for n in DECL.findall(text) if not n.endswith("_fn")))


@pytest.mark.parametrize("header", ["tcrc.h", "tcrc_shim.h"])
def test_every_declared_function_is_exported(header):
    names = declared(header)
    assert len(names) > 10
    missing = [n for n in names if not hasattr(capi.lib, n)]
    assert not missing, missing


def test_binding_table_matches_headers():
    names = set(declared("tcrc.h")) | set(declared("tcrc_shim.h"))
    unbound = sorted(names - set(capi.SIGNATURES))
    unknown = sorted(set(capi.SIGNATURES) - names)
    assert not unknown, unknown
    assert not unbound, unbound


def test_mask_unmask_match_reference_rule():
    # util/crc32c_test.cc:120-126
    crc = O.value(b"foo")
    assert capi.lib.tcrc_mask(crc) == O.mask(crc) == 0xfebe8a61
    assert capi.lib.tcrc_unmask(capi.lib.tcrc_mask(crc)) == crc
    for v in (0, 1, 0xffffffff, 0x80000000, 0xa282ead8, 0x12345678):
        assert capi.lib.tcrc_mask(v) == O.mask(v)
        assert capi.lib.tcrc_unmask(v) == O.unmask(v)
        assert capi.lib.tcrc_unmask(capi.lib.tcrc_unmask(capi.lib.tcrc_mask(capi.lib.tcrc_mask(v)))) == v


def test_no_device_means_no_engine():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    rc = capi.lib.tcrc_init(0, C.byref(ctx))
    assert rc == capi.ERR_NO_DEVICE
    assert b"no CPU fallback" in capi.lib.tcrc_last_error()
    with pytest.raises(capi.TcrcError):
        capi.Engine(0)


def test_product_tree_does_not_reference_the_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "terarkdb_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".cc", ".hip", "Makefile")):
                text = open(os.path.join(d, f), errors="ignore").read()
                if "oracle" in text.lower():
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_shard_range_matches_the_python_arithmetic():
    """tcrc_shard_range is host arithmetic (no GPU): the C ABI's block-range split is sharding.shard_range, for
    even and uneven splits, empty corpora and the 64 GiB corpus of the multi-GPU benchmark."""
    from terarkdb_amd.sharding import shard_range
    for n in (0, 1, 7, 1003, 8388608, 2**40 + 5):
        for count in (1, 2, 3, 8, 64):
            covered = 0
            for i in range(count):
                lo, hi = C.c_uint64(0), C.c_uint64(0)
                capi.lib.tcrc_shard_range(n, i, count, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == shard_range(n, i, count)
                assert lo.value == covered
                covered = hi.value
            assert covered == n
