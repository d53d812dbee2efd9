"""Checks over the two MANIFEST files the reference ships (tests/golden/README.md): every record must
verify, reading must return every logical record with no drops, and writing those records again must
reproduce the files byte for byte.  `mk(recycle, log_number=...)` builds a writer/reader fixture."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = {"reference_MANIFEST-CORRECT.bin": (25, 25), "reference_MANIFEST-WRONG.bin": (10, 4)}  # physical, logical


def load(name):
    return open(os.path.join(HERE, "golden", name), "rb").read()


def read_all(t):
    out = []
    while True:
        r = t.read()
        if r == "EOF":
            return out
        out.append(r)


def roundtrip(mk, name, put_image):
    """put_image(t, bytes) places the file's bytes into fixture t's file."""
    img = load(name)
    n_phys, n_logical = FILES[name]
    reader_side = mk(0, log_number=0)
    put_image(reader_side, img)
    records = read_all(reader_side)
    assert len(records) == n_logical
    assert reader_side.dropped_bytes() == 0 and reader_side.report_message() == ""
    writer_side = mk(0, log_number=0)
    for r in records:
        writer_side.write(r)
    assert writer_side.contents() == img
    return records
