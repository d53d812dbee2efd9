"""The command-line tool (terarkdb_amd/bin/tcrc) on a GPU box: output shapes of the reference's
`db_bench --benchmarks=crc32c`, `sst_dump --command=verify`, `ldb dump_wal` and the backup checksum."""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import sst_files as S
import wal_dump_cases as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TCRC = os.path.join(ROOT, "terarkdb_amd", "bin", "tcrc")


def run(*args):
    p = subprocess.run([TCRC, *args], capture_output=True, text=True, timeout=120)
    return p.returncode, p.stdout, p.stderr


def test_crc32c_benchmark_lines():
    rc, out, err = run("crc32c", "--block_size=4096", "--bytes=67108864")
    assert rc == 0, err
    lines = out.splitlines()
    pat = r"^%-12s : +[0-9.]+ micros/op [0-9]+ ops/sec; +[0-9.]+ MB/s \(4096 per op\) "
    assert re.match(pat % "crc32c", lines[0]) and re.match(pat % "crc32c_batch", lines[1])
    # the value db_bench prints for its 'x'-filled block
    want = O.value(b"x" * 4096)
    assert ("crc=0x%x" % want) in err


def test_verify_and_file_checksum(tmp_path):
    good = S.build_table(str(tmp_path / "000011.sst"), n_data=200, index="two_level", partition_entries=16)
    bad = S.build_table(str(tmp_path / "000012.sst"), n_data=50, seed=3)
    off, size = bad.handle("data17")
    S.flip(bad.path, off + 9, 2)
    rc, out, err = run("verify", good.path, bad.path)
    assert rc == 1
    assert out == "The file is ok\n"
    assert ("%s is corrupted: %s\n" % (bad.path, S.expected_status(bad))) in err
    rc, out, err = run("file_checksum", good.path)
    raw = np.fromfile(good.path, dtype=np.uint8)
    assert rc == 0 and out == "%s crc32c %d size %d\n" % (good.path, O.value(raw), raw.size)


def test_dump_wal(tmp_path):
    path = str(tmp_path / "000031.log")
    batches = W._batches()
    img = W._write_log(path, batches, 31)
    recs, _, _ = W._offsets_by_oracle(img, 31)
    rc, out, err = run("dump_wal", "--walfile=" + path, "--header")
    assert rc == 0 and err == ""
    assert out == W.HEADER + "\n" + "".join(b.row(off, False) for b, (_, off) in zip(batches, recs))


def test_verify_wal(tmp_path):
    clean = tmp_path / "000123.log"
    img = O.build_wal_image(0, n_random=200)
    clean.write_bytes(img)
    want = O.wal_scan_blocks(img, 123)
    damaged = bytearray(img)
    recs, n = O.list_physical(img, 123)
    off, typ, ln, _ = (int(x) for x in recs[n // 2])
    damaged[off + 7 + ln // 2] ^= 0x80
    torn = tmp_path / "000124.log"
    torn.write_bytes(bytes(damaged[:-3]))  # a flipped bit in the middle and a record cut short at the end
    rc, out, err = run("verify_wal", str(clean), str(torn))
    assert rc == 1, err
    lines = out.splitlines()
    assert lines[0] == "%s: %d bytes, %d physical records, 0 with a checksum mismatch" % (clean, len(img), int(want[:, 0].sum()))
    assert lines[1].startswith("%s: %d bytes, " % (torn, len(img) - 3))
    assert "1 with a checksum mismatch (first in 32 KiB block %d)" % (off // 32768) in lines[1]
    assert lines[1].endswith(", torn record at the end of the file")
    rc, out, err = run("verify_wal", str(tmp_path / "missing.log"))
    assert rc == 1 and "IO error" in err


def _make_db_dir(tmp_path):
    """A directory shaped like a database: table files, write-ahead logs, a MANIFEST, other files."""
    import shutil
    d = tmp_path / "db"
    d.mkdir()
    lays = [S.build_table(str(d / ("%06d.sst" % (10 + i))), n_data=30 + 5 * i, seed=60 + i,
                          index="two_level" if i % 2 else "binary") for i in range(5)]
    zipt = S.build_table(str(d / "000099.sst"), n_data=2)
    raw = bytearray(open(zipt.path, "rb").read())
    raw[-8:] = (0x1122334455667788).to_bytes(8, "little")  # another table format: declined, not corrupt
    open(zipt.path, "wb").write(raw)
    imgs = {}
    for num, recycle in ((3, False), (4, True)):
        img = O.build_wal_image(recycle, log_number=num, n_random=80, seed=num)
        open(str(d / ("%06d.log" % num)), "wb").write(img)
        imgs[num] = img
    here = os.path.dirname(os.path.abspath(__file__))
    shutil.copy(os.path.join(here, "golden", "reference_MANIFEST-CORRECT.bin"), str(d / "MANIFEST-000005"))
    for name in ("CURRENT", "LOCK", "OPTIONS-000007", "LOG"):
        open(str(d / name), "w").write("x\n")
    return d, lays, imgs


def test_verify_db_directory(tmp_path):
    import ctypes as C
    import gpu_util as G
    import terarkdb_amd.capi as capi
    d, lays, imgs = _make_db_dir(tmp_path)
    eng = G.engine()

    def verify(shard=0, shards=1):
        rep = (C.c_uint64 * 8)()
        msg, bad = C.create_string_buffer(1024), C.create_string_buffer(1024)
        rc = capi.lib.tcrc_shim_verify_db(eng.ctx, str(d).encode(), shard, shards, rep, msg, 1024, bad, 1024)
        return rc, [int(x) for x in rep], msg.value.decode(), bad.value.decode()

    n_log_recs = sum(int(O.wal_scan_blocks(imgs[k], k)[:, 0].sum()) for k in imgs)
    manifest = open(str(d / "MANIFEST-000005"), "rb").read()
    n_manifest = int(O.wal_scan_blocks(manifest, 5)[:, 0].sum())
    rc, rep, msg, bad = verify()
    assert (rc, msg, bad) == (0, "OK", "")
    assert rep[0] == 5 and rep[1] == 1
    assert rep[2] == sum(l.n_meta + l.n_index + l.n_data for l in lays)
    assert rep[3] == sum(sz + 1 for l in lays for _, _, sz in l.order)
    assert rep[4] == 3 and rep[5] == sum(len(v) for v in imgs.values()) + len(manifest)
    assert rep[6] == n_log_recs + n_manifest and rep[7] == 0
    # two shards cover the directory between them
    r0, r1 = verify(0, 2), verify(1, 2)
    assert r0[0] == 0 and r1[0] == 0
    assert [a + b for a, b in zip(r0[1], r1[1])] == rep
    # files are dealt by size: no shard carries more than one (largest) file's worth above another
    for shards in (2, 3):
        parts = [verify(k, shards) for k in range(shards)]
        assert all(p[0] == 0 for p in parts)
        assert [sum(col) for col in zip(*[p[1] for p in parts])] == rep
        loads = [p[1][3] + p[1][5] for p in parts]
        largest = max(os.path.getsize(os.path.join(str(d), f)) for f in os.listdir(str(d)))
        assert max(loads) - min(loads) <= largest
    # a corrupt table stops the run with the reference's status; a bad log record is counted
    off, size = lays[2].handle("data7")
    S.flip(lays[2].path, off + 3, 1)
    with open(str(d / "000003.log"), "r+b") as f:
        f.seek(40000)
        b = f.read(1)[0]
        f.seek(40000)
        f.write(bytes([b ^ 0x20]))
    rc, rep2, msg, bad = verify()
    assert rc == 1 and bad == lays[2].path and msg == S.expected_status(lays[2])
    assert rep2[0] == 3 and rep2[4] == 0  # the run ended at the third table file
    os.remove(lays[2].path)
    rc, rep3, msg, bad = verify()
    assert (rc, msg) == (0, "OK") and rep3[0] == 4 and rep3[7] == 1
    eng.close()
    # the same through the command line
    rc, out, err = run("verify_db", str(d))
    assert rc == 1 and "log files: 3" in out and "1 with a checksum mismatch" in out and "000003.log" in out
    assert "table files: 4 verified" in out and "1 declined" in out and out.rstrip().endswith("OK")


def test_python_file_helpers(tmp_path):
    import gpu_util as G
    import terarkdb_amd.capi as capi
    import terarkdb_amd.files as F
    d, lays, imgs = _make_db_dir(tmp_path)
    eng = G.engine()
    r = F.verify_table_file(eng, lays[1].path)
    assert r.ok and r.status == "OK" and r.data_blocks == lays[1].n_data and r.mismatches == 0
    assert F.verify_table_file(eng, str(d / "000099.sst")).declined
    S.flip(lays[1].path, lays[1].handle("index")[0] + 2, 0)
    r = F.verify_table_file(eng, lays[1].path)
    assert not r.ok and not r.declined and r.status == S.expected_status(lays[1])
    db = F.verify_db_directory(eng, str(d))
    assert not db.ok and db.bad_file == lays[1].path and db.status == r.status and db.table_files == 2
    rows, errs = F.dump_wal(eng, str(d / "MANIFEST-000005"))
    assert errs.count("log record too small") >= 0 and isinstance(rows, str)
    raw = np.fromfile(lays[0].path, dtype=np.uint8)
    assert F.file_checksum(eng, lays[0].path) == (O.value(raw), raw.size)
    assert F.file_checksum(eng, lays[0].path, size_limit=1000) == (O.value(raw[:1000]), 1000)
    lg = F.verify_log_file(eng, str(d / "000003.log"), 3)
    want_blocks = O.wal_scan_blocks(imgs[3], 3)
    assert (lg.blocks == want_blocks).all() and lg.records == int(want_blocks[:, 0].sum())
    assert (lg.file_size, lg.mismatches, lg.first_bad_block, lg.torn_tail) == (len(imgs[3]), 0, -1, False)
    with pytest.raises(capi.TcrcError):
        F.verify_log_file(eng, str(d / "missing.log"), 1)
    with pytest.raises(capi.TcrcError):
        F.file_checksum(eng, str(d / "missing"))
    with pytest.raises(capi.TcrcError):
        F.verify_table_file(eng, str(d / "missing.sst"))
    eng.close()


def test_verify_db_module_under_torchrun(tmp_path):
    """The multi-GPU directory verification with one rank (the RCCL path; more ranks need more GPUs)."""
    import socket
    import sys
    d, lays, imgs = _make_db_dir(tmp_path)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), "-m", "terarkdb_amd.verify_db", str(d)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "table files: 5 verified" in p.stdout and "1 declined" in p.stdout and p.stdout.rstrip().endswith("OK")
    S.flip(lays[3].path, lays[3].handle("data2")[0] + 1, 6)
    p = subprocess.run([sys.executable, "-m", "terarkdb_amd.verify_db", str(d)], capture_output=True, text=True,
                       timeout=240, cwd=ROOT)
    assert p.returncode == 1 and ("%s is corrupted: %s" % (lays[3].path, S.expected_status(lays[3]))) in p.stdout


def test_integration_example_program(tmp_path):
    """The call sequences of the integration guide, compiled and run as one program."""
    exe = os.path.join(ROOT, "terarkdb_amd", "bin", "integration_example")
    p = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "integration example ok" in p.stdout and "verdicts from the engine" in p.stdout
