"""File-level helpers over the engine for Python callers: thin views of the C wrappers in
include/tcrc_shim.h (whole-file table verification, database-directory verification, WAL dump,
log-file verification, whole-file crc32c).  Every function needs an Engine, i.e. a GPU."""
import ctypes as C
from dataclasses import dataclass

from . import capi


@dataclass
class TableVerifyResult:
    ok: bool
    declined: bool          # xxHash checksums, compressed index, other table formats: left to the host code
    status: str             # "OK" or the reference's Status text
    file_size: int
    meta_blocks: int
    index_blocks: int
    data_blocks: int
    bytes_covered: int
    mismatches: int
    engine_calls: int


def verify_table_file(engine, path, extent_bytes=0, read_threads=0, index_type_hint=0):
    """BlockBasedTable::VerifyChecksum over one .sst file (table/block_based_table_reader.cc:2616-2695)."""
    stats = (C.c_uint64 * 7)()
    msg = C.create_string_buffer(2048)
    rc = capi.lib.tcrc_shim_sst_verify(engine.ctx, path.encode(), extent_bytes, read_threads, index_type_hint, stats,
                                       msg, 2048)
    if rc < 0:
        raise capi.TcrcError(msg.value.decode())
    return TableVerifyResult(rc == 0, rc == 2, msg.value.decode(), *[int(x) for x in stats])


@dataclass
class DbVerifyResult:
    ok: bool
    status: str
    bad_file: str
    table_files: int
    table_files_declined: int
    table_blocks: int
    table_bytes_covered: int
    log_files: int
    log_bytes: int
    log_records: int
    log_bad_records: int


def verify_db_directory(engine, directory, shard_index=0, shard_count=1):
    """DBImpl::VerifyChecksum's loop over the table files of a directory (db/db_impl.cc:4657-4690) plus a
    checksum scan of its logs and MANIFEST files; shard_index / shard_count deal the files to several GPUs."""
    rep = (C.c_uint64 * 8)()
    msg, bad = C.create_string_buffer(2048), C.create_string_buffer(2048)
    rc = capi.lib.tcrc_shim_verify_db(engine.ctx, directory.encode(), shard_index, shard_count, rep, msg, 2048, bad, 2048)
    if rc < 0:
        raise capi.TcrcError(msg.value.decode())
    return DbVerifyResult(rc == 0, msg.value.decode(), bad.value.decode(), *[int(x) for x in rep])


def dump_wal(engine, path, header=False, print_values=False):
    """`ldb dump_wal` (tools/ldb_cmd.cc:1967-2037): returns (rows, corruption_lines)."""
    cap = 1 << 20
    while True:
        out, err = C.create_string_buffer(cap), C.create_string_buffer(1 << 16)
        n_out, n_err = C.c_size_t(0), C.c_size_t(0)
        rc = capi.lib.tcrc_shim_dump_wal(engine.ctx, path.encode(), int(header), int(print_values), out, cap,
                                         C.byref(n_out), err, 1 << 16, C.byref(n_err))
        if rc != 0:
            raise capi.TcrcError(err.value.decode())
        if n_out.value < cap:
            return out.value.decode(), err.value.decode()
        cap = n_out.value + 1


def file_checksum(engine, path, size_limit=0, buffer_size=0):
    """Whole-file crc32c as the backup engine computes it (utilities/backupable/backupable_db.cc:1414-1452):
    returns (crc32c, bytes covered)."""
    crc, size = C.c_uint32(0), C.c_uint64(0)
    msg = C.create_string_buffer(1024)
    rc = capi.lib.tcrc_shim_file_checksum(engine.ctx, path.encode(), size_limit, buffer_size, C.byref(crc), C.byref(size),
                                          msg, 1024)
    if rc != 0:
        raise capi.TcrcError(msg.value.decode())
    return crc.value, size.value


@dataclass
class LogVerifyResult:
    file_size: int
    records: int           # physical records that pass the structural checks
    mismatches: int        # of them, records whose checksum does not match
    first_bad_block: int   # first 32 KiB block that holds one, -1 if none
    torn_tail: bool        # the file ends inside a header or a record
    blocks: object         # numpy uint32 array [n_blocks, 4]: n_records, first_bad, end_offset, end_code


def verify_log_file(engine, path, log_number, extent_bytes=0, read_threads=0):
    """The checksum pass recovery's reader makes over one log file (db/log_reader.cc:276-406 with checksum = true),
    the file streamed through the engine in pieces."""
    import os

    import numpy as np
    n_blocks = (os.path.getsize(path) + 32767) // 32768 if os.path.exists(path) else 0  # a missing file: the call reports it
    blocks = np.zeros((max(n_blocks, 1), 4), dtype=np.uint32)
    report = (C.c_uint64 * 5)()
    msg = C.create_string_buffer(1024)
    rc = capi.lib.tcrc_shim_wal_verify_file(engine.ctx, path.encode(), log_number, extent_bytes, read_threads,
                                            blocks.ctypes.data, n_blocks, report, msg, 1024)
    if rc != 0:
        raise capi.TcrcError(msg.value.decode())
    first = int(report[3])
    return LogVerifyResult(int(report[0]), int(report[1]), int(report[2]), -1 if first == 2**64 - 1 else first,
                           int(report[4]) in (2, 4), blocks[:n_blocks])
