"""ctypes binding of libtcrc.so (declarations mirror include/tcrc.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TCRC_LIBRARY") or os.path.join(_HERE, "lib", "libtcrc.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C terarkdb_amd/csrc` (or __graft_entry__.build()). "
        "The engine has no CPU fallback.")

# PyTorch-ROCm wheels ship their own copy of the HIP/HSA runtime.  Two copies in one process
# cannot both see the GPU, so when torch is around it is loaded first and libtcrc.so then binds
# to the runtime torch brought (same soname) instead of pulling in /opt/rocm's.
try:
    import torch  # noqa: F401
except ImportError:  # a process without torch just uses the system runtime
    pass

lib = C.CDLL(LIB_PATH)

OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_TOO_LARGE = -1, -2, -3, -4
F_MASK, F_EXPECT_AFTER, F_EXPECT_BEFORE, F_STORE_AFTER, F_SHORT_SPANS = 1, 2, 4, 8, 16
CNT_MISMATCH, CNT_FIRST_BAD, CNT_CHECKED, CNT_WORDS = 0, 1, 2, 3


class WalBlock(C.Structure):
    _fields_ = [("n_records", C.c_uint32), ("first_bad", C.c_uint32), ("end_offset", C.c_uint32),
                ("end_code", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("batch_launches", "batch_buffers", "host_calls", "host_bytes_in", "host_nanos",
                                          "log_images", "log_bytes", "long_calls", "long_bytes", "host_lanes")]


class Desc(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("len", C.c_uint32), ("init", C.c_uint32)]


_vp, _u64, _u32, _i = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
SIGNATURES = {
    "tcrc_init": (_i, [_i, C.POINTER(_vp)]),
    "tcrc_shutdown": (_i, [_vp]),
    "tcrc_last_error": (C.c_char_p, []),
    "tcrc_describe": (C.c_char_p, [_vp]),
    "tcrc_mask": (_u32, [_u32]),
    "tcrc_unmask": (_u32, [_u32]),
    "tcrc_dev_counts_reset": (_i, [_vp, _vp, _vp]),
    "tcrc_dev_wal_verify": (_i, [_vp, _vp, _u64, _u64, _vp, _vp, C.POINTER(_u64), _vp]),
    "tcrc_dev_extend_long": (_i, [_vp, _vp, _vp, _u64, _u32, _vp, _vp]),
    "tcrc_dev_extend_long_noalloc": (_i, [_vp, _vp, _vp, _u64, _u32, _vp, _vp]),
    "tcrc_dev_extend_long_enqueue": (_i, [_vp, _vp, _vp, _u64, _u32, _vp, _vp]),
    "tcrc_dev_probe_read": (_i, [_vp, _vp, _u64, _vp, _u32, _vp]),
    "tcrc_dev_blocks_stamp": (_i, [_vp, _vp, _u64, _u32, _u64, _u32, _i, _vp, _u32, _vp, _vp]),
    "tcrc_dev_blocks_verify": (_i, [_vp, _vp, _u64, _u32, _u64, _u32, _i, _vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    "tcrc_dev_batch_stamp": (_i, [_vp, _vp, _vp, _u64, _vp, _u32, _vp, _vp]),
    "tcrc_dev_batch_verify": (_i, [_vp, _vp, _vp, _u64, _vp, _u32, _vp, _vp, _vp, _vp, _vp]),
}
_sz = C.c_size_t
EXTEND_FN = C.CFUNCTYPE(_u32, _u32, _vp, _sz)  # the reference's `Function` type (util/crc32c.cc:439)
STAMP_FN = C.CFUNCTYPE(_i, _vp, _vp, _sz, C.POINTER(Desc), _sz, C.POINTER(_u32))
SCAN_FN = C.CFUNCTYPE(_i, _vp, _vp, _sz, _u64, C.POINTER(WalBlock))
CHECK_FN = C.CFUNCTYPE(_i, _vp, _vp, _sz, C.POINTER(Desc), _sz, C.POINTER(_u32), C.POINTER(C.c_uint8))
SIGNATURES.update({
    "tcrc_host_alloc": (_vp, [_vp, _sz]),
    "tcrc_host_free": (None, [_vp, _vp]),
    "tcrc_host_batch_stamp": (_i, [_vp, _vp, _u64, _vp, _u64, _vp, _u32, _vp]),
    "tcrc_host_blocks_stamp": (_i, [_vp, _vp, _u64, _u32, _u64, _u32, _i, _u32, _vp]),
    "tcrc_host_batch_verify": (_i, [_vp, _vp, _u64, _vp, _u64, _vp, _u32, _vp, _vp, _vp, _vp]),
    "tcrc_host_wal_verify": (_i, [_vp, _vp, _u64, _u64, _vp, _vp, C.POINTER(_u64)]),
    "tcrc_host_wal_verify_part": (_i, [_vp, _vp, _u64, _u64, _u64, _i, _vp, _vp, C.POINTER(_u64)]),
    "tcrc_dev_wal_verify_part": (_i, [_vp, _vp, _u64, _u64, _u64, _i, _vp, _vp, C.POINTER(_u64), _vp]),
    "tcrc_dev_wal_frame": (_i, [_vp, _vp, _vp, _u64, _u64, _i, _u32, _vp, _u64, C.POINTER(_u64), C.POINTER(_u32),
                                C.POINTER(_u64), _vp]),
    "tcrc_wal_framed_size": (_u64, [_vp, _u64, _i, _u32]),
    "tcrc_wal_plan_create": (_i, [_vp, _vp, _u64, _u64, _i, _u32, C.POINTER(_vp)]),
    "tcrc_wal_plan_free": (None, [_vp]),
    "tcrc_wal_plan_caller_waits": (C.c_int, [_vp, C.c_int]),
    "tcrc_wal_plan_image_bytes": (_u64, [_vp]),
    "tcrc_wal_plan_block_offset": (_u32, [_vp]),
    "tcrc_wal_plan_physical_records": (_u64, [_vp]),
    "tcrc_dev_wal_frame_enqueue": (_i, [_vp, _vp, _vp, _vp, _u64, _vp]),
    "tcrc_dev_wal_list": (_i, [_vp, _vp, _u64, _u64, _vp, _vp, _u64, C.POINTER(_u64), _vp]),
    "tcrc_host_wal_list": (_i, [_vp, _vp, _u64, _u64, _vp, _vp, _u64, C.POINTER(_u64)]),
    "tcrc_host_blocks_verify": (_i, [_vp, _vp, _u64, _u32, _u64, _u32, _i, _u32, _vp, _vp, _vp, _vp]),
    "tcrc_shard_range": (None, [_u64, _u32, _u32, C.POINTER(_u64), C.POINTER(_u64)]),
    "tcrc_group_init": (_i, [C.POINTER(_i), _i, C.POINTER(_vp)]),
    "tcrc_group_shutdown": (_i, [_vp]),
    "tcrc_group_size": (_i, [_vp]),
    "tcrc_group_ctx": (_vp, [_vp, _i]),
    "tcrc_group_device": (_i, [_vp, _i]),
    "tcrc_group_dev_blocks_verify": (_i, [_vp, C.POINTER(_vp), _u64, _u32, _u64, _u32, _i, _u32, C.POINTER(_vp), _vp]),
    "tcrc_group_host_blocks_verify": (_i, [_vp, _vp, _u64, _u32, _u64, _u32, _i, _u32, _vp, _vp, _vp]),
    "tcrc_rccl_get_unique_id": (_i, [_vp]),
    "tcrc_rccl_comm_init_rank": (_i, [_vp, _i, _vp, _i, C.POINTER(_vp)]),
    "tcrc_rccl_comm_destroy": (_i, [_vp]),
    "tcrc_rccl_allreduce_counts": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "tcrc_get_stats": (_i, [_vp, C.POINTER(Stats)]),
    "tcrc_reset_stats": (_i, [_vp]),
    "tcrc_debug_tables": (_i, [_i, _vp, _vp, _sz, C.POINTER(_sz)]),
    "tcrc_extend": (_i, [_vp, _u32, _vp, _sz, C.POINTER(_u32)]),
    "tcrc_value": (_i, [_vp, _vp, _sz, C.POINTER(_u32)]),
    "tcrc_crc32c_extend": (_u32, [_u32, _vp, _sz]),
    "tcrc_set_scalar_fallback": (None, [_vp]),
    "tcrc_scalar_fallback_calls": (_u64, []),
    "tcrc_shim_memfile_new": (_vp, []),
    "tcrc_shim_memfile_free": (None, [_vp]),
    "tcrc_shim_memfile_size": (_sz, [_vp]),
    "tcrc_shim_memfile_data": (_vp, [_vp]),
    "tcrc_shim_memfile_drop": (None, [_vp, _sz]),
    "tcrc_shim_memfile_append": (None, [_vp, _vp, _sz]),
    "tcrc_shim_wal_writer_new": (_vp, [_vp, _vp, _u64, _i, _i, _i]),
    "tcrc_shim_wal_writer_new_cb": (_vp, [STAMP_FN, _vp, _vp, _u64, _i, _i, _i]),
    "tcrc_shim_wal_writer_open": (_vp, [_vp, C.c_char_p, _u64, _i, _i]),
    "tcrc_shim_wal_writer_add_record": (_i, [_vp, _vp, _sz]),
    "tcrc_shim_wal_writer_add_records_fixed": (_i, [_vp, _vp, _sz, _sz]),
    "tcrc_shim_wal_writer_write_buffer": (_i, [_vp]),
    "tcrc_shim_wal_writer_status": (C.c_char_p, [_vp]),
    "tcrc_shim_wal_writer_physical_records": (_u64, [_vp]),
    "tcrc_shim_wal_writer_crc_bytes": (_u64, [_vp]),
    "tcrc_shim_wal_writer_free": (None, [_vp]),
    "tcrc_shim_wal_reader_new": (_vp, [_vp, _vp, _i, _u64, _i]),
    "tcrc_shim_wal_reader_new_cb": (_vp, [SCAN_FN, _vp, _vp, _i, _u64, _i]),
    "tcrc_shim_wal_reader_open": (_vp, [_vp, C.c_char_p, _i, _u64, _i]),
    "tcrc_shim_wal_reader_open_cb": (_vp, [SCAN_FN, _vp, C.c_char_p, _i, _u64, _i]),
    "tcrc_shim_wal_reader_attach_image": (_i, [_vp, _vp, _sz]),
    "tcrc_shim_wal_reader_read_record": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_sz)]),
    "tcrc_shim_wal_reader_dropped_bytes": (_sz, [_vp]),
    "tcrc_shim_wal_reader_report": (C.c_char_p, [_vp]),
    "tcrc_shim_wal_reader_is_eof": (_i, [_vp]),
    "tcrc_shim_wal_reader_unmark_eof": (None, [_vp]),
    "tcrc_shim_wal_reader_last_record_offset": (_u64, [_vp]),
    "tcrc_shim_wal_reader_verifier_calls": (_u64, [_vp]),
    "tcrc_shim_wal_reader_force_error": (None, [_vp, _sz]),
    "tcrc_shim_wal_reader_force_eof": (None, [_vp, _sz]),
    "tcrc_shim_wal_reader_read_after_partial": (_i, [_vp]),
    "tcrc_shim_wal_reader_free": (None, [_vp]),
    "tcrc_shim_block_trailers": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "tcrc_shim_write_raw_blocks": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "tcrc_shim_check_blocks": (_i, [_vp, _vp, _sz, _u64, _vp, _vp, _sz, _i, C.c_char_p, _vp, C.c_char_p, _sz]),
    "tcrc_shim_sst_verify": (_i, [_vp, C.c_char_p, _sz, _i, _i, C.POINTER(_u64), C.c_char_p, _sz]),
    "tcrc_shim_sst_verify_cb": (_i, [CHECK_FN, _vp, C.c_char_p, _sz, _i, _i, C.POINTER(_u64), C.c_char_p, _sz]),
    "tcrc_shim_dump_wal": (_i, [_vp, C.c_char_p, _i, _i, C.c_char_p, _sz, C.POINTER(_sz), C.c_char_p, _sz,
                                C.POINTER(_sz)]),
    "tcrc_shim_dump_wal_cb": (_i, [SCAN_FN, _vp, C.c_char_p, _i, _i, C.c_char_p, _sz, C.POINTER(_sz), C.c_char_p, _sz,
                                   C.POINTER(_sz)]),
    "tcrc_shim_footer_encode": (_i, [_i, _u32, _i, _u64, _u64, _u64, _u64, _vp]),
    "tcrc_shim_footer_decode": (_i, [_vp, _sz, C.POINTER(_i), C.POINTER(_u32), C.POINTER(_u64), _vp, C.c_char_p, _sz]),
    "tcrc_shim_cache_records_serialize": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "tcrc_shim_cache_records_check": (_i, [_vp, _vp, _sz, _vp, _vp, _sz, _vp]),
    "tcrc_shim_wal_verify_file": (_i, [_vp, C.c_char_p, _u64, _sz, _i, _vp, _sz, _vp, C.c_char_p, _sz]),
    "tcrc_shim_verify_db": (_i, [_vp, C.c_char_p, _u32, _u32, C.POINTER(_u64), C.c_char_p, _sz, C.c_char_p, _sz]),
    "tcrc_shim_verify_db_group": (_i, [_vp, C.c_char_p, C.POINTER(_u64), C.c_char_p, _sz, C.c_char_p, _sz]),
    "tcrc_shim_file_checksum": (_i, [_vp, C.c_char_p, _u64, _sz, C.POINTER(_u32), C.POINTER(_u64), C.c_char_p, _sz]),
    "tcrc_shim_copy_file": (_i, [_vp, C.c_char_p, C.c_char_p, _u64, _sz, _i, C.POINTER(_u32), C.POINTER(_u64),
                                 C.c_char_p, _sz]),
})
for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(lib, _name)
    _f.restype = _res
    _f.argtypes = _args


class TcrcError(RuntimeError):
    pass


def check(rc):
    if rc != OK:
        raise TcrcError(f"tcrc error {rc}: {lib.tcrc_last_error().decode()}")


def _dp(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    """One engine per GPU.  Tensors passed in must live on that GPU."""

    def __init__(self, device=0):
        self.ctx = C.c_void_p()
        check(lib.tcrc_init(device, C.byref(self.ctx)))
        self.device = device

    def close(self):
        if self.ctx:
            lib.tcrc_shutdown(self.ctx)
            self.ctx = C.c_void_p()

    def describe(self):
        return lib.tcrc_describe(self.ctx).decode()

    @staticmethod
    def _stream(stream):
        if stream is None:
            import torch
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)
        return C.c_void_p(stream)

    def blocks_stamp(self, base, n_blocks, block_len, stride, out, init=0, suffix_byte=-1, suffix=None,
                     flags=F_MASK, stream=None):
        check(lib.tcrc_dev_blocks_stamp(self.ctx, _dp(base), n_blocks, block_len, stride, init, suffix_byte,
                                        _dp(suffix), flags, _dp(out), self._stream(stream)))

    def blocks_verify(self, base, n_blocks, block_len, stride, counts, expected=None, out=None, ok=None, init=0,
                      suffix_byte=-1, suffix=None, flags=0, stream=None):
        check(lib.tcrc_dev_blocks_verify(self.ctx, _dp(base), n_blocks, block_len, stride, init, suffix_byte,
                                         _dp(suffix), flags, _dp(expected), _dp(out), _dp(ok), _dp(counts),
                                         self._stream(stream)))

    def batch_stamp(self, base, desc, n, out, suffix=None, flags=0, stream=None):
        check(lib.tcrc_dev_batch_stamp(self.ctx, _dp(base), _dp(desc), n, _dp(suffix), flags, _dp(out),
                                       self._stream(stream)))

    def batch_verify(self, base, desc, n, counts, expected=None, out=None, ok=None, suffix=None, flags=0,
                     stream=None):
        check(lib.tcrc_dev_batch_verify(self.ctx, _dp(base), _dp(desc), n, _dp(suffix), flags, _dp(expected),
                                        _dp(out), _dp(ok), _dp(counts), self._stream(stream)))

    def probe_read(self, base, nbytes, scratch, wgs_per_cu=1, stream=None):
        check(lib.tcrc_dev_probe_read(self.ctx, _dp(base), nbytes, _dp(scratch), wgs_per_cu, self._stream(stream)))

    def wal_verify(self, img, size, log_number, blocks, counts, stream=None, want_records=True):
        """blocks: int32/uint32 device tensor of shape [ceil(size/32768), 4]; returns the record count (which
        makes the call wait for the scan), or None with want_records=False (the call then only enqueues)."""
        n = C.c_uint64(0)
        check(lib.tcrc_dev_wal_verify(self.ctx, _dp(img), size, log_number, _dp(blocks), _dp(counts),
                                      C.byref(n) if want_records else None, self._stream(stream)))
        return n.value if want_records else None

    def wal_frame(self, src, offsets, lens, log_number, recycle, dst, block_offset=0, stream=None):
        """Frames the records src[offsets[i] : offsets[i] + lens[i]] into dst (both device uint8 tensors);
        returns (bytes written, block offset afterwards, physical records)."""
        n = len(offsets)
        arr = (Desc * max(n, 1))()
        for i in range(n):
            arr[i].offset, arr[i].len, arr[i].init = int(offsets[i]), int(lens[i]), 0
        written, bo, phys = C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
        check(lib.tcrc_dev_wal_frame(self.ctx, _dp(src), arr, n, log_number, int(recycle), block_offset, _dp(dst),
                                     dst.numel(), C.byref(written), C.byref(bo), C.byref(phys), self._stream(stream)))
        return written.value, bo.value, phys.value

    def wal_plan(self, offsets, lens, log_number, recycle, block_offset=0):
        """Plans the framing of the records (offsets[i], lens[i]) - host arithmetic only, plus the upload of the
        plan; returns a WalPlan to hand to wal_frame_enqueue any number of times, and to close()."""
        n = len(offsets)
        arr = (Desc * max(n, 1))()
        for i in range(n):
            arr[i].offset, arr[i].len, arr[i].init = int(offsets[i]), int(lens[i]), 0
        h = C.c_void_p()
        check(lib.tcrc_wal_plan_create(self.ctx, arr, n, log_number, int(recycle), block_offset, C.byref(h)))
        return WalPlan(h)

    def wal_frame_enqueue(self, plan, src, dst, stream=None):
        """Enqueues the framing kernel of `plan` on the stream: src -> dst (device uint8 tensors).  Nothing is
        waited for; the image is complete once the stream has run."""
        check(lib.tcrc_dev_wal_frame_enqueue(self.ctx, plan.handle, _dp(src), _dp(dst), dst.numel(), self._stream(stream)))

    def wal_list(self, img, size, log_number, blocks, records, capacity, stream=None):
        """records: uint8 device tensor of 16 * capacity bytes (tcrc_wal_record); returns the total count."""
        n = C.c_uint64(0)
        check(lib.tcrc_dev_wal_list(self.ctx, _dp(img), size, log_number, _dp(blocks), _dp(records), capacity,
                                    C.byref(n), self._stream(stream)))
        return n.value

    def extend_long(self, base, offsets, lens, inits, out, segment_len=0, stream=None, noalloc=False, enqueue=False):
        """Host descriptor arrays, device data; out is a device int32/uint32 tensor of len(offsets).
        noalloc=True takes the entry point that allocates nothing (at most 64 buffers; only enqueues);
        enqueue=True the one that only enqueues (at most 64 buffers)."""
        n = len(offsets)
        arr = (Desc * max(n, 1))()
        for i in range(n):
            arr[i].offset, arr[i].len, arr[i].init = int(offsets[i]), int(lens[i]), int(inits[i])
        fn = (lib.tcrc_dev_extend_long_noalloc if noalloc else
              lib.tcrc_dev_extend_long_enqueue if enqueue else lib.tcrc_dev_extend_long)
        check(fn(self.ctx, _dp(base), arr, n, segment_len, _dp(out), self._stream(stream)))

    def extend(self, init, data):
        """crc32c::Extend(init, data) for host bytes (bytes, bytearray or a contiguous numpy array)."""
        import numpy as np
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        out = _u32(0)
        check(lib.tcrc_extend(self.ctx, init, C.c_void_p(a.ctypes.data) if a.size else None, a.nbytes, C.byref(out)))
        return out.value

    def value(self, data):
        return self.extend(0, data)

    def stats(self):
        st = Stats()
        check(lib.tcrc_get_stats(self.ctx, C.byref(st)))
        return {n: getattr(st, n) for n, _ in Stats._fields_}

    def counts_reset(self, counts, stream=None):
        check(lib.tcrc_dev_counts_reset(self.ctx, _dp(counts), self._stream(stream)))


class WalPlan:
    """A framing plan (tcrc_wal_plan): image size, block offset afterwards and physical record count are known
    as soon as it exists."""

    def __init__(self, handle):
        self.handle = handle
        self.image_bytes = lib.tcrc_wal_plan_image_bytes(handle)
        self.block_offset = lib.tcrc_wal_plan_block_offset(handle)
        self.physical_records = lib.tcrc_wal_plan_physical_records(handle)

    def caller_waits(self, yes=True):
        """No event behind the launches from now on: the caller waits for its stream before close()."""
        check(lib.tcrc_wal_plan_caller_waits(self.handle, int(yes)))

    def close(self):
        if self.handle:
            lib.tcrc_wal_plan_free(self.handle)
            self.handle = C.c_void_p()


def make_desc_tensor(offsets, lens, inits, device):
    """Packs (offset u64, len u32, init u32) triples into a device tensor of tcrc_desc."""
    import numpy as np
    import torch
    n = len(offsets)
    a = np.zeros(n, dtype=[("offset", "<u8"), ("len", "<u4"), ("init", "<u4")])
    a["offset"] = offsets
    a["len"] = lens
    a["init"] = inits
    return torch.from_numpy(a.view(np.uint8).copy()).to(device)
